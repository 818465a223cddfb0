"""One rank per GPU from one command line (bench.py --gpus N, tools/fit_stars.py).

The parent process never imports torch and never touches HIP: it builds the library once (so that N ranks do not
find it stale together), starts N fresh children with the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR=127.0.0.1 / MASTER_PORT) and waits.  A child that exits non-zero ends its siblings (they would wait in a
collective for ever) and becomes the parent's exit code.  No process that has initialised a GPU is ever replaced by
another program (os.exec*): ranks are children, started before any GPU call.
"""
import os
import signal
import socket
import subprocess
import sys
import time

__all__ = ["free_port", "rank_env", "launch_ranks"]


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank, world, port, base=None):
    """The torchrun environment of one rank on one node."""
    return dict(base if base is not None else os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                HSA_ENABLE_IPC_MODE_LEGACY="0")


def launch_ranks(n, cmd, prepare=None, poll_s=0.05, timeout_s=None, rank0_stdout=None):
    """Start `cmd` (argv list) n times as ranks 0..n-1; rank 0 inherits stdout (or gets `rank0_stdout`), the others'
    stdout is discarded, stderr is inherited by all.  `prepare()` runs once in the parent first (e.g. build_lib: the
    parent holds no GPU, so compiling there is safe, and the ranks then find the library fresh).  Returns the exit code:
    0 if every rank returned 0, else the first non-zero code seen (siblings still running are terminated, then killed);
    124 if `timeout_s` ran out."""
    if prepare is not None:
        prepare()
    port = free_port()
    procs = []

    def end(ps):
        for q in ps:
            if q.poll() is None:
                q.terminate()
        t_end = time.monotonic() + 5.0
        for q in ps:
            try:
                q.wait(timeout=max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                q.kill()
                q.wait()

    # A parent that is interrupted, terminated or hit by an exception must not leave N GPU ranks behind (they hold their
    # devices and may wait in a collective for ever): SIGTERM becomes an exception here, and `finally` ends whoever still runs.
    def on_term(signum, frame):
        raise KeyboardInterrupt("signal %d" % signum)
    old_term, term_set = None, False
    try:
        old_term = signal.signal(signal.SIGTERM, on_term)
        term_set = True                      # (old_term None = a handler installed from C: it cannot be put back from here)
    except ValueError:                       # not the main thread: no handler, the finally below still covers exceptions
        pass
    try:
        for r in range(n):
            procs.append(subprocess.Popen(list(cmd), env=rank_env(r, n, port),
                                          stdout=(rank0_stdout if r == 0 else subprocess.DEVNULL)))
        rc = 0
        pending = list(procs)
        t0 = time.monotonic()
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    end(pending)                 # a failed rank leaves the others waiting in a collective: end them
                    pending = []
                    break
            if pending and timeout_s is not None and time.monotonic() - t0 > timeout_s:
                end(pending)
                return 124
            if pending:
                time.sleep(poll_s)
        return rc
    finally:
        # the clean-up itself must not be interrupted (a second SIGTERM / SIGINT while `end` waits would leave ranks alive):
        # both signals are ignored until every rank is gone, then the previous handlers come back
        old_int = None
        try:
            signal.signal(signal.SIGTERM, signal.SIG_IGN)
            old_int = signal.signal(signal.SIGINT, signal.SIG_IGN)
        except ValueError:
            pass
        end([q for q in procs if q.poll() is None])
        try:
            if old_int is not None:
                signal.signal(signal.SIGINT, old_int)
            if term_set:                     # our handler was installed: the one before comes back (a C-level one: the default)
                signal.signal(signal.SIGTERM, old_term if old_term is not None else signal.SIG_DFL)
            elif old_term is not None:
                signal.signal(signal.SIGTERM, old_term)
        except ValueError:
            pass


if __name__ == "__main__":          # python -m thepayne_amd.launch N prog args...
    sys.exit(launch_ranks(int(sys.argv[1]), sys.argv[2:]))
