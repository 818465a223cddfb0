"""thepayne_amd.launch: one rank per GPU from one command line (what `bench.py --gpus N` and tools/fit_stars.py use),
driven here with stub children on CPU: the ranks rendezvous over gloo and gather their rank ids; a rank that fails
ends its siblings and becomes the parent's exit code; `prepare` runs once, in the parent, before any rank starts."""
import os
import sys
import textwrap
import time

from thepayne_amd.launch import launch_ranks, rank_env, free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GATHER = textwrap.dedent('''
    import os, sys, json
    import torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["LOCAL_RANK"] == os.environ["RANK"]
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert os.path.exists(sys.argv[1] + "/prepared")            # the parent's prepare() ran before any rank
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = torch.tensor([float(rank)], dtype=torch.float64)
    bufs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine)
    got = sorted(int(b.item()) for b in bufs)
    assert got == list(range(world)), got
    if rank == 0:
        print(json.dumps({"ranks": got}), flush=True)
    open(sys.argv[1] + "/done_%d" % rank, "w").write("ok")
    dist.destroy_process_group()
''')

FAIL = textwrap.dedent('''
    import os, sys, time
    rank = int(os.environ["RANK"])
    open(sys.argv[1] + "/started_%d" % rank, "w").write("x")
    if rank == 1:
        time.sleep(0.5)
        sys.exit(7)
    time.sleep(120)                                              # a sibling waiting in a collective that will never complete
''')


def test_ranks_rendezvous_and_gather(tmp_path):
    script = tmp_path / "gather.py"
    script.write_text(GATHER)
    out = tmp_path / "rank0.out"
    with open(out, "w") as fh:
        rc = launch_ranks(3, [sys.executable, str(script), str(tmp_path)],
                          prepare=lambda: open(tmp_path / "prepared", "w").write("1"), rank0_stdout=fh, timeout_s=240)
    assert rc == 0
    assert all((tmp_path / ("done_%d" % r)).exists() for r in range(3))
    assert '"ranks": [0, 1, 2]' in out.read_text()             # rank 0's line is relayed, the other ranks' stdout is dropped


def test_a_failing_rank_ends_its_siblings_and_the_parent_fails(tmp_path):
    script = tmp_path / "fail.py"
    script.write_text(FAIL)
    t0 = time.monotonic()
    rc = launch_ranks(3, [sys.executable, str(script), str(tmp_path)], timeout_s=60)
    dt = time.monotonic() - t0
    assert rc == 7                                               # the failing rank's code
    assert dt < 30, dt                                           # the sleeping siblings were terminated, not waited for
    assert all((tmp_path / ("started_%d" % r)).exists() for r in range(3))


def test_timeout_and_environment():
    env = rank_env(2, 4, 12345, base={})
    assert env == {"RANK": "2", "LOCAL_RANK": "2", "WORLD_SIZE": "4", "LOCAL_WORLD_SIZE": "4", "MASTER_ADDR": "127.0.0.1",
                   "MASTER_PORT": "12345", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    assert 1024 < free_port() < 65536
    t0 = time.monotonic()
    rc = launch_ranks(2, [sys.executable, "-c", "import time; time.sleep(60)"], timeout_s=1.0)
    assert rc == 124 and time.monotonic() - t0 < 20


DRIVER = textwrap.dedent('''
    import sys
    sys.path.insert(0, sys.argv[1])
    from thepayne_amd.launch import launch_ranks
    child = "import os, sys, time; open(sys.argv[1] + '/pid_%s' % os.environ['RANK'], 'w').write(str(os.getpid())); time.sleep(300)"
    sys.exit(launch_ranks(2, [sys.executable, "-c", child, sys.argv[2]], timeout_s=600))
''')


def test_a_terminated_parent_leaves_no_ranks_behind(tmp_path):
    """SIGTERM to the process that called launch_ranks (a cancelled job, a driver's timeout): the ranks are ended with it."""
    import signal
    import subprocess
    script = tmp_path / "driver.py"
    script.write_text(DRIVER)
    parent = subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path)])
    t0 = time.monotonic()
    while not all((tmp_path / ("pid_%d" % r)).exists() and (tmp_path / ("pid_%d" % r)).read_text() for r in range(2)):
        assert time.monotonic() - t0 < 60 and parent.poll() is None
        time.sleep(0.05)
    pids = [int((tmp_path / ("pid_%d" % r)).read_text()) for r in range(2)]
    parent.send_signal(signal.SIGTERM)
    parent.wait(timeout=30)
    assert parent.returncode != 0

    def alive(pid):
        try:
            with open("/proc/%d/stat" % pid) as fh:
                return fh.read().split(")")[-1].split()[0] != "Z"
        except OSError:
            return False
    t0 = time.monotonic()
    while any(alive(p) for p in pids) and time.monotonic() - t0 < 10:
        time.sleep(0.05)
    assert not any(alive(p) for p in pids), pids


def test_every_collective_of_the_multi_gpu_paths_over_gloo(tmp_path):
    """tools/collectives_check.py (what tests/test_rccl_world1_gpu.py runs over RCCL on the GPU box): a forced group of one rank,
    and two ranks, give the same tables."""
    import numpy as np
    check = os.path.join(ROOT, "tools", "collectives_check.py")
    got = {}
    for n in (1, 2):
        out = str(tmp_path / ("g%d.npz" % n))
        with open(tmp_path / ("g%d.log" % n), "w") as fh:
            assert launch_ranks(n, [sys.executable, check, "--backend", "gloo", "--out", out], rank0_stdout=fh, timeout_s=300) == 0
        got[n] = np.load(out)
    for k in ("table", "kept", "once"):
        assert np.array_equal(got[1][k], got[2][k]), k
    assert int(got[1]["world"]) == 1 and int(got[2]["world"]) == 2
