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
