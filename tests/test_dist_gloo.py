"""The multi-star path on 2 CPU ranks (gloo): sharding + the single all_gather."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import sys, numpy as np
    sys.path.insert(0, %r)
    from thepayne_amd import dist as pdist
    from thepayne_amd.sampler import NestedSampler

    def fit(star, idx):
        mu, sig = star
        ll = lambda V: -0.5 * np.sum(((V - mu) / sig) ** 2, axis=1)
        s = NestedSampler(ll, lambda U: U.copy(), 2, nlive=60, bound="single", sample="unif", batched=True,
                          rstate=np.random.default_rng(idx))
        s.run_nested(dlogz=0.5)
        return s.summary()

    stars = [(0.3 + 0.1 * i, 0.05) for i in range(5)]
    table = pdist.fit_stars(stars, fit, summary_length=5 + 5 * 2, backend="gloo")
    rank, world, _ = pdist.init_from_env("gloo")
    np.save(sys.argv[1] + "/table_%%d.npy" %% rank, table)
    pdist.finalize()
''') % ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_shard_stars_and_gather_summaries(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err[-3000:]
    t0, t1 = np.load(tmp_path / "table_0.npy"), np.load(tmp_path / "table_1.npy")
    assert t0.shape == (5, 15) and np.array_equal(t0, t1)          # every rank holds every star's summary
    assert np.isfinite(t0).all()
    # posterior means follow the per-star truth (columns 5, 10 = mean of parameter 0, 1)
    for i in range(5):
        assert abs(t0[i, 5] - (0.3 + 0.1 * i)) < 0.03 and abs(t0[i, 10] - (0.3 + 0.1 * i)) < 0.03


def test_shard_is_a_partition():
    from thepayne_amd.dist import shard
    for n in (0, 1, 7, 8, 9):
        for w in (1, 2, 8):
            got = sorted(i for r in range(w) for i in shard(n, r, w))
            assert got == list(range(n))


WORKER2 = textwrap.dedent('''
    import sys, numpy as np
    sys.path.insert(0, %r)
    from thepayne_amd import dist as pdist
    rank, world, _ = pdist.init_from_env("gloo")
    rng = np.random.default_rng(7)                      # same candidates on every rank
    out = {}
    for B in (1, 2, 5, 64, 67):
        theta = rng.normal(size=(B, 3))
        calls = []
        def fn(block):
            calls.append(len(block))
            return (block ** 2).sum(axis=1) + 0.5
        out["B%%d" %% B] = pdist.sharded_lnlike(fn, theta, rank, world)
        out["ref%%d" %% B] = (theta ** 2).sum(axis=1) + 0.5
        out["n%%d" %% B] = np.array(calls)
    np.savez(sys.argv[1] + "/sharded_%%d.npz" %% rank, **out)
    pdist.finalize()
''') % ROOT


def test_two_ranks_split_one_batch_and_gather_lnl(tmp_path):
    """Within-star sharding: contiguous blocks of the batch per rank, one all_gather."""
    script = tmp_path / "worker2.py"
    script.write_text(WORKER2)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err[-3000:]
    r0, r1 = np.load(tmp_path / "sharded_0.npz"), np.load(tmp_path / "sharded_1.npz")
    for B in (1, 2, 5, 64, 67):
        assert np.array_equal(r0["B%d" % B], r0["ref%d" % B]) and np.array_equal(r1["B%d" % B], r0["ref%d" % B])
        per = (B + 1) // 2
        assert r0["n%d" % B].sum() == per and r1["n%d" % B].sum() == B - per      # each rank evaluated only its block


WORKER3 = textwrap.dedent('''
    import sys, numpy as np, torch
    sys.path.insert(0, %r)
    from thepayne_amd import dist as pdist
    rank, world, _ = pdist.init_from_env("gloo")
    out = {}
    for B in (1, 3, 64, 67):
        theta = torch.as_tensor(np.random.default_rng(B).normal(size=(B, 4)))
        sb = pdist.ShardedBatch(B, rank, world, torch.device("cpu"))
        n = []
        def fn(lo, hi, dst):
            n.append(hi - lo)
            dst.copy_((theta[lo:hi] ** 2).sum(dim=1) - 1.0)
        for it in range(3):                                  # buffers are reused: three iterations of a sampler
            sb.step(fn)
        out["B%%d" %% B] = sb.result().numpy()
        out["ref%%d" %% B] = ((theta ** 2).sum(dim=1) - 1.0).numpy()
        out["n%%d" %% B] = np.array(n)
    np.savez(sys.argv[1] + "/sb_%%d.npz" %% rank, **out)
    pdist.finalize()
''') % ROOT


def test_sharded_batch_keeps_its_buffers_over_iterations(tmp_path):
    """bench.py --shard-batch / a sampler iteration over G ranks: ShardedBatch.step, three times, two gloo ranks."""
    script = tmp_path / "worker3.py"
    script.write_text(WORKER3)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0, err[-3000:]
    r0, r1 = np.load(tmp_path / "sb_0.npz"), np.load(tmp_path / "sb_1.npz")
    for B in (1, 3, 64, 67):
        assert np.array_equal(r0["B%d" % B], r0["ref%d" % B]) and np.array_equal(r1["B%d" % B], r0["ref%d" % B])
        per = (B + 1) // 2
        assert list(r0["n%d" % B]) == [per] * 3 and list(r1["n%d" % B]) == ([B - per] * 3 if B > per else [])
