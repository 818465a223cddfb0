#!/usr/bin/env python
"""Every collective the multi-GPU paths issue, run once through a process group of THIS launch's ranks and saved.

    python -m thepayne_amd.launch 1 python tools/collectives_check.py --backend nccl --out rccl.npz      (one GPU: RCCL, world size 1)
    python -m thepayne_amd.launch 2 python tools/collectives_check.py --backend gloo --out gloo.npz      (CPU)

What it calls is what an 8-GPU job calls -- dist.init_from_env (device_id on "nccl"), dist.gather_summaries (all_gather of fp64
rows), dist.ShardedBatch.step / result and dist.sharded_lnlike (all_gather_into_tensor), bench.py's summary gather + MAX
all_reduce + barrier, destroy_process_group -- on deterministic inputs, so that the RCCL run of a one-GPU box can be held
against the gloo run (tests/test_rccl_world1_gpu.py).  It proves that librccl loads and these calls work on this image with
device tensors; it cannot prove scaling.  Started as a fresh child by thepayne_amd.launch: no process that has touched a GPU
is re-executed.  A failing check exits non-zero.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", required=True, choices=["nccl", "gloo"])
    ap.add_argument("--out", required=True)
    ap.add_argument("--stars", type=int, default=5)
    ap.add_argument("--batch", type=int, default=37)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    from thepayne_amd import dist as pdist
    rank, world, local_rank = pdist.init_from_env(backend=a.backend, force=True)
    assert dist.is_initialized() and dist.get_backend() == a.backend and dist.get_world_size() == world
    dev = torch.device("cuda", local_rank) if a.backend == "nccl" else torch.device("cpu")
    L = 41                                                     # [logZ, logZerr, niter, ncall, eff, wall] + 7 x 5
    # -- star summaries (fit_stars' collective)
    summ = lambda i: np.sin(0.37 * (i + 1) * np.arange(1, L + 1)) * 1e3
    local = {i: summ(i) for i in pdist.shard(a.stars, rank, world)}
    table = pdist.gather_summaries(local, a.stars, rank, world, L)
    assert np.array_equal(table, np.array([summ(i) for i in range(a.stars)]))
    # -- one batch split over the ranks, kept-buffer form and one-shot form
    B = a.batch
    lnl_of = lambda lo, hi: -0.5 * (torch.arange(lo, hi, dtype=torch.float64, device=dev) ** 2) - 1.0 / 3.0
    sb = pdist.ShardedBatch(B, rank, world, dev)
    for _ in range(3):
        sb.step(lambda lo, hi, out: out.copy_(lnl_of(lo, hi)))
    kept = sb.result()
    assert kept.device.type == dev.type and kept.dtype == torch.float64
    theta = np.arange(B, dtype=np.float64)[:, None] * np.ones((1, 3))
    per = (B + world - 1) // world
    lo0 = min(B, rank * per)
    once = pdist.sharded_lnlike(lambda blk: lnl_of(lo0, lo0 + len(blk)), theta, rank, world)
    full = lnl_of(0, B).cpu().numpy()
    assert np.array_equal(kept.cpu().numpy(), full) and np.array_equal(once, full)
    # -- bench.py's collectives: the MAX of the ranks' times, the barrier, the gather of per-star summaries
    tmax = torch.tensor([1.0 + rank], dtype=torch.float64, device=dev)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    assert float(tmax.item()) == float(world)
    dist.barrier()
    mine = torch.tensor([3.25 * (rank + 1), -7.0, float(rank)], dtype=torch.float64, device=dev)
    bufs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine)
    got = torch.stack(bufs).cpu().numpy()
    assert sorted(int(r) for r in got[:, 2]) == list(range(world))
    if rank == 0:
        np.savez(a.out, table=table, kept=kept.cpu().numpy(), once=once, bench=got[np.argsort(got[:, 2])], tmax=tmax.cpu().numpy(),
                 world=np.array(world), backend=np.array(a.backend), device=np.array(str(dev)))
        print("collectives ok: backend %s, world %d, device %s" % (a.backend, world, dev), flush=True)
    pdist.finalize()
    assert not dist.is_initialized()


if __name__ == "__main__":
    main()
