"""Multi-GPU: independent stars, one process per GPU, one collective.

The likelihood of one star needs no exchange between GPUs, so a batch of stars shards
embarrassingly: rank r fits stars r, r+W, r+2W, ... on its own MI355X with the ANN
weights replicated (5-80 MB).  The only collective is the gather of each star's
fixed-length posterior summary (~50 doubles) at the end, over RCCL/xGMI through
``torch.distributed`` (backend "nccl" on ROCm; "gloo" on CPU for the tests).  The
payload is latency-bound (< 1 KiB per star), so there is nothing to bucket or overlap.
"""
import os

import numpy as np

__all__ = ["init_from_env", "shard", "gather_summaries", "fit_stars", "sharded_lnlike", "ShardedBatch", "finalize"]


def _group_up():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def init_from_env(backend=None, force=False):
    """Initialise torch.distributed from the torchrun environment.  Returns
    (rank, world, local_rank); a no-op single-process setup when WORLD_SIZE is unset/1 -- unless `force` (or
    PAYNE_DIST_FORCE=1) asks for a process group of ONE rank: every collective below then runs through the backend
    (RCCL on a GPU) instead of being skipped, which is how a one-GPU box exercises the code an 8-GPU node will run."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force = force or os.environ.get("PAYNE_DIST_FORCE", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        if "MASTER_PORT" not in os.environ:        # (a forced group of one rank: any free port; torchrun sets it for real groups)
            from .launch import free_port
            os.environ["MASTER_PORT"] = str(free_port()) if world == 1 else "29511"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        if backend is None:
            backend = "nccl" if ndev > 0 else "gloo"
            if ndev > 0 and local_world > ndev:
                # more ranks than GPUs (two stars per GPU keep two independent batches in flight: 13.6 M against
                # 10.6 M evaluations/s per MI355X): RCCL refuses two ranks on one device ("Duplicate GPU detected"),
                # and the only collective is a < 1 KiB gather -- it goes over gloo, the GPUs stay compute-only
                backend = "gloo"
        kw = {}
        if ndev > 0:
            local_rank = local_rank % ndev
            torch.cuda.set_device(local_rank)
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    elif torch.cuda.is_available():
        local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
    return rank, world, local_rank


def shard(n_items, rank, world):
    """Indices of the items rank owns (round-robin: item i -> rank i % world)."""
    return list(range(rank, n_items, world))


def gather_summaries(local, n_total, rank, world, length):
    """local: {star index: summary[length]} of this rank -> [n_total, length] on every rank."""
    import torch
    import torch.distributed as dist
    out = np.full((n_total, length), np.nan)
    for i, s in local.items():
        out[i] = s
    if world == 1 and not _group_up():
        return out
    per = (n_total + world - 1) // world
    use_cuda = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    mine = torch.full((per, length + 1), float("nan"), dtype=torch.float64, device=dev)
    for j, i in enumerate(sorted(local)):
        mine[j, 0] = float(i)
        mine[j, 1:] = torch.as_tensor(np.asarray(local[i], dtype=np.float64), device=dev)
    bufs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine)                      # the one collective of the job
    for b in bufs:
        b = b.cpu().numpy()
        for row in b:
            if np.isfinite(row[0]):
                out[int(row[0])] = row[1:]
    return out


def fit_stars(stars, fit_fn, summary_length, backend=None):
    """Fit ``stars`` (any list of per-star inputs) sharded over the ranks; ``fit_fn(star,
    index)`` returns that star's summary vector.  Every rank gets the full table."""
    rank, world, _ = init_from_env(backend)
    local = {i: np.asarray(fit_fn(stars[i], i), dtype=np.float64) for i in shard(len(stars), rank, world)}
    return gather_summaries(local, len(stars), rank, world, summary_length)


def sharded_lnlike(lnlike_fn, theta, rank, world):
    """One star, one batch, G GPUs (pays only when a batch takes much longer than the ~20 us
    of the exchange, i.e. for 65k-pixel spectra): rank r evaluates the contiguous block r of the B candidate
    vectors with ``lnlike_fn(block) -> lnL[block]`` (numpy or a torch tensor on the rank's device) and one
    ``all_gather`` of <= ceil(B/G) doubles per rank rebuilds lnL[B] on every rank.  The candidates themselves
    are replicated (same proposals on every rank: shared RNG seed), so nothing else travels."""
    import torch
    import torch.distributed as dist
    theta = np.asarray(theta) if not hasattr(theta, "shape") else theta
    B = theta.shape[0]
    if world == 1 and not _group_up():
        out = lnlike_fn(theta)
        return out.cpu().numpy() if hasattr(out, "cpu") else np.asarray(out, dtype=np.float64)
    per = (B + world - 1) // world
    lo, hi = min(B, rank * per), min(B, (rank + 1) * per)
    use_cuda = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    mine = torch.full((per,), float("nan"), dtype=torch.float64, device=dev)
    if hi > lo:
        part = lnlike_fn(theta[lo:hi])
        part = part if isinstance(part, torch.Tensor) else torch.as_tensor(np.asarray(part, dtype=np.float64))
        mine[:hi - lo] = part.to(device=dev, dtype=torch.float64)
    full = torch.empty(per * world, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(full, mine)
    full = full.cpu().numpy().reshape(world, per)
    return np.concatenate([full[r, :max(0, min(B, (r + 1) * per) - min(B, r * per))] for r in range(world)])


class ShardedBatch(object):
    """The same exchange with its buffers kept (what a sampler iteration or `bench.py --shard-batch` repeats): one batch of B
    candidates, rank r owns the contiguous block [lo, hi), `step(fn)` has `fn(lo, hi, out)` write the block's B/G
    log-likelihoods into `out` (a float64 view on `device`) and rebuilds all B on every rank with ONE
    all_gather_into_tensor (<= ceil(B/G) doubles per rank; RCCL over xGMI on GPUs, gloo on CPU)."""

    def __init__(self, B, rank, world, device):
        import torch
        self.B, self.rank, self.world = int(B), int(rank), int(world)
        self.per = (self.B + self.world - 1) // self.world
        self.lo, self.hi = min(self.B, self.rank * self.per), min(self.B, (self.rank + 1) * self.per)
        self.mine = torch.full((self.per,), float("nan"), dtype=torch.float64, device=device)
        self.full = torch.empty(self.per * self.world, dtype=torch.float64, device=device)

    def step(self, fn):
        import torch.distributed as dist
        if self.hi > self.lo:
            fn(self.lo, self.hi, self.mine[:self.hi - self.lo])
        if self.world > 1 or _group_up():
            dist.all_gather_into_tensor(self.full, self.mine)
        else:
            self.full.copy_(self.mine)

    def result(self):
        """lnL[B] (the padding of the last blocks dropped), on the device of the buffers."""
        import torch
        per, B = self.per, self.B
        return torch.cat([self.full[r * per:r * per + max(0, min(B, (r + 1) * per) - min(B, r * per))] for r in range(self.world)])


def finalize():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
