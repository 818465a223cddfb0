#!/usr/bin/env python
"""Headline benchmark: likelihood evaluations / second (BASELINE.json).

Workload C2: one star, 4096-pixel 2x300 YST1 ANN, 3600 observed
pixels, a batch of 512 candidate parameter vectors (the sampler's live points) per
step.  A step = one pass of the hot path (ANN forward -> vsini/Doppler/instrument
smoothing -> chi^2) over the batch, theta already resident in HBM.  With --gpus N
every rank fits its own synthetic star (weak scaling, no data-path collective);
posterior-style summaries are all-gathered over RCCL once after the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C3|C5|small] [--shard-batch]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` without a torchrun environment starts the N ranks itself (fresh child
processes, before this process touches a GPU).  Prints ONE JSON line (rank 0).

The headline is C2.  On one GPU the same line also carries, under `also_measured`, short runs of C3 (C2 + photometry in
seven filters, the joint likelihood of SURVEY 8(d)) and C5 (65 536 pixels, 2048 candidates: the HBM-bound regime), each
with its own kernel times and roofline block.  `--shard-batch` (N > 1) is SURVEY 8(e)-2: ONE star, every batch split
in contiguous blocks over the ranks, the B log-likelihoods rebuilt on every rank by one all_gather per step
(strong scaling; pays for C5-class batches).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_FP32_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32 MFMA = packed-fp32 vector peak (spec)
PEAK_HBM_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the short nested-sampling run behind `end_to_end`")
    ap.add_argument("--unchecked", action="store_true", help="timing experiments with a deliberately incomplete library "
                    "(tools/exp/ablate.py): skip the result checks and mark the line invalid")
    ap.add_argument("--variant", type=int, default=0, help="payne_opts.variant (kernel variants of include/payne_hip.h; A/B runs)")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent batches in flight (one engine + HIP stream each); the headline uses 1")
    ap.add_argument("--shard-batch", action="store_true",
                    help="N > 1: one star, each batch split over the ranks, one all_gather of lnL per step (SURVEY 8(e)-2)")
    ap.add_argument("--no-also", action="store_true", help="skip the `also_measured` runs (C3, C5) behind the C2 headline")
    ap.add_argument("--e2e-calls", type=int, default=700000,
                    help="likelihood calls per end-to-end sampler run (three runs; a C2 run converges -- nothing left of the evidence at "
                         "dlogz = 1e-9 -- after ~727 k calls, so longer runs do not exist)")
    ap.add_argument("--repeats", type=int, default=15,
                    help="the block of --steps steps is timed this many times back to back (each between barriers); the MEDIAN "
                         "block is the headline, min / max ride along: a 20-step block is 0.8 ms, one clock ramp moved it 3 %%")
    ap.add_argument("--min-timed-ms", type=float, default=120.0,
                    help="keep timing blocks (beyond --repeats) until this much timed work has been done: an MI355X that was idle takes "
                         "15-20 ms of work to reach its clocks (tools/exp/block_ramp.py: a 20-step block 30.7 us a step in the first 9 ms, "
                         "28.9 from 20 ms on), and --steps 20 --repeats 15 is 9 ms in all; 0: exactly --repeats blocks")
    ap.add_argument("--force-dist", action="store_true",
                    help="one rank, but through a torch.distributed process group (RCCL, world size 1): the collectives of "
                         "the N > 1 path on a one-GPU box (tests/test_rccl_world1_gpu.py)")
    return ap.parse_args()


# ----------------------------------------------------------------------------
# N ranks from one command line (thepayne_amd/launch.py).  The parent never imports torch or touches HIP: it builds
# the library once, starts N copies of this script with the torchrun environment and relays rank 0's line.
# ----------------------------------------------------------------------------
def launch_ranks(n):
    from thepayne_amd.launch import launch_ranks as _launch
    from thepayne_amd.build import build_lib
    return _launch(n, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], prepare=build_lib)


# ----------------------------------------------------------------------------
# CPU baseline: the oracle in the reference's calling pattern (one theta per
# lnprobfn call, fp64 numpy), one process per core.  Runs BEFORE the GPU is touched.
# ----------------------------------------------------------------------------
def _cpu_worker(args):
    cfg_name, budgets, seed = args
    import numpy as np
    import oracle as O
    from thepayne_amd import synth
    cfg = synth.CONFIGS[cfg_name]
    net = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], seed=0)
    obs = synth.obs_grid(net["wavelength"], cfg["nobs"])
    T = synth.TRUTH
    _, clean = O.getspec(net, Teff=T["Teff"], logg=T["logg"], feh=T["feh"], afe=T["afe"], rad_vel=T["vrad"],
                         rot_vel=T["vrot"], vmic=np.nan, inst_R=2.355 * T["inst_R"], outwave=obs)
    flux = clean + np.random.default_rng(0).normal(0, 0.01, len(obs))
    names = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Inst_R']
    th = synth.draw_candidates(4096, seed=100 + seed)
    kw = {}
    if cfg.get("phot"):                       # C3: the joint likelihood (+ seven filters, log(A) parametrisation)
        phot = synth.make_phot_nets()
        from thepayne_amd.engine import highav_coefficients
        phot["hiav"] = highav_coefficients(phot["filters"])
        names = names + ['log(A)', 'Av']
        th = synth.draw_candidates_c3(4096, seed=100 + seed)
        kw = dict(phot=phot, obs_phot=synth.c3_obs_phot(phot["filters"]), photscale=True)
    out = []
    for loop, budget in zip((True, False), budgets):   # the reference's per-pixel chi^2 loop (likelihood.py:95-97), then vectorised
        if budget <= 0:
            out.append((0, 1.0))
            continue
        L = O.OracleLikelihood(net, obs, flux, np.full(len(obs), 0.01), names, pixel_loop=loop, **kw)
        O.lnprobfn(th[0], L)                  # warm
        n, t0 = 0, time.perf_counter()
        while True:                           # time-boxed: the sample is whatever fits the budget
            O.lnprobfn(th[n % len(th)], L)
            n += 1
            dt = time.perf_counter() - t0
            if dt >= budget:
                break
        out.append((n, dt))
    return out


def usable_cores():
    """CPUs this process may really use: affinity mask capped by the cgroup quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()
            if q != "max":
                n = min(n, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(cfg_name, budgets=(12.0, 6.0), max_procs=32):
    """budgets: seconds of the SURVEY 8(d) leg (chi^2 by the reference's per-pixel loop) and of the vectorised-chi^2 leg."""
    import multiprocessing as mp
    cores = min(usable_cores(), max_procs)
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = "1"                   # one thread per process, like the reference's einsum path
    ctx = mp.get_context("spawn")             # never fork a process that may hold a GPU
    with ctx.Pool(cores) as pool:
        t0 = time.perf_counter()
        res = pool.map(_cpu_worker, [(cfg_name, tuple(budgets), i) for i in range(cores)])
        wall = time.perf_counter() - t0
    loop = [r[0][0] / r[0][1] for r in res]
    vect = [r[1][0] / r[1][1] for r in res]
    return dict(value=float(sum(loop)), unit="likelihood-evals/s", cores=cores, kind="port",
                per_core=float(sum(loop) / cores),
                **({"vectorised_chi2": {"value": float(sum(vect)), "per_core": float(sum(vect) / cores),
                                        "what": "the same port with chi^2 as one numpy expression instead of the reference's "
                                                "per-pixel Python loop"}} if budgets[1] > 0 else {}),
                sample="numpy oracle lnprobfn, one theta per call, fp64, chi^2 by the per-pixel Python loop of "
                       "Payne/fitting/likelihood.py:95-97, %s workload: %d processes x %.0f s (%d calls; then %.0f s "
                       "with vectorised chi^2, %d calls; %.1f s wall incl. start-up)"
                       % (cfg_name, cores, budgets[0], sum(r[0][0] for r in res), budgets[1],
                          sum(r[1][0] for r in res), wall),
                reference_measured_elsewhere="251 evals/s/core (C2), 212 (C3, joint), 11.6 (C5): the reference itself on the "
                                             "survey container's CPU (BASELINE.md)")


# ----------------------------------------------------------------------------
# HBM traffic from rocprofv3 PMC passes.  Counters cannot be read from inside this process, so
# they come from the committed passes of this very command (tools/gpu_profiles.sh) -- and only
# when those were taken with the build that is running now (source hash recorded beside them).
# ----------------------------------------------------------------------------
def pmc_file(cfg_name):
    import glob
    from thepayne_amd.build import source_hash
    h = source_hash()
    for meta in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s_pmc_meta.json" % cfg_name.lower())), reverse=True):
        try:
            m = json.load(open(meta))
        except (OSError, ValueError):
            continue
        if m.get("source_hash") == h:
            return os.path.join(ROOT, "profiles", m["hbm_csv"]), m
    return None, {"source_hash": h}


def pmc_bytes(path, keys):
    """2 x FETCH_SIZE + WRITE_SIZE (KB, separate passes; FETCH doubled as MI355X_MICROARCH.md prescribes for
    16-B/lane streaming reads on gfx950) per launch of the kernels whose name contains one of `keys` (the one launched most
    often: a context's one-off launches -- e.g. the single payne_dense_dma3 launch of a C5 context's set-up beside the
    payne_dense_big3_kernel of every step -- do not count)."""
    import csv
    tot, seen = {}, {}
    for r in csv.DictReader(open(path)):
        if any(k in r["kernel"] for k in keys) and int(r["launches"]) >= seen.get(r["counter"], 0):     # the steady-state variant
            tot[r["counter"]] = float(r["mean_value_per_launch_KB_raw"])
            seen[r["counter"]] = int(r["launches"])
    if "FETCH_SIZE" not in tot or "WRITE_SIZE" not in tot:
        return None
    return (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0


KERNEL_KEYS = {"dense_out": ("payne_dense_dma", "payne_dense_big3"), "post": ("payne_post",), "dense_hidden": ("payne_dense_hidden_kernel",),
               "sed": ("payne_sed",)}
METRIC_C2 = "likelihood-evals/sec (4k-pixel ANN, 512 live points)"


# ----------------------------------------------------------------------------
# One configuration on this rank's GPU: problem, timed steps, per-kernel device times.
# ----------------------------------------------------------------------------
def make_problem(cfg_name, B, rank, local_rank, variant=0, streams=1):
    """Engines (one per batch in flight) + resident theta / lnL for config `cfg_name`; star seed = rank."""
    import numpy as np
    import torch
    from thepayne_amd import nnio, synth
    from thepayne_amd.engine import PayneEngine
    cfg = dict(synth.CONFIGS["C2" if cfg_name == "LinNet300" else cfg_name])
    if cfg_name == "LinNet300":               # FitPayne's default network (fitstar.py:81) as the reference defines it, on the C2 shape
        raw = synth.make_torch_net("LinNet", npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=(300, 300, 300), seed=21)
        net = nnio.normalize_spec_net(raw, "LinNet")
    else:
        raw = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], seed=0)
        net = nnio.normalize_spec_net(raw)
    obs = synth.obs_grid(raw["wavelength"], cfg["nobs"])
    eng0 = PayneEngine(net, obs=(obs,), b_max=1, device=local_rank)
    T = synth.TRUTH
    rng = np.random.default_rng(rank)
    truth = np.full((1, eng0.ncols), np.nan)
    truth[0, :8] = [T["Teff"] + 20.0 * rank, T["logg"], T["feh"], T["afe"], T["vrad"], T["vrot"], np.nan, T["inst_R"]]
    clean = eng0.predict_batch(truth, stage=2, fwhm_R=True).cpu().numpy()[0].astype(np.float64)
    flux = clean + rng.normal(0, 0.01, len(obs))
    eflux = np.full(len(obs), 0.01)
    eng0.close()
    kw = {}
    phot = None
    if cfg.get("phot"):                       # C3: + seven filters, photscale parametrisation
        phot = synth.make_phot_nets()
        kw = dict(phot=phot, obs_phot=synth.c3_obs_phot(phot["filters"]), photscale=True)
    S = max(1, streams)
    engines = [PayneEngine(net, obs=(obs, flux, eflux), b_max=B, device=local_rank, variant=(variant if j == 0 else 0), **kw)
               for j in range(S)]
    eng = engines[0]
    theta = eng.make_theta(B)
    if phot is None:
        th = synth.draw_candidates(B, seed=1 + rank)
    else:
        th = synth.draw_candidates_c3(B, seed=1 + rank)
        theta[:, eng.phot_off] = torch.as_tensor(th[:, 7], device=theta.device)          # log(A)
        theta[:, eng.phot_off + 2] = torch.as_tensor(th[:, 8], device=theta.device)      # Av
    theta[:, 0:6] = torch.as_tensor(th[:, 0:6], device=theta.device)
    theta[:, 7] = torch.as_tensor(th[:, 6], device=theta.device)
    lnl = torch.empty(B, dtype=torch.float64, device=theta.device)
    D, H, N = net["layers"][0][0].shape[1], net["layers"][0][0].shape[0], cfg["npix"]
    dims = dict(D=D, H=H, N=N, nobs=cfg["nobs"], B=B, n1=1 << int(np.ceil(np.log2(N))), n_hh=len(net["layers"]) - 2, kind=net["kind"],
                F=(len(phot["filters"]) if phot else 0), HP=(phot["w1"].shape[1] if phot else 0))
    return dict(cfg=cfg, engines=engines, theta=theta, lnl=lnl, dims=dims)


def alg_work(d):
    """algorithmic FLOPs per evaluation and compulsory HBM bytes per batch (+ the photometric nets of C3:
    F x (6 -> HP -> HP -> 1), fp64 arithmetic on fp32 weights)."""
    import numpy as np
    D, H, N, B, F, HP = d["D"], d["H"], d["N"], d["B"], d["F"], d["HP"]
    nhh = d.get("n_hh", 1)                    # H x H layers: 1 (YST1), 4 (LinNet), 2 (SMLP)
    L2N = np.log2(N)
    fl_post_alg = 2 * 2 * 2.5 * N * L2N + 60.0 * N          # the algorithm: two convolution stages, two transforms each
    # what the post kernel itself executes: three transforms when the output layer hands over rows already transformed (its
    # weights carry the first stage's forward transform at no extra product: payne_last_kernel kind 4 == "frequency")
    fl_post = d.get("post_transforms", 4) * 2.5 * N * L2N + 60.0 * N
    fl_sed = F * 2.0 * (6 * HP + HP * HP + HP)
    flops_eval = 2.0 * (D * H + nhh * H * H + H * N) + fl_post_alg + fl_sed
    bytes_batch = 4.0 * (D * H + H + nhh * (H * H + H) + H * N + N) + 8.0 * N + 16.0 * d["nobs"] + B * (8.0 * 12 + 4) \
        + 4.0 * F * (6 * HP + HP + HP * HP + HP + HP + 1)
    return dict(flops_eval=flops_eval, bytes_batch=bytes_batch, flops_post=fl_post, flops_post_alg=fl_post_alg, flops_sed=fl_sed,
                flops_out=2.0 * H * N)


def more_blocks(blocks, repeats, min_timed_s, cap=4000):
    """Is another timed block due?  At least `repeats` blocks, then blocks until `min_timed_s` seconds of timed work are behind (`blocks`:
    the MAX-over-ranks times so far, the same list on every rank, so every rank stops at the same block); never more than `cap`."""
    return len(blocks) < max(1, repeats) or (sum(blocks) < min_timed_s and len(blocks) < cap)


def run_config(cfg_name, args, steps, warmup, rank, world, local_rank, B=0, streams=1, shard=False, repeats=1, min_timed_s=0.0):
    """Time `steps` steps of config `cfg_name` (after `warmup`) between barriers; MAX over ranks.  Returns the numbers the
    JSON line is made of.  shard: one star, the batch split over the ranks, one all_gather of lnL per step."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from thepayne_amd import synth
    grouped = dist.is_available() and dist.is_initialized()      # (--force-dist: a group of one rank still runs its collectives)
    B = B or synth.CONFIGS["C2" if cfg_name == "LinNet300" else cfg_name]["batch"]
    star = 0 if shard else rank
    P = make_problem(cfg_name, B, star, local_rank, variant=args.variant, streams=streams)
    engines, theta, lnl = P["engines"], P["theta"], P["lnl"]
    eng = engines[0]
    S = len(engines)
    sts = [torch.cuda.Stream(device=local_rank) for _ in range(S)] if S > 1 else [torch.cuda.current_stream()]
    thetas = [theta] + [theta.clone() for _ in range(S - 1)]
    lnls = [lnl] + [torch.empty_like(lnl) for _ in range(S - 1)]
    if shard and world > 1:
        from thepayne_amd.dist import ShardedBatch
        sb = ShardedBatch(B, rank, world, theta.device)
        th_blk = theta[sb.lo:sb.hi].contiguous()
    torch.cuda.synchronize()

    def step(i):
        if shard and world > 1:
            sb.step(lambda lo, hi, out: eng.lnlike_batch(th_blk, out=out))   # the per-iteration exchange: <= ceil(B/G) doubles per rank
            return
        j = i % S
        if S == 1:
            eng.lnlike_batch(theta, out=lnl)
        else:
            with torch.cuda.stream(sts[j]):
                engines[j].lnlike_batch(thetas[j], out=lnls[j])

    def barrier():
        torch.cuda.synchronize()
        if world > 1 or grouped:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        step(i)
    # `repeats` blocks of EXACTLY `steps` steps, each between barrier + synchronize on both sides, each block's time the MAX over
    # ranks; the median block is the result (the first block after a warm-up of five steps rides a clock ramp: min / max say so)
    # Each block is timed twice: by a HIP event pair on the launch stream around its `steps` steps (what the device
    # spent on them -- `value`), and by the host clock between the two barriers (`ms_per_step_wall`: + the synchronize / barrier
    # edges of a block, ~55 us whatever its length, i.e. 2.8 us per step of a 20-step block and 0.3 of a 200-step one).
    # Batches in flight on several streams have no single launch stream: their blocks are timed by the host clock alone.
    use_events = (S == 1)
    blocks, blocks_local, blocks_wall = [], [], []
    # At least `repeats` blocks, and blocks until `min_timed_s` of timed work is behind (every rank sees the same MAX-reduced times, so
    # every rank stops at the same block): a GPU that was idle is 6 % slower for its first 15-20 ms of work (clock ramp), which is the
    # whole of fifteen 20-step blocks; the median is then taken over blocks most of which ran at the clocks the job runs at.
    while more_blocks(blocks, repeats, min_timed_s):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        if use_events:
            ev0.record(sts[0])
        for i in range(steps):
            step(i)
        if use_events:
            ev1.record(sts[0])
        barrier()
        dw = time.perf_counter() - t0
        dl = 1e-3 * ev0.elapsed_time(ev1) if use_events else dw
        tmax = torch.tensor([dl, dw], dtype=torch.float64, device=theta.device)
        if world > 1 or grouped:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        blocks.append(float(tmax[0].item()))
        blocks_wall.append(float(tmax[1].item()))
        blocks_local.append(dl)
    order = sorted(range(len(blocks)), key=lambda j: blocks[j])
    mid = order[len(order) // 2]
    dt, dt_local = blocks[mid], blocks_local[mid]
    dt_wall = sorted(blocks_wall)[len(blocks_wall) // 2]
    if shard and world > 1:
        lnl = sb.result()
    for l_ in lnls[1:]:                  # every in-flight batch evaluated the same candidates: same answers
        assert args.unchecked or bool(torch.equal(torch.nan_to_num(l_), torch.nan_to_num(lnl)))
    assert args.unchecked or int(torch.isfinite(lnl).sum()) >= B - max(4, B // 128), \
        "non-finite lnL in the benchmark batch"   # Inst_R tail draws are NaN by contract

    kern = None
    if not args.no_kernel_timing and not (shard and world > 1):
        # per-kernel device time (HIP events attached to the launches, on the launch stream), same steps replayed
        eng.profile(True)
        for _ in range(max(steps, 100) if min_timed_s > 0 and dt / steps < 2e-4 else steps):   # (short steps: at least a hundred launches a kernel)
            eng.lnlike_batch(theta, out=lnl)
        torch.cuda.synchronize()
        kern = eng.profile_read()
        eng.profile(False)
    res = dict(P, dt=dt, dt_local=dt_local, dt_wall=dt_wall, timed_by=("hip-events" if use_events else "host-clock"), blocks=blocks, steps=steps, warmup=warmup, B=B, S=S, kern=kern, lnl=lnl, shard=bool(shard and world > 1),
               evals=(B * steps if shard else world * B * steps))
    return res


def roofline_blocks(cfg_name, res, args):
    """roofline / mfma_kernel / kernels_us / whole_step / hbm blocks of one configuration's run."""
    d, kern = res["dims"], res["kern"]
    B, H, N, n1 = d["B"], d["H"], d["N"], d["n1"]
    rows = res["engines"][0].kernels_used().get("rows", "pixels") if res.get("engines") else "pixels"
    d = dict(d, post_transforms=3 if rows == "frequency" else 4)
    W = alg_work(d)
    ms_per_step = 1e3 * res["dt"] / res["steps"]
    out = {}
    per = {k: (1e3 * v[0] / v[1] if v[1] else 0.0) for k, v in kern.items()}      # us per launch
    dom = max(("dense_out", "post"), key=lambda k: per[k])
    csv_path, meta = pmc_file(cfg_name)
    traffic = {k: (pmc_bytes(csv_path, key) if csv_path else None) for k, key in KERNEL_KEYS.items()}
    if not d["F"] or not per.get("sed"):
        traffic.pop("sed", None)
    tsrc = ("%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, build %s = the running one)"
            % (os.path.relpath(csv_path, ROOT), meta["source_hash"])) if csv_path else \
           ("none: no PMC passes under profiles/ were taken with the running build (%s)" % meta["source_hash"])
    big = n1 > 16384
    if big and dom == "post":
        # Spectra larger than LDS.  What binds the kernel depends on its form: payne_post_chip_kernel / payne_post_chip32_kernel keep
        # a convolution stage on the compute unit (5 transfers of the spectrum): their arithmetic / LDS phases and their memory phases
        # alternate on a CU -> `frac` is the algorithmic FLOP rate against the fp32 vector peak; payne_post_big_kernel streams the spectrum through a
        # global workspace 25 times and is HBM/L2-bound -> `frac` is SURVEY 8(d)'s modelled streaming bytes over the measured time.
        # Both figures and the counter rate are printed for either kernel.
        transfers = eng_round_trips(n1, args.variant)
        alg8 = (W["bytes_batch"] / B + 8 * 4.0 * N) * B
        t = max(per[dom], 1e-9) * 1e-6
        ach = alg8 / t / 1e9
        kname = res["engines"][0].kernels_used()["post"] if res.get("engines") else ""
        chip = "chip" in kname
        fl = B * W["flops_post"]
        tf = fl / t / 1e12
        extra = {"traffic": traffic["post"], "traffic_source": tsrc, "avg_us_per_launch": per[dom],
                 "alg_flops_per_launch": fl, "fp32_vector_tflops": tf, "fp32_vector_frac": tf / PEAK_FP32_TFLOPS,
                 "alg_bytes_per_launch": alg8, "modelled_hbm_GBs": ach, "modelled_hbm_frac": ach / PEAK_HBM_GBS,
                 "alg_basis": "SURVEY 8(d): compulsory %.1f KB + 8 spectrum passes x 4N = %.2f MB per evaluation "
                              "(modelled bytes over the measured kernel time)" % (W["bytes_batch"] / B / 1e3, alg8 / B / 1e6),
                 "passes_used": transfers, "workspace_bytes_per_launch": transfers * 4.0 * n1 * B,
                 "workspace_rate_GBs": transfers * 4.0 * n1 * B / t / 1e9}
        if traffic["post"]:
            extra["counter_rate_GBs"] = traffic["post"] / t / 1e9
            extra["hbm_frac_counters"] = traffic["post"] / t / 1e9 / PEAK_HBM_GBS
        if chip:
            out["roofline"] = dict({"bound": "fp32-vector", "kernel": kname, "achieved": tf, "peak": PEAK_FP32_TFLOPS,
                                    "unit": "TFLOP/s", "frac": tf / PEAK_FP32_TFLOPS,
                                    "note": "convolution stages on the compute unit (5 transfers of the spectrum instead of 25).  By ablation "
                                            "(tools/exp/chip_ablate.py, NOTES R4.12) 41 % of the kernel is the stages' arithmetic and LDS exchanges, "
                                            "59 % their loads / stores / gather and the observed-grid loop at ~20 B/clk/CU (the L2-to-CU port's "
                                            "ceiling); one workgroup per CU: the two alternate, never overlap.  `frac` = algorithmic FLOPs over the "
                                            "fp32 vector peak; `hbm_frac_counters` = counters' bytes over the HBM peak; `modelled_hbm_frac` prices "
                                            "SURVEY 8(d)'s eight streaming passes, which this kernel does not make"}, **extra)
        else:
            out["roofline"] = dict({"bound": "hbm", "kernel": kname or "payne_post_big_kernel", "achieved": ach, "peak": PEAK_HBM_GBS,
                                    "unit": "GB/s", "frac": ach / PEAK_HBM_GBS}, **extra)
    else:
        flops = {"dense_out": B * W["flops_out"], "post": B * W["flops_post"]}
        ach = flops[dom] / (max(per[dom], 1e-9) * 1e-6) / 1e12
        out["roofline"] = {"bound": "mfma" if dom == "dense_out" else "fp32-vector",
                           "kernel": "payne_dense_dma_kernel (output layer)" if dom == "dense_out" else "payne_post_kernel",
                           "achieved": ach, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS,
                           "traffic": traffic[dom], "traffic_source": tsrc,
                           "alg_flops_per_launch": flops[dom], "avg_us_per_launch": per[dom]}
        out["rows_handed_to_post_kernel"] = rows
        if dom == "post":
            out["roofline"]["transforms_in_kernel"] = d["post_transforms"]
            if d["post_transforms"] != 4:
                # SURVEY 8(d)'s own figure for everything after the network (four transforms), over the same launch time: the
                # transform the output layer's weights carry is done, at no product of its own
                a8 = B * W["flops_post_alg"] / (max(per[dom], 1e-9) * 1e-6) / 1e12
                out["roofline"]["survey_8d"] = {"alg_flops_per_launch": B * W["flops_post_alg"], "achieved": a8, "frac": a8 / PEAK_FP32_TFLOPS,
                                                "what": "`achieved` / `frac` above count the three transforms this kernel executes; these count "
                                                        "the four of the algorithm it completes"}
            out["roofline"]["note"] = ("FFT / interpolation pipeline in LDS: no MFMA instructions; priced against the "
                                       "packed-fp32 vector peak (= the fp32 MFMA peak, 157.3 TFLOP/s). The MFMA kernel of "
                                       "the step is under `mfma_kernel`.")
        t_out = max(per["dense_out"], 1e-9) * 1e-6
        split = not (args.variant & (4096 | 1 | 1024))          # the default output layer: products of split operands
        kout = res["engines"][0].kernels_used().get("out", "") if res.get("engines") else ""
        pairs = split and ("2h" in kout or "big3_kernel<true>" in kout)       # two fp16 parts, three products (else three bf16 parts, six)
        nprod = 3.0 if pairs else 6.0
        out["mfma_kernel"] = {"kernel": ((kout or "payne_dense_dma3_kernel") + (" (output layer, 2 x fp16 split, 3 products)" if pairs else
                                                                              " (output layer, 3 x bf16 split, 6 products)")) if split else
                                        "payne_dense_dma_kernel (output layer, fp32 matrix instruction)",
                              "alg_flops_per_launch": flops["dense_out"],
                              "avg_us_per_launch": per["dense_out"],
                              "achieved_tflops": flops["dense_out"] / t_out / 1e12,
                              "frac_of_fp32_peak": flops["dense_out"] / t_out / 1e12 / PEAK_FP32_TFLOPS}
        if split:                                                # what the matrix pipe executes: `nprod` 16-bit flops per algorithmic flop
            out["mfma_kernel"].update({"executed_bf16_tflops": nprod * flops["dense_out"] / t_out / 1e12,
                                       "frac_of_bf16_peak": nprod * flops["dense_out"] / t_out / 1e12 / PEAK_BF16_TFLOPS,
                                       "note": "fp32-class products as %d exact partial products of 16-bit parts (fp16 pairs: three; bf16 triples: six), fp32 "
                                               "accumulator: `frac_of_fp32_peak` compares the algorithmic fp32 work with what the fp32 matrix instruction "
                                               "could do at best, `frac_of_bf16_peak` the executed 16-bit work with the dense fp16 / bf16 peak (2.5 PFLOP/s)"
                                               % int(nprod)})
    out["kernels_us"] = per
    out["alg_flops_per_eval"] = W["flops_eval"]
    t_k = max(sum(per.values()), 1e-9) * 1e-6
    out["whole_step"] = {"alg_flops": W["flops_eval"] * B,
                         "tflops_on_kernel_time": W["flops_eval"] * B / t_k / 1e12,
                         "tflops_on_wall_time": W["flops_eval"] * B / (1e-3 * ms_per_step) / 1e12,
                         "frac_of_fp32_peak_on_wall_time": W["flops_eval"] * B / (1e-3 * ms_per_step) / 1e12 / PEAK_FP32_TFLOPS}
    # whole-step HBM rate (the north star asks for the achieved HBM-bandwidth fraction)
    alg_step = W["bytes_batch"] + (0.0 if big else 2.0 * 4.0 * B * N)   # + spectra written by the output layer, read back once
    hbm = {"alg_bytes_per_step": alg_step, "alg_rate_GBs": alg_step / (1e-3 * ms_per_step) / 1e9,
           "peak": PEAK_HBM_GBS, "unit": "GB/s", "source": tsrc}
    if traffic and all(v is not None for v in traffic.values()):
        sb = float(sum(traffic.values()))
        hbm.update({"pmc_bytes_per_step": sb, "achieved": sb / (1e-3 * ms_per_step) / 1e9,
                    "frac": sb / (1e-3 * ms_per_step) / 1e9 / PEAK_HBM_GBS})
    out["hbm"] = hbm
    return out


def workload_text(cfg_name, d):
    net = {"YST1": "2x%d YST1" % d["H"], "LinNet": "5x%d LinNet (sigmoid; FitPayne's default NNtype)" % d["H"], "SMLP": "3x%d SMLP" % d["H"]}[d.get("kind", "YST1")]
    s = "%s: single star per GPU, %d-pixel %s ANN, %d observed pixels" % (cfg_name, d["N"], net, d["nobs"])
    if d["F"]:
        s += ", photometry in %d filters (6-%d-%d-1 nets, log(A) parametrisation)" % (d["F"], d["HP"], d["HP"])
    return s + ", batch of %d candidate vectors per step (dynesty live points)" % d["B"]


def end_to_end(cfg_name, B, value, calls, runs=3, cpu=None, fit=True):
    """the same likelihood as the batched nested sampler sees it -- prior transform, random-walk
    proposals (device), transfers and the dead-point bookkeeping included.  `runs` runs of `calls` likelihood calls
    each (different sampler seeds; dlogz small enough that none stops early); the median rate is quoted.
    Beside the calls/s: ITERATIONS/s (dead points: the unit of the reference's progress line, fitstar.py:337-338, 398-401) and
    calls per iteration, and -- `fit` -- ONE complete fit at the reference's defaults (delta_logz_final = 0.01, walks = 25,
    fitstar.py:262-271; the remaining live points added, :410) with its wall time, next to what the CPU port's measured rate makes
    of the same number of calls on the same cores."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sampler_bench
    both = [sampler_bench.run(cfg_name, maxcall=calls, nlive=B, walks=25, modes=("device_chunks", "device_chunks_hostturn"), seed=1 + i,
                              dlogz=1e-9) for i in range(runs)]
    rs = [r["device_chunks"] for r in both]
    rates = sorted(r["evals_per_s"] for r in rs)
    med = float(np.median(rates))
    hv = [r["device_chunks_hostturn"] for r in both]
    hv_rates = sorted(r["evals_per_s"] for r in hv)
    hv_med = float(np.median(hv_rates))
    n_calls, n_it, secs = int(sum(r["calls"] for r in rs)), int(sum(r["iterations"] for r in rs)), float(sum(r["seconds"] for r in rs))
    out = {"value": med, "unit": "likelihood-evals/s", "runs": runs, "rates": rates,
           "calls": n_calls, "iterations": n_it, "seconds": secs,
           "iterations_per_s": float(np.median([r["iterations"] / r["seconds"] for r in rs])),
           "calls_per_iteration": n_calls / max(1, n_it),
           # ONE denominator: this run's headline `value` (the median block of `steps` steps of the same process)
           "frac_of_kernel_only": med / value,
           "resyncs": int(sum(r.get("resyncs", 0) for r in rs)), "logz": [round(r["logz"], 3) for r in rs],
           "what": "static nested sampler, %d live points, rwalk x25 on the device, bound='multi' (ellipsoid decomposition "
                   "by recursive 2-means), dead points consumed in bulk, the sampler's default loop: the turn between two proposal "
                   "queues made on the device, the next queue enqueued before the current one has finished "
                   "(sampler['pipeline'] = 'device'; tests/test_sampler_gpu.py::test_turn_on_the_device_is_the_same_run_statistically); "
                   "median of %d runs.  A lock-step queue of K = nlive proposals is consumed under a rising threshold, so ~ln 2 of its "
                   "calls end as dead points' replacements: `calls_per_iteration` is walks / that fraction" % (B, runs),
           # the same runs with the turn made on the host and the next queue launched ahead of the bookkeeping (pipeline=True: the
           # default until round 5)
           "host_turn": {"value": hv_med, "rates": hv_rates, "frac_of_kernel_only": hv_med / value,
                         "logz": [round(r["logz"], 3) for r in hv]}}
    if fit:
        try:
            f = sampler_bench.run(cfg_name, maxcall=None, nlive=B, walks=25, modes=("device_chunks",), seed=11, dlogz=0.01,
                                  complete=True)["device_chunks"]
            blk = {"fit_wall_s": f["seconds"], "iterations": f["iterations"], "calls": f["calls"], "logz": round(f["logz"], 3),
                   "logzerr": round(f["logzerr"], 3), "dlogz": 0.01, "walks": 25, "nlive": B,
                   "iterations_per_s": f["iterations"] / max(f["seconds"], 1e-9),
                   "what": "one complete fit: delta_logz_final = 0.01 (fitstar.py:262-271), walks = 25, the remaining live points "
                           "added (fitstar.py:410); wall time of the sampling loop, set-up (network upload, first live points) excluded"}
            if cpu:
                blk["cpu_port_projected_s"] = f["calls"] / max(cpu["value"], 1e-9)
                blk["cpu_port_projection"] = ("the fit's %d calls / (%d cores x %.0f evals/s/core of `cpu_baseline`)"
                                              % (f["calls"], cpu["cores"], cpu["per_core"]))
            out["fit"] = blk
        except Exception as ex:                              # the headline must not die with a side run
            out["fit"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    return out


# ----------------------------------------------------------------------------
# The line the driver parses.  Everything measured goes to bench_detail.json beside this file (the full record: notes, kernel names,
# per-configuration roofline blocks); the LAST stdout line is the contract's fields only, numbers rounded, no prose, < 4 KB
# (tests/test_host_logic.py::test_bench_line_is_small_and_complete).
# ----------------------------------------------------------------------------
LINE_LIMIT = 4096
DETAIL_FILE = os.path.join(ROOT, "bench_detail.json")


def _r(x, sig=6):
    """Numbers at `sig` significant digits (the line carries measurements, not bit patterns)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float("%.*g" % (sig, float(x)))
    except (TypeError, ValueError):
        return x


def _pick(d, keys):
    return {k: _r(d[k]) for k in keys if k in d and d[k] is not None or k in d and k in ("traffic", "vs_baseline")}


def final_line(full):
    """The compact record of one bench run (a dict whose JSON is the last stdout line) from the full one."""
    out = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_first15", "ms_per_step_wall", "repeats",
                       "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "invalid"))
    c = full.get("config", {})
    out["config"] = {"workload": c.get("workload_short", c.get("workload", ""))[:160],
                     **_pick(c, ("batch", "npix", "nobs", "stars", "kernel_variant"))}
    if "parallelism" in c:
        out["config"]["parallelism"] = str(c["parallelism"])[:96]
    if "products" in c:
        out["config"]["products"] = str(c["products"])[:64]
    if full.get("n_gpus", 1) > 1 or full.get("collective_backend"):
        out["collective_backend"] = full.get("collective_backend")
        out["rccl_world"] = full.get("rccl_world")
        out["per_rank_evals_per_s"] = [_r(v, 5) for v in full.get("per_rank_evals_per_s", [])][:16]
    rf = full.get("roofline")
    if rf:
        out["roofline"] = _pick(rf, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_us_per_launch",
                                     "alg_flops_per_launch", "alg_bytes_per_launch", "hbm_frac_counters"))
        out["roofline"]["kernel"] = str(rf.get("kernel", ""))[:48]
        out["roofline"].setdefault("traffic", None)
    mk = full.get("mfma_kernel")
    if mk:
        out["mfma_kernel"] = {"kernel": str(mk.get("kernel", "")).split(" (")[0][:48], **_pick(mk, ("avg_us_per_launch", "achieved_tflops",
                                                                                              "frac_of_fp32_peak", "frac_of_bf16_peak"))}
    if "kernels_us" in full:
        out["kernels_us"] = {k: _r(v, 4) for k, v in full["kernels_us"].items() if v}
    ws = full.get("whole_step")
    if ws:
        out["whole_step_frac_of_fp32_peak"] = _r(ws.get("frac_of_fp32_peak_on_wall_time"), 4)
    hb = full.get("hbm")
    if hb:
        out["hbm"] = _pick(hb, ("achieved", "peak", "unit", "frac", "pmc_bytes_per_step", "alg_bytes_per_step"))
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "per_core", "kind"))
        out["cpu_baseline"]["sample"] = cb.get("sample_short", str(cb.get("sample", ""))[:120])
    ee = full.get("end_to_end")
    if ee:
        out["end_to_end"] = _pick(ee, ("value", "unit", "frac_of_kernel_only", "iterations_per_s", "calls_per_iteration", "calls",
                                       "iterations", "runs"))
        if "host_turn" in ee:
            out["end_to_end"]["host_turn"] = _r(ee["host_turn"].get("value"))
        if "fit" in ee:
            out["end_to_end"]["fit"] = _pick(ee["fit"], ("fit_wall_s", "iterations", "calls", "logz", "logzerr", "dlogz", "walks", "nlive",
                                                         "iterations_per_s", "cpu_port_projected_s", "error"))
    al = full.get("also_measured")
    if al:
        out["also_measured"] = {}
        for name, blk in al.items():
            if "error" in blk:
                out["also_measured"][name] = {"error": str(blk["error"])[:80]}
                continue
            e = {"value": _r(blk.get("value")), "ms_per_step": _r(blk.get("ms_per_step"), 5),
                 "roofline_frac": _r(blk.get("roofline", {}).get("frac"), 4)}
            if "kernels_us" in blk:
                e["kernels_us"] = {k: _r(v, 4) for k, v in blk["kernels_us"].items() if v}
            t = blk.get("roofline", {}).get("traffic")
            if t:
                e["traffic"] = _r(t, 4)
            if "end_to_end" in blk and "value" in blk["end_to_end"]:
                e["end_to_end"] = _r(blk["end_to_end"]["value"])
            if "cpu_baseline" in blk:
                e["cpu_per_core"] = _r(blk["cpu_baseline"].get("per_core"), 4)
            out["also_measured"][name] = e
    out["detail"] = os.path.basename(DETAIL_FILE)
    # the line must stay parsable whatever a run adds: the optional blocks shrink, then go, until it fits
    def slim():
        out["also_measured"] = {n: ({"value": b.get("value"), "ms_per_step": b.get("ms_per_step"), "roofline_frac": b.get("roofline_frac")}
                                    if "error" not in b else b) for n, b in out["also_measured"].items()}
    steps = [slim if "also_measured" in out else None] + [(lambda k=k: out.pop(k, None)) for k in
             ("kernels_us", "mfma_kernel", "hbm", "per_rank_evals_per_s", "also_measured")]
    for act in steps:
        if len(json.dumps(out)) < LINE_LIMIT:
            break
        if act is not None:
            act()
    return out


def emit(full):
    """Full record -> bench_detail.json; compact record -> the last stdout line."""
    try:
        with open(DETAIL_FILE, "w") as fh:
            json.dump(full, fh, indent=1)
    except OSError as ex:                                    # a read-only tree must not cost the line
        print("bench: could not write %s: %s" % (DETAIL_FILE, ex), file=sys.stderr)
    line = json.dumps(final_line(full))
    assert len(line) < LINE_LIMIT, len(line)
    print(line, flush=True)


# ----------------------------------------------------------------------------
def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    import numpy as np
    from thepayne_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    cpu = cpu_c3 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline("C2" if args.config == "LinNet300" else args.config)
        if args.config == "C2" and not args.no_also:
            cpu_c3 = cpu_baseline("C3", budgets=(6.0, 0.0))   # the joint likelihood's baseline (the reference itself: 212 evals/s/core)

    import torch
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = torch.cuda.device_count()
    if world > ndev:
        raise SystemExit("--gpus %d but this node shows %d GPU(s): one rank per GPU (RCCL refuses two ranks on a device)"
                         % (world, ndev))
    torch.cuda.set_device(local_rank)
    grouped = world > 1 or args.force_dist
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            from thepayne_amd.launch import free_port
            os.environ["MASTER_PORT"] = str(free_port()) if world == 1 else "29512"
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    res = run_config(args.config, args, args.steps, args.warmup, rank, world, local_rank, B=args.batch,
                     streams=args.streams, shard=args.shard_batch, repeats=args.repeats, min_timed_s=1e-3 * args.min_timed_ms)
    B, d, lnl, theta = res["B"], res["dims"], res["lnl"], res["theta"]

    # ---- the one collective of the multi-star job: gather per-star summaries (RCCL)
    summary = torch.stack([lnl.max(), lnl.mean(), lnl.std(), theta[lnl.argmax(), 0], theta[lnl.argmax(), 1],
                           torch.tensor(B * args.steps / res["dt_local"], dtype=torch.float64, device=theta.device),
                           torch.tensor(float(rank), dtype=torch.float64, device=theta.device)])
    gathered = [summary]
    if grouped:
        gathered = [torch.empty_like(summary) for _ in range(world)]
        dist.all_gather(gathered, summary)
    if rank != 0:
        for e in res["engines"]:
            e.close()
        dist.destroy_process_group()
        return
    table = torch.stack(gathered).cpu().numpy()
    assert sorted(int(r) for r in table[:, 6]) == list(range(world)), "the gather did not see every rank"

    out = {
        "metric": METRIC_C2 if args.config == "C2" else "likelihood-evals/sec (%s)" % args.config,
        "value": res["evals"] / res["dt"], "unit": "likelihood-evals/s",
        "n_gpus": dist.get_world_size() if grouped else 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * res["dt"] / args.steps,
        "repeats": len(res["blocks"]), "ms_per_step_min": 1e3 * min(res["blocks"]) / args.steps,
        "ms_per_step_max": 1e3 * max(res["blocks"]) / args.steps,
        "ms_per_step_first15": 1e3 * sorted(res["blocks"][:15])[len(res["blocks"][:15]) // 2] / args.steps,
        "ms_per_step_wall": 1e3 * res["dt_wall"] / args.steps,
        "timing": ("median of `repeats` blocks of `steps` steps (at least --repeats of them, and blocks until --min-timed-ms of timed work: "
                   "`ms_per_step_first15` is the median of the first fifteen, which an idle GPU's clock ramp covers when blocks are short), MAX over ranks; every block sits between barrier + synchronize on both "
                   "sides and is timed by a HIP event pair on the launch stream around its steps (`value`, `ms_per_step`) and by the host "
                   "clock between the barriers (`ms_per_step_wall`: + ~55 us of synchronize / barrier edges per block)")
                  if res["timed_by"] == "hip-events" else
                  "median of `repeats` blocks of `steps` steps, each block between barrier + synchronize (host clock), MAX over ranks",
        "higher_is_better": True, "scaling": "strong" if res["shard"] else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload_text(args.config, d),
                   "batch": B, "npix": d["N"], "nobs": d["nobs"], "stars": 1 if res["shard"] else world,
                   "batches_in_flight": res["S"], "kernel_variant": args.variant,
                   "products": "fp32 matrix instruction" if args.variant & 4096 else ("3 x bf16 parts, 6 products, fp32 accumulator" if args.variant & 1048576
                               else "2 x fp16 parts, 3 products, fp32 accumulator"),
                   "arithmetic": "fp32 storage and accumulation throughout (wavelengths, tapers, chi^2 sums, photometric nets fp64); "
                                 "the dense layers' products: " + ("v_mfma_f32_32x32x2_f32" if args.variant & 4096 else
                                 "operands split in two fp16 parts (power-of-two scales, 21-22 significant bits), three exact partial products, "
                                 "fp32 accumulator -- as accurate as the fp32 fma chain against an fp64 product (tests/test_gpu_parity.py); "
                                 "--variant 1048576 = three bf16 parts, six products; 4096 / 2097152 = the fp32 matrix instruction"),
                   "parallelism": ("1 star, every batch split in %d contiguous blocks, one all_gather of lnL per step (RCCL)" % world)
                                  if res["shard"] else "1 star per GPU, no data-path collective"},
        **({"invalid": "--unchecked: a timing experiment, not a benchmark result"} if args.unchecked else {}),
        "rccl_world": dist.get_world_size() if grouped else 1,
        "collective_backend": dist.get_backend() if grouped else None,
        "per_rank_evals_per_s": [float(v) for v in table[np.argsort(table[:, 6]), 5]],
    }
    if res["kern"] is not None:
        out.update(roofline_blocks(args.config, res, args))
    if cpu is not None:
        out["cpu_baseline"] = cpu
    for e in res["engines"]:
        e.close()
    res = None
    torch.cuda.empty_cache()

    if world == 1 and not args.no_also and args.config == "C2" and not args.batch and args.streams == 1:
        # the configurations the headline does not show: short runs, same code path, own kernel times and roofline blocks
        also = {}
        # C2r: C2 on a model grid that is NOT a power of two long (3600 pixels of the same lambda_0 and R, 3200 observed pixels):
        # what a trained Payne network looks like (Payne/utils/readc3k.py:441-447 builds the grid from a range and a resolution)
        for name, (k, w, rep) in (("C3", (args.steps, args.warmup, args.repeats)), ("C2r", (args.steps, args.warmup, min(args.repeats, 5))),
                                  ("LinNet300", (args.steps, args.warmup, min(args.repeats, 5))),
                                  ("C32k", (5, 2, 3)), ("C5", (3, 1, 3))):   # (C3 as long as the headline: a step is 0.04 ms)
            try:
                r = run_config(name, args, k, w, 0, 1, local_rank, repeats=rep, min_timed_s=1e-3 * args.min_timed_ms)
                blk = {"value": r["evals"] / r["dt"], "unit": "likelihood-evals/s", "steps": k, "warmup": w, "repeats": len(r["blocks"]),
                       "ms_per_step": 1e3 * r["dt"] / k, "ms_per_step_wall": 1e3 * r["dt_wall"] / k, "ms_per_step_min": 1e3 * min(r["blocks"]) / k,
                       "ms_per_step_max": 1e3 * max(r["blocks"]) / k, "workload": workload_text(name, r["dims"]),
                       "kernels": r["engines"][0].kernels_used()}
                if name == "C3" and cpu_c3 is not None:
                    blk["cpu_baseline"] = cpu_c3
                if r["kern"] is not None:
                    blk.update(roofline_blocks(name, r, args))
                for e in r["engines"]:
                    e.close()
                also[name] = blk
            except Exception as ex:                          # the headline must not die with a side run
                also[name] = {"error": "%s: %s" % (type(ex).__name__, ex)}
            r = None
            torch.cuda.empty_cache()
        out["also_measured"] = also

    if world == 1 and not args.no_e2e and B <= 4096 and d["N"] <= 16384:
        out["end_to_end"] = end_to_end(args.config, B, out["value"], args.e2e_calls, cpu=cpu)
        if args.config == "C2" and "also_measured" in out and "value" in out["also_measured"].get("C3", {}):
            try:
                out["also_measured"]["C3"]["end_to_end"] = end_to_end("C3", B, out["also_measured"]["C3"]["value"], args.e2e_calls, fit=False)
            except Exception as ex:
                out["also_measured"]["C3"]["end_to_end"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    emit(out)
    if grouped:
        dist.destroy_process_group()


def eng_round_trips(n1, variant=0):
    """Transfers of the spectrum (4 n1 bytes each, a read or a write) per evaluation.  payne_post_chip_kernel (65 536 points) and
    payne_post_chip2_kernel (32 768 points): the raw row in (1), the rotation stage's result out (1), the instrumental stage's gather of it (1), its
    result out (1), the observed grid's gather (1) = 5.  payne_post_big_kernel (other lengths above 16 384, or
    PAYNE_V_BIG_WORKSPACE): the row in and its copy out (2), per convolution stage the forward and the inverse transform at 2
    transfers per pass (the four-step form has 2 passes, the plain form one per radix-8 pass) + the taper pass (2), the resampling
    pass between the stages (2) and the closing interpolation read (1)."""
    import numpy as np
    if n1 in (65536, 32768) and not (variant & (32 | 256 | 65536)):      # payne_post_chip_kernel / payne_post_chip2_kernel
        return 5
    M = n1 // 2
    tiled = (M % 512 == 0) and (M // 512 in (32, 64, 128)) and not (variant & 32)
    per_fft = 2 if tiled else int(np.ceil(np.log2(M) / 3.0))
    per_stage = 2 * (2 * per_fft) + 2            # two transforms + the taper pass, reads and writes
    return 2 + 2 * per_stage + 2 + 1


if __name__ == "__main__":
    main()
