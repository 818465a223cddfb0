#!/usr/bin/env python
"""Headline benchmark: likelihood evaluations / second (BASELINE.json).

Workload C2: one star, 4096-pixel 2x300 YST1 ANN, 3600 observed
pixels, a batch of 512 candidate parameter vectors (the sampler's live points) per
step.  A step = one pass of the hot path (ANN forward -> vsini/Doppler/instrument
smoothing -> chi^2) over the batch, theta already resident in HBM.  With --gpus N
every rank fits its own synthetic star (weak scaling, no data-path collective);
posterior-style summaries are all-gathered over RCCL once after the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|small]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from thepayne_amd import synth  # noqa: E402

PEAK_FP32_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32 MFMA = vector peak (spec)
PEAK_HBM_GBS = 8000.0


def alg_flops_per_eval(D, H, N):
    """2(DH + H^2 + HN) + 2 stages x 2 transforms x 2.5 N log2 N + 60 N."""
    L = np.log2(N)
    return 2.0 * (D * H + H * H + H * N) + 2 * 2 * 2.5 * N * L + 60.0 * N


# ----------------------------------------------------------------------------
# CPU baseline: the oracle in the reference's calling pattern (one theta per
# lnprobfn call, fp64 numpy), one process per core.  Runs BEFORE the GPU is touched.
# ----------------------------------------------------------------------------
def _cpu_worker(args):
    cfg_name, budget_s, seed = args
    import oracle as O
    cfg = synth.CONFIGS[cfg_name]
    net = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], seed=0)
    obs = synth.obs_grid(net["wavelength"], cfg["nobs"])
    T = synth.TRUTH
    _, clean = O.getspec(net, Teff=T["Teff"], logg=T["logg"], feh=T["feh"], afe=T["afe"], rad_vel=T["vrad"],
                         rot_vel=T["vrot"], vmic=np.nan, inst_R=2.355 * T["inst_R"], outwave=obs)
    flux = clean + np.random.default_rng(0).normal(0, 0.01, len(obs))
    L = O.OracleLikelihood(net, obs, flux, np.full(len(obs), 0.01),
                           ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Inst_R'])
    th = synth.draw_candidates(4096, seed=100 + seed)
    O.lnprobfn(th[0], L)                      # warm
    n, t0 = 0, time.perf_counter()
    while True:                               # time-boxed: the sample is whatever fits the budget
        O.lnprobfn(th[n % len(th)], L)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s:
            return n, dt


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


def cpu_baseline(cfg_name, budget_s=10.0, max_procs=32):
    import multiprocessing as mp
    cores = min(usable_cores(), max_procs)
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = "1"                   # one thread per process, like the reference's einsum path
    ctx = mp.get_context("spawn")             # never fork a process that may hold a GPU
    with ctx.Pool(cores) as pool:
        t0 = time.perf_counter()
        res = pool.map(_cpu_worker, [(cfg_name, budget_s, i) for i in range(cores)])
        wall = time.perf_counter() - t0
    rates = [r[0] / r[1] for r in res]
    return dict(value=float(sum(rates)), unit="likelihood-evals/s", cores=cores, kind="port",
                per_core=float(np.mean(rates)),
                sample="numpy oracle lnprobfn, one theta per call, fp64, %s workload: %d processes x %.0f s "
                       "(%d calls in all; %.1f s wall incl. start-up)"
                       % (cfg_name, cores, budget_s, sum(r[0] for r in res), wall))


def pmc_step_bytes():
    """HBM bytes one likelihood batch moves, all three kernels (same PMC file and corrections as pmc_traffic)."""
    parts = [pmc_traffic(k) for k in ("dense_hidden", "dense_out", "post")]
    return None if any(p is None for p in parts) else float(sum(parts))


def pmc_traffic(kind):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes of the same command
    (profiles/r1_c2_rocprofv3_pmc_hbm.csv; FETCH_SIZE and WRITE_SIZE in separate runs, KB).  FETCH_SIZE
    is doubled as MI355X_MICROARCH.md prescribes for 16-B/lane streaming reads on gfx950."""
    path = os.path.join(ROOT, "profiles", "r1_c2_rocprofv3_pmc_hbm.csv")
    key = {"dense_out": "payne_dense_dma_kernel", "post": "payne_post_kernel", "dense_hidden": "payne_dense_hidden_kernel"}[kind]
    try:
        import csv
        tot, seen = {}, {}
        for r in csv.DictReader(open(path)):
            if key in r["kernel"] and int(r["launches"]) >= seen.get(r["counter"], 0):     # the steady-state variant
                tot[r["counter"]] = float(r["mean_value_per_launch_KB_raw"])
                seen[r["counter"]] = int(r["launches"])
        return (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0
    except Exception:
        return None


# ----------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the short nested-sampling run behind `end_to_end`")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("PAYNE_BENCH_STREAMS", "1")),
                    help="independent batches in flight (one engine + HIP stream each); a batched sampler with "
                         "two chain populations keeps two batches in flight")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    cfg = dict(synth.CONFIGS[args.config])
    B = args.batch or cfg["batch"]

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.config)

    import torch
    import torch.distributed as dist
    from thepayne_amd import nnio
    from thepayne_amd.engine import PayneEngine
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    # ---- this rank's star (seed = rank): same ANN replicated, own noise + truth jitter
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
    eng = PayneEngine(net, obs=(obs, flux, eflux), b_max=B, device=local_rank)
    th7 = synth.draw_candidates(B, seed=1 + rank)
    theta = eng.make_theta(B)
    theta[:, 0:6] = torch.as_tensor(th7[:, 0:6], device=theta.device)
    theta[:, 7] = torch.as_tensor(th7[:, 6], device=theta.device)
    lnl = torch.empty(B, dtype=torch.float64, device=theta.device)
    # extra in-flight batches: own context (workspaces), own stream, own theta / lnL
    S = max(1, args.streams)
    engines = [eng] + [PayneEngine(net, obs=(obs, flux, eflux), b_max=B, device=local_rank) for _ in range(S - 1)]
    streams = [torch.cuda.Stream(device=local_rank) for _ in range(S)] if S > 1 else [torch.cuda.current_stream()]
    thetas = [theta] + [theta.clone() for _ in range(S - 1)]
    lnls = [lnl] + [torch.empty_like(lnl) for _ in range(S - 1)]
    torch.cuda.synchronize()

    def step(i):
        j = i % S
        if S == 1:
            engines[0].lnlike_batch(thetas[0], out=lnls[0])
        else:
            with torch.cuda.stream(streams[j]):
                engines[j].lnlike_batch(thetas[j], out=lnls[j])

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=theta.device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    for l_ in lnls[1:]:                  # every in-flight batch evaluated the same candidates: same answers
        assert os.environ.get("PAYNE_SKIP") or bool(torch.equal(torch.nan_to_num(l_), torch.nan_to_num(lnl)))
    assert os.environ.get("PAYNE_SKIP") or int(torch.isfinite(lnl).sum()) >= B - 4, "non-finite lnL in the benchmark batch"   # Inst_R tail draws are NaN by contract

    # ---- per-kernel device time (HIP events on the launch stream), same K steps replayed
    kern = None
    if not args.no_kernel_timing:
        eng.profile(True)
        for _ in range(args.steps):
            eng.lnlike_batch(theta, out=lnl)
        torch.cuda.synchronize()
        kern = eng.profile_read()
        eng.profile(False)

    # ---- the one collective of the multi-star job: gather per-star summaries (RCCL)
    summary = torch.stack([lnl.max(), lnl.mean(), lnl.std(), theta[lnl.argmax(), 0], theta[lnl.argmax(), 1]])
    if world > 1:
        gathered = [torch.empty_like(summary) for _ in range(world)]
        dist.all_gather(gathered, summary)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    D, H, N = net["layers"][0][0].shape[1], net["layers"][0][0].shape[0], cfg["npix"]
    evals = world * B * args.steps
    out = {
        "metric": "likelihood-evals/sec (4k-pixel ANN, 512 live points)" if args.config == "C2" else
                  "likelihood-evals/sec (%s)" % args.config,
        "value": evals / dt, "unit": "likelihood-evals/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: single star per GPU, %d-pixel 2x%d YST1 ANN, %d observed pixels, batch of %d "
                               "candidate vectors per step (dynesty live points)" % (args.config, N, H, cfg["nobs"], B),
                   "batch": B, "npix": N, "nobs": cfg["nobs"], "stars": world,
                   "batches_in_flight": S,
                   "parallelism": "1 star per GPU, no data-path collective"},
    }
    if kern is not None:
        per = {k: (1e3 * v[0] / v[1] if v[1] else 0.0) for k, v in kern.items()}      # us per launch
        n1 = 1 << int(np.ceil(np.log2(N)))
        dom = max(("dense_out", "post"), key=lambda k: per[k])
        if n1 > 16384 and dom == "post":
            # spectra larger than LDS stream through a global workspace: HBM/L2-bound.
            # Algorithmic bytes per candidate: raw row in, 2 conv stages x (2 FFTs x log8(N/2) passes + taper)
            # x (read + write) of 4N, mask scan 8N, resample 8N, observed arrays 16 Nobs.
            npass = int(np.ceil(np.log2(n1 / 2) / 3.0))
            if os.environ.get("PAYNE_BIG_TILED", "1") != "0" and (n1 // 2) % 512 == 0 and (n1 // 2) // 512 in (32, 64, 128):
                npass = 2                                 # four-step transform: two round trips per FFT
            bytes_eval = 4.0 * n1 * (1 + 2 * (2 * npass + 1) * 2 + 2) + 8.0 * N + 16.0 * cfg["nobs"]
            out_passes = npass
            ach = bytes_eval * B / (max(per[dom], 1e-9) * 1e-6) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "payne_post_big_kernel", "achieved": ach, "peak": PEAK_HBM_GBS,
                               "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": None,
                               "alg_bytes_per_launch": bytes_eval * B, "avg_us_per_launch": per[dom],
                               "round_trips_per_fft": out_passes}
        else:
            # dominant kernel decides the roofline line; both are FLOP-bound at C2 (AI ~ 300 FLOP/B)
            flops = {"dense_out": 2.0 * B * H * N, "post": B * (2 * 2 * 2.5 * N * np.log2(N) + 60.0 * N)}
            ach = flops[dom] / (max(per[dom], 1e-9) * 1e-6) / 1e12
            out["roofline"] = {"bound": "mfma",
                               "kernel": "payne_dense_dma_kernel (output layer)" if dom == "dense_out" else "payne_post_kernel",
                               "achieved": ach, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS,
                               "traffic": pmc_traffic(dom) if args.config == "C2" else None,
                               "alg_flops_per_launch": flops[dom], "avg_us_per_launch": per[dom]}
            if dom == "post":
                out["roofline"]["note"] = ("FFT / interpolation pipeline in LDS: no MFMA instructions; priced against the "
                                           "fp32 peak (vector = MFMA f32 = 157.3 TFLOP/s). The MFMA kernel of the step is "
                                           "under `mfma_kernel`.")
            out["mfma_kernel"] = {"kernel": "payne_dense_dma_kernel (output layer)", "alg_flops_per_launch": flops["dense_out"],
                                  "avg_us_per_launch": per["dense_out"],
                                  "achieved_tflops": flops["dense_out"] / (max(per["dense_out"], 1e-9) * 1e-6) / 1e12,
                                  "frac_of_fp32_peak": flops["dense_out"] / (max(per["dense_out"], 1e-9) * 1e-6) / 1e12 / PEAK_FP32_TFLOPS}
        out["kernels_us"] = per
        if args.config == "C2" and B == 512:
            # whole-step HBM rate: the north star asks for the achieved HBM-bandwidth fraction; this path is
            # FLOP/latency-bound at C2 (AI ~ 300 FLOP/B), so the fraction is small by construction
            step_bytes = pmc_step_bytes()
            alg = 4.0 * (D * H + H + H * H + H + H * N + N) + 8.0 * N + 16.0 * cfg["nobs"] + B * (8.0 * 12 + 8) \
                + 2.0 * 4.0 * B * N                     # + the spectra written by the output layer and read back once
            if step_bytes is not None:
                gbs = step_bytes / (1e-3 * out["ms_per_step"]) / 1e9
                out["hbm"] = {"achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                              "pmc_bytes_per_step": step_bytes, "alg_bytes_per_step": alg,
                              "source": "profiles/r1_c2_rocprofv3_pmc_hbm.csv (2 x FETCH_SIZE + WRITE_SIZE, three kernels)"}
        out["alg_flops_per_eval"] = alg_flops_per_eval(D, H, N)
        out["whole_path_tflops"] = alg_flops_per_eval(D, H, N) * B / (max(sum(per.values()), 1e-9) * 1e-6) / 1e12
    if cpu is not None:
        out["cpu_baseline"] = cpu
    if world == 1 and not args.no_e2e and B <= 4096 and cfg["npix"] <= 16384:
        # the same likelihood as the batched nested sampler sees it -- prior transform,
        # random-walk proposals (device), transfers and the dead-point bookkeeping included
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
        import sampler_bench
        eng.close()
        e2e = sampler_bench.run(args.config, maxcall=250000, nlive=B, walks=25, modes=("device_chunks",))["device_chunks"]
        out["end_to_end"] = {"value": e2e["evals_per_s"], "unit": "likelihood-evals/s", "calls": e2e["calls"],
                             "iterations": e2e["iterations"], "seconds": e2e["seconds"],
                             "what": "static nested sampler, %d live points, rwalk x25 on the device, bound='multi' "
                                     "(ellipsoid decomposition by recursive 2-means), dead points consumed in bulk" % B}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
