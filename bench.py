#!/usr/bin/env python
"""Headline benchmark: likelihood evaluations / second (BASELINE.json).

Workload C2: one star, 4096-pixel 2x300 YST1 ANN, 3600 observed
pixels, a batch of 512 candidate parameter vectors (the sampler's live points) per
step.  A step = one pass of the hot path (ANN forward -> vsini/Doppler/instrument
smoothing -> chi^2) over the batch, theta already resident in HBM.  With --gpus N
every rank fits its own synthetic star (weak scaling, no data-path collective);
posterior-style summaries are all-gathered over RCCL once after the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C5|small]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` without a torchrun environment starts the N ranks itself (fresh child
processes, before this process touches a GPU).  Prints ONE JSON line (rank 0).
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
    return ap.parse_args()


# ----------------------------------------------------------------------------
# N ranks from one command line.  The parent never imports torch or touches HIP: it only
# starts N copies of this script with the torchrun environment and relays rank 0's line.
# ----------------------------------------------------------------------------
def launch_ranks(n):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:            # a failed rank leaves the others waiting in a collective: end them
                    q.terminate()
        time.sleep(0.05)
    return rc


# ----------------------------------------------------------------------------
# CPU baseline: the oracle in the reference's calling pattern (one theta per
# lnprobfn call, fp64 numpy), one process per core.  Runs BEFORE the GPU is touched.
# ----------------------------------------------------------------------------
def _cpu_worker(args):
    cfg_name, budget_s, seed = args
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
    out = []
    for loop in (True, False):                # the reference's per-pixel chi^2 loop (likelihood.py:95-97), then vectorised
        L = O.OracleLikelihood(net, obs, flux, np.full(len(obs), 0.01), names, pixel_loop=loop)
        O.lnprobfn(th[0], L)                  # warm
        n, t0 = 0, time.perf_counter()
        while True:                           # time-boxed: the sample is whatever fits the budget
            O.lnprobfn(th[n % len(th)], L)
            n += 1
            dt = time.perf_counter() - t0
            if dt >= budget_s / 2:
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


def cpu_baseline(cfg_name, budget_s=12.0, max_procs=32):
    import multiprocessing as mp
    cores = min(usable_cores(), max_procs)
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = "1"                   # one thread per process, like the reference's einsum path
    ctx = mp.get_context("spawn")             # never fork a process that may hold a GPU
    with ctx.Pool(cores) as pool:
        t0 = time.perf_counter()
        res = pool.map(_cpu_worker, [(cfg_name, budget_s, i) for i in range(cores)])
        wall = time.perf_counter() - t0
    loop = [r[0][0] / r[0][1] for r in res]
    vect = [r[1][0] / r[1][1] for r in res]
    return dict(value=float(sum(loop)), unit="likelihood-evals/s", cores=cores, kind="port",
                per_core=float(sum(loop) / cores),
                vectorised_chi2={"value": float(sum(vect)), "per_core": float(sum(vect) / cores),
                                 "what": "the same port with chi^2 as one numpy expression instead of the reference's "
                                         "per-pixel Python loop"},
                sample="numpy oracle lnprobfn, one theta per call, fp64, chi^2 by the per-pixel Python loop of "
                       "Payne/fitting/likelihood.py:95-97, %s workload: %d processes x %.0f s (%d calls; then %.0f s "
                       "with vectorised chi^2, %d calls; %.1f s wall incl. start-up)"
                       % (cfg_name, cores, budget_s / 2, sum(r[0][0] for r in res), budget_s / 2,
                          sum(r[1][0] for r in res), wall),
                reference_measured_elsewhere="251 evals/s/core (C2), 11.6 (C5): the reference itself on the survey "
                                             "container's CPU (BASELINE.md)")


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


def pmc_bytes(path, key):
    """2 x FETCH_SIZE + WRITE_SIZE (KB, separate passes; FETCH doubled as MI355X_MICROARCH.md prescribes for
    16-B/lane streaming reads on gfx950) per launch of the kernels whose name contains `key`."""
    import csv
    tot, seen = {}, {}
    for r in csv.DictReader(open(path)):
        if key in r["kernel"] and int(r["launches"]) >= seen.get(r["counter"], 0):     # the steady-state variant
            tot[r["counter"]] = float(r["mean_value_per_launch_KB_raw"])
            seen[r["counter"]] = int(r["launches"])
    if "FETCH_SIZE" not in tot or "WRITE_SIZE" not in tot:
        return None
    return (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0


KERNEL_KEYS = {"dense_out": "payne_dense_dma", "post": "payne_post", "dense_hidden": "payne_dense_hidden_kernel"}


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
    ndev = torch.cuda.device_count()
    if world > ndev:
        raise SystemExit("--gpus %d but this node shows %d GPU(s): one rank per GPU (RCCL refuses two ranks on a device)"
                         % (world, ndev))
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
    eng = PayneEngine(net, obs=(obs, flux, eflux), b_max=B, device=local_rank, variant=args.variant)
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
    dt_local = time.perf_counter() - t0
    tmax = torch.tensor([dt_local], dtype=torch.float64, device=theta.device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    for l_ in lnls[1:]:                  # every in-flight batch evaluated the same candidates: same answers
        assert args.unchecked or bool(torch.equal(torch.nan_to_num(l_), torch.nan_to_num(lnl)))
    assert args.unchecked or int(torch.isfinite(lnl).sum()) >= B - 4, "non-finite lnL in the benchmark batch"   # Inst_R tail draws are NaN by contract

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
    summary = torch.stack([lnl.max(), lnl.mean(), lnl.std(), theta[lnl.argmax(), 0], theta[lnl.argmax(), 1],
                           torch.tensor(B * args.steps / dt_local, dtype=torch.float64, device=theta.device),
                           torch.tensor(float(rank), dtype=torch.float64, device=theta.device)])
    gathered = [summary]
    if world > 1:
        gathered = [torch.empty_like(summary) for _ in range(world)]
        dist.all_gather(gathered, summary)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    table = torch.stack(gathered).cpu().numpy()
    assert sorted(int(r) for r in table[:, 6]) == list(range(world)), "the gather did not see every rank"

    D, H, N = net["layers"][0][0].shape[1], net["layers"][0][0].shape[0], cfg["npix"]
    n1 = 1 << int(np.ceil(np.log2(N)))
    L2N = np.log2(N)
    flops_eval = 2.0 * (D * H + H * H + H * N) + 2 * 2 * 2.5 * N * L2N + 60.0 * N
    bytes_batch = 4.0 * (D * H + H + H * H + H + H * N + N) + 8.0 * N + 16.0 * cfg["nobs"] + B * (8.0 * 12 + 4)   # compulsory
    evals = world * B * args.steps
    out = {
        "metric": "likelihood-evals/sec (4k-pixel ANN, 512 live points)" if args.config == "C2" else
                  "likelihood-evals/sec (%s)" % args.config,
        "value": evals / dt, "unit": "likelihood-evals/s",
        "n_gpus": dist.get_world_size() if world > 1 else 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: single star per GPU, %d-pixel 2x%d YST1 ANN, %d observed pixels, batch of %d "
                               "candidate vectors per step (dynesty live points)" % (args.config, N, H, cfg["nobs"], B),
                   "batch": B, "npix": N, "nobs": cfg["nobs"], "stars": world,
                   "batches_in_flight": S, "kernel_variant": args.variant,
                   "arithmetic": "fp32 storage and accumulation throughout (wavelengths, tapers, chi^2 sums fp64); the output "
                                 "layer's products: " + ("v_mfma_f32_32x32x2_f32" if args.variant & 4096 else
                                 "operands split exactly in three bf16 parts, six exact partial products, fp32 accumulator "
                                 "(as accurate as the fp32 fma chain against an fp64 product: tests/test_gpu_parity.py; "
                                 "--variant 4096 = the fp32 matrix instruction)"),
                   "parallelism": "1 star per GPU, no data-path collective"},
        **({"invalid": "--unchecked: a timing experiment, not a benchmark result"} if args.unchecked else {}),
        "rccl_world": dist.get_world_size() if world > 1 else 1,
        "per_rank_evals_per_s": [float(v) for v in table[np.argsort(table[:, 6]), 5]],
    }
    if kern is not None:
        per = {k: (1e3 * v[0] / v[1] if v[1] else 0.0) for k, v in kern.items()}      # us per launch
        dom = max(("dense_out", "post"), key=lambda k: per[k])
        csv_path, meta = pmc_file(args.config)
        traffic = {k: (pmc_bytes(csv_path, key) if csv_path else None) for k, key in KERNEL_KEYS.items()}
        tsrc = ("%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, build %s = the running one)"
                % (os.path.relpath(csv_path, ROOT), meta["source_hash"])) if csv_path else \
               ("none: no PMC passes under profiles/ were taken with the running build (%s)" % meta["source_hash"])
        big = n1 > 16384
        if big and dom == "post":
            # spectra larger than LDS stream through a global workspace: HBM/L2-bound.
            # SURVEY 8(d) prices the streaming variant at the compulsory bytes + passes x 4N per evaluation with passes = 8.
            passes_used = eng_round_trips(n1)
            alg8 = (bytes_batch / B + 8 * 4.0 * N) * B
            t = max(per[dom], 1e-9) * 1e-6
            ach = alg8 / t / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "payne_post_big_kernel", "achieved": ach, "peak": PEAK_HBM_GBS,
                               "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": traffic["post"],
                               "traffic_source": tsrc,
                               "alg_bytes_per_launch": alg8,
                               "alg_basis": "SURVEY 8(d): compulsory %.1f KB + 8 spectrum passes x 4N = %.2f MB per evaluation"
                                            % (bytes_batch / B / 1e3, alg8 / B / 1e6),
                               "passes_used": passes_used,
                               "workspace_bytes_per_launch": passes_used * 4.0 * n1 * B,
                               "workspace_rate_GBs": passes_used * 4.0 * n1 * B / t / 1e9,
                               "avg_us_per_launch": per[dom]}
            if traffic["post"]:
                out["roofline"]["counter_rate_GBs"] = traffic["post"] / t / 1e9
        else:
            flops = {"dense_out": 2.0 * B * H * N, "post": B * (2 * 2 * 2.5 * N * L2N + 60.0 * N)}
            ach = flops[dom] / (max(per[dom], 1e-9) * 1e-6) / 1e12
            out["roofline"] = {"bound": "mfma" if dom == "dense_out" else "fp32-vector",
                               "kernel": "payne_dense_dma_kernel (output layer)" if dom == "dense_out" else "payne_post_kernel",
                               "achieved": ach, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS,
                               "traffic": traffic[dom], "traffic_source": tsrc,
                               "alg_flops_per_launch": flops[dom], "avg_us_per_launch": per[dom]}
            if dom == "post":
                out["roofline"]["note"] = ("FFT / interpolation pipeline in LDS: no MFMA instructions; priced against the "
                                           "packed-fp32 vector peak (= the fp32 MFMA peak, 157.3 TFLOP/s). The MFMA kernel of "
                                           "the step is under `mfma_kernel`.")
            t_out = max(per["dense_out"], 1e-9) * 1e-6
            split = not (args.variant & (4096 | 1 | 1024))          # the default output layer: six bf16 products per fp32 product
            out["mfma_kernel"] = {"kernel": "payne_dense_dma3_kernel (output layer, 3 x bf16 split)" if split else
                                            "payne_dense_dma_kernel (output layer, fp32 matrix instruction)",
                                  "alg_flops_per_launch": flops["dense_out"],
                                  "avg_us_per_launch": per["dense_out"],
                                  "achieved_tflops": flops["dense_out"] / t_out / 1e12,
                                  "frac_of_fp32_peak": flops["dense_out"] / t_out / 1e12 / PEAK_FP32_TFLOPS}
            if split:                                                # what the matrix pipe executes: 6 bf16 flops per algorithmic flop
                out["mfma_kernel"].update({"executed_bf16_tflops": 6.0 * flops["dense_out"] / t_out / 1e12,
                                           "frac_of_bf16_peak": 6.0 * flops["dense_out"] / t_out / 1e12 / PEAK_BF16_TFLOPS,
                                           "note": "fp32-accurate products as six bf16 partial products: `frac_of_fp32_peak` compares the "
                                                   "algorithmic fp32 work with what the fp32 matrix instruction could do at best, "
                                                   "`frac_of_bf16_peak` the executed bf16 work with the dense bf16 peak (2.5 PFLOP/s); "
                                                   "the kernel is bound by the 36 KB of operand planes a k-step brings in"})
        out["kernels_us"] = per
        out["alg_flops_per_eval"] = flops_eval
        t_k = max(sum(per.values()), 1e-9) * 1e-6
        out["whole_step"] = {"alg_flops": flops_eval * B,
                             "tflops_on_kernel_time": flops_eval * B / t_k / 1e12,
                             "tflops_on_wall_time": flops_eval * B / (1e-3 * out["ms_per_step"]) / 1e12,
                             "frac_of_fp32_peak_on_wall_time": flops_eval * B / (1e-3 * out["ms_per_step"]) / 1e12 / PEAK_FP32_TFLOPS}
        # whole-step HBM rate (the north star asks for the achieved HBM-bandwidth fraction)
        alg_step = bytes_batch + (0.0 if big else 2.0 * 4.0 * B * N)   # + spectra written by the output layer, read back once
        hbm = {"alg_bytes_per_step": alg_step, "alg_rate_GBs": alg_step / (1e-3 * out["ms_per_step"]) / 1e9,
               "peak": PEAK_HBM_GBS, "unit": "GB/s", "source": tsrc}
        if all(v is not None for v in traffic.values()):
            sb = float(sum(traffic.values()))
            hbm.update({"pmc_bytes_per_step": sb, "achieved": sb / (1e-3 * out["ms_per_step"]) / 1e9,
                        "frac": sb / (1e-3 * out["ms_per_step"]) / 1e9 / PEAK_HBM_GBS})
        out["hbm"] = hbm
    if cpu is not None:
        out["cpu_baseline"] = cpu
    if world == 1 and not args.no_e2e and B <= 4096 and cfg["npix"] <= 16384:
        # the same likelihood as the batched nested sampler sees it -- prior transform,
        # random-walk proposals (device), transfers and the dead-point bookkeeping included
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import sampler_bench
        eng.close()
        e2e = sampler_bench.run(args.config, maxcall=250000, nlive=B, walks=25, modes=("device_chunks",))["device_chunks"]
        out["end_to_end"] = {"value": e2e["evals_per_s"], "unit": "likelihood-evals/s", "calls": e2e["calls"],
                             "iterations": e2e["iterations"], "seconds": e2e["seconds"],
                             "frac_of_kernel_only": e2e["evals_per_s"] / out["value"],
                             "what": "static nested sampler, %d live points, rwalk x25 on the device, bound='multi' "
                                     "(ellipsoid decomposition by recursive 2-means), dead points consumed in bulk" % B}
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def eng_round_trips(n1):
    """Spectrum-sized round trips through the global workspace per evaluation in payne_post_big_kernel: the row
    load, per convolution stage 2 transforms x (2 with the four-step form | one per radix-8 pass) + the taper
    pass, the resampling pass and the closing interpolation read."""
    import numpy as np
    M = n1 // 2
    tiled = (M % 512 == 0) and (M // 512 in (32, 64, 128))
    per_fft = 2 if tiled else int(np.ceil(np.log2(M) / 3.0))
    # each pass reads and writes the spectrum once: count a pass as 2 transfers of 4 n1 bytes
    return 1 + 2 * (2 * per_fft + 1) * 2 + 2 + 1


if __name__ == "__main__":
    main()
