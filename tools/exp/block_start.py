"""What a timed block of K steps pays for starting on an idle GPU (bench.py's blocks: barrier, event, K steps, event, barrier).

Block time by the event pair for K = 1 .. 200 -> a + b K; the host's time per `lnlike_batch` call (enqueue only); the same blocks
with the steps enqueued by ONE native call per step and nothing else in Python (`lib.payne_lnlike_batch` with the arguments made
ahead).  GPU box only.
"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench


def blocks(fn, K, reps=15):
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _i in range(K):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(1e3 * e0.elapsed_time(e1))
    return float(np.median(out)), float(np.min(out))


def main():
    P = bench.make_problem("C2", 512, 0, 0)
    eng, theta, lnl = P["engines"][0], P["theta"], P["lnl"]
    for _ in range(50):
        eng.lnlike_batch(theta, out=lnl)
    torch.cuda.synchronize()
    # host time per call, nothing waiting
    t0 = time.perf_counter()
    for _ in range(200):
        eng.lnlike_batch(theta, out=lnl)
    t_host = (time.perf_counter() - t0) / 200
    torch.cuda.synchronize()
    tp, op, st, ctx, lib = theta.data_ptr(), lnl.data_ptr(), eng._stream(), eng._ctx, eng.lib
    t0 = time.perf_counter()
    for _ in range(200):
        lib.payne_lnlike_batch(ctx, tp, 512, op, st)
    t_raw = (time.perf_counter() - t0) / 200
    torch.cuda.synchronize()
    print(f"host per call: engine.lnlike_batch {1e6 * t_host:.2f} us (GPU-bound if ~ the step), raw ctypes call {1e6 * t_raw:.2f} us")
    # enqueue-only host cost with an idle queue: 5 calls after a synchronize
    hs = []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _i in range(5):
            eng.lnlike_batch(theta, out=lnl)
        hs.append((time.perf_counter() - t0) / 5)
    hr = []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _i in range(5):
            lib.payne_lnlike_batch(ctx, tp, 512, op, st)
        hr.append((time.perf_counter() - t0) / 5)
    print(f"enqueue only (5 calls into an idle queue): engine {1e6 * np.median(hs):.2f} us a call, raw {1e6 * np.median(hr):.2f} us a call")
    for name, fn in (("engine.lnlike_batch", lambda: eng.lnlike_batch(theta, out=lnl)),
                     ("raw call", lambda: lib.payne_lnlike_batch(ctx, tp, 512, op, st))):
        Ks = (1, 2, 3, 5, 10, 20, 50, 200)
        med = [blocks(fn, K) for K in Ks]
        A = np.vstack([np.ones(len(Ks)), Ks]).T
        a, b = np.linalg.lstsq(A[3:], np.array([m[0] for m in med])[3:], rcond=None)[0]
        print(f"{name}: block = {a:.1f} + {b:.2f} K us (fit on K >= 5)")
        for K, (m, mn) in zip(Ks, med):
            print(f"   K {K:4d}: median {m:9.1f} us  min {mn:9.1f}  per step {m / K:7.2f}   {512 * K / m:6.2f} M/s")


if __name__ == "__main__":
    main()
