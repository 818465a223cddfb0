"""Does a 20-step block's time depend on how long the GPU has been busy?  (bench.py --steps 20 --warmup 5 is 9 ms of timed work in all.)
A fresh process, five warm-up steps, then 80 blocks of 20 steps timed as bench.py times them; per-step microseconds by block index."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench

P = bench.make_problem("C2", 512, 0, 0)
eng, theta, lnl = P["engines"][0], P["theta"], P["lnl"]
for _ in range(5):
    eng.lnlike_batch(theta, out=lnl)
out = []
t_start = time.perf_counter()
for b in range(80):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        eng.lnlike_batch(theta, out=lnl)
    e1.record()
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t_start, 1e3 * e0.elapsed_time(e1) / 20))
for i in range(0, 80, 4):
    print("blocks %2d-%2d (t = %6.1f ms): " % (i, i + 3, 1e3 * out[i][0]) + "  ".join("%.2f" % o[1] for o in out[i:i + 4]))
print("median of the first 15: %.2f us/step; of the last 15: %.2f" % (np.median([o[1] for o in out[:15]]), np.median([o[1] for o in out[-15:]])))
