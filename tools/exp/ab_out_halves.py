"""The output layer's tile finished in three parts (payne_dense_dma2p_kernel; PAYNE_OUT_PARTS=2: in two halves, payne_dense_dma2hh_kernel) against the whole tile (PAYNE_V_OUT_WHOLE_TILE): same rows to the
bit, and the step's times side by side (same box, interleaved).  GPU box."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from thepayne_amd import _lib

outs = {}
for v in (0, _lib.V_OUT_WHOLE_TILE):
    class A: variant = v
    P = bench.make_problem("C2", 512, 0, 0, variant=v)
    eng, theta, lnl = P["engines"][0], P["theta"], P["lnl"]
    eng.lnlike_batch(theta, out=lnl); torch.cuda.synchronize()
    rows = eng.predict_batch(theta, stage=0).cpu().numpy()
    outs[v] = (lnl.cpu().numpy().copy(), rows, eng.kernels_used())
    eng.close()
print({v: o[2]["out"] for v, o in outs.items()})
a, b = outs[0], outs[_lib.V_OUT_WHOLE_TILE]
print("lnL equal to the bit:", np.array_equal(np.nan_to_num(a[0]), np.nan_to_num(b[0])), " rows equal to the bit:", np.array_equal(a[1], b[1]),
      " max |d rows|", float(np.abs(a[1] - b[1]).max()))
for rep in range(int(os.environ.get("REPS", "4"))):
    for name, v, parts in (("halves", 0, "2"), ("whole", _lib.V_OUT_WHOLE_TILE, "2")):       # (("3 parts", 0, "3"): with tools/exp/out_three_parts.patch applied)
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "300", "--warmup", "30", "--no-cpu-baseline", "--no-e2e", "--no-also",
                              "--variant", str(v)], capture_output=True, text=True, env=dict(os.environ, PAYNE_OUT_PARTS=parts))
        try:
            d = json.loads(res.stdout.strip().splitlines()[-1])
            print("%-7s step %.2f us  %.2f M/s  kernels %s" % (name, 1e3 * d["ms_per_step"], d["value"] / 1e6, {k: round(x, 2) for k, x in d["kernels_us"].items()}), flush=True)
        except Exception:
            print(name, "failed", res.stdout[-300:], res.stderr[-800:])
