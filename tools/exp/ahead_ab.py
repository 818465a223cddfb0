"""End to end with the walk's hidden layers computed ahead (the default) against every batch launching its own
(PAYNE_V_NO_HIDDEN_AHEAD), interleaved on one box.   python tools/exp/ahead_ab.py [config] [runs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import sampler_bench
from thepayne_amd import _lib
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for i in range(runs):
    for name, v in (("ahead", 0), ("own hidden launch", _lib.V_NO_HIDDEN_AHEAD)):
        r = sampler_bench.run(cfg, maxcall=700000, modes=("device_chunks",), seed=1 + i, dlogz=1e-9, variant=v)["device_chunks"]
        print("%-18s %9d calls/s  (%d calls, %d iterations, ln Z %.3f)" % (name, r["evals_per_s"], r["calls"], r["iterations"], r["logz"]), flush=True)
