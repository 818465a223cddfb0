#!/usr/bin/env python
"""Kernel times of payne_dense_dma2h_kernel with pieces compiled out (-DPAYNE_EXP_D2H=bits, NO stamps: production shape;
results are garbage, bench.py --unchecked).  bits: 1 every piece from one kilobyte, 2 one of the three products, 4 no stores, 8 / 16 half of the waves store after two / four stages of five and leave.

    python tools/exp/d2h_ablate_time.py [bits ...]     # default 0 1 2 4 3 7
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from thepayne_amd import build  # noqa: E402

bits = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 3, 7]
libs = {b: (build.build_variant("d2ht%d" % b, ["-DPAYNE_EXP_D2H=%d" % b]) if b else build.build_lib()) for b in bits}
for rep in range(2):
    for b in bits:
        env = dict(os.environ, PAYNE_HIP_LIB=libs[b])
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "300", "--warmup", "30", "--no-cpu-baseline", "--no-e2e",
                              "--no-also", "--unchecked"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(res.stdout.strip().splitlines()[-1])
            print("bits %2d  step %.2f us  kernels %s" % (b, 1e3 * d["ms_per_step"], {k: round(v, 2) for k, v in d["kernels_us"].items()}), flush=True)
        except Exception:
            print("bits", b, "failed", res.stdout[-500:], res.stderr[-800:])
