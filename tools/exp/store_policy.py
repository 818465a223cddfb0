#!/usr/bin/env python
"""Cache policy of the output layer's row stores (payne_dense_dma2h_kernel): what do the output layer and the post kernel that reads the
rows pay for each?  Twins built with -DPAYNE_EXP_ST=n (1 plain, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0 nt; 0 = the product: nt), C2 bench, twice.

    python tools/exp/store_policy.py [n ...]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from thepayne_amd import build  # noqa: E402

which = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 5]
libs = {b: (build.build_variant("st%d" % b, ["-DPAYNE_EXP_ST=%d" % b]) if b else build.build_lib()) for b in which}
for rep in range(int(os.environ.get("REPS", "2"))):
    for b in which:
        env = dict(os.environ, PAYNE_HIP_LIB=libs[b])
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "300", "--warmup", "30", "--no-cpu-baseline", "--no-e2e",
                              "--no-also"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(res.stdout.strip().splitlines()[-1])
            print("policy %d  step %.2f us  kernels %s" % (b, 1e3 * d["ms_per_step"], {k: round(v, 2) for k, v in d["kernels_us"].items()}), flush=True)
        except Exception:
            print("policy", b, "failed", res.stdout[-300:], res.stderr[-600:])
