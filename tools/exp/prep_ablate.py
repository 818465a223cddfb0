#!/usr/bin/env python
"""What in prep_candidate (the hidden-layer launch's last workgroups) costs what: twins with -DPAYNE_EXP_PREP=bits (1: the window's guesses
taken on trust, no second memory round trip; 2: the window left to the post kernel), C2 bench (--unchecked: bit 1 can be wrong), twice."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from thepayne_amd import build  # noqa: E402
which = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3]
libs = {b: (build.build_variant("prep%d" % b, ["-DPAYNE_EXP_PREP=%d" % b]) if b else build.build_lib()) for b in which}
for rep in range(2):
    for b in which:
        env = dict(os.environ, PAYNE_HIP_LIB=libs[b])
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "300", "--warmup", "30", "--no-cpu-baseline", "--no-e2e", "--no-also",
                              "--unchecked"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(res.stdout.strip().splitlines()[-1])
            print("bits %d  step %.2f us  kernels %s" % (b, 1e3 * d["ms_per_step"], {k: round(v, 2) for k, v in d["kernels_us"].items()}), flush=True)
        except Exception:
            print("bits", b, "failed", res.stdout[-300:], res.stderr[-600:])
