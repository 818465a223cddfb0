"""Where does payne_post_kernel's time go in the PRODUCTION code?  Builds twins of the library with one group of phases
compiled out (-DPAYNE_EXP_SKIP=<mask>, results are wrong by design) and times the post kernel of the C2 batch with
each: the drop against the full kernel is what that group costs in place.  (Cycle stamps need a diagnostic build whose
extra state changes the kernel's register allocation and occupancy: its shares are not the production kernel's.)

    python tools/exp/ablate.py            # on the GPU box; the twins are built on first use (hipcc, ~1 min each)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from thepayne_amd import build  # noqa: E402

MASKS = {"full": 0, "no fwd transforms": 1, "no tapers": 2, "no inv transforms": 4, "no resampling": 8, "no obs/chi2": 16,
         "no row load": 32, "no transforms": 5, "nothing but start/end": 63}
if __name__ == "__main__":
    only_build = "--build" in sys.argv
    base = None
    for name, mask in MASKS.items():
        lib = os.path.join(build.variant_dir(), "libpayne_hip_exp%d.so" % mask)
        if "--rebuild" in sys.argv or not os.path.exists(lib):       # (twins built elsewhere and shipped without their objects are used as they are)
            lib = build.build_variant("exp%d" % mask, ["-DPAYNE_EXP_SKIP=%d" % mask])
        if only_build:
            continue
        env = dict(os.environ, PAYNE_HIP_LIB=lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "300", "--warmup", "30", "--no-cpu-baseline",
                              "--no-e2e", "--unchecked"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            print(name, "FAILED", out.stderr[-400:])
            continue
        t = d["kernels_us"]["post"]
        base = t if base is None else base
        print("%-26s post %6.2f us   (%+6.2f)" % (name, t, t - base), flush=True)
