"""Where does an on-chip convolution stage (payne_post_chip_kernel, C5) spend its time?  Twins of the library with one group of the
stage's pieces compiled out (-DPAYNE_EXP_CHIP=<mask>, results are wrong by design); the drop of the post kernel's time against
the full kernel is what that group costs in place.

    python tools/exp/chip_ablate.py [C5|C32k]          # on the GPU box; the twins are built on first use
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from thepayne_amd import build  # noqa: E402

MASKS = {"full": 0, "no radix-32 transforms": 1, "no twiddles": 2, "no LDS exchanges": 4, "no taper arithmetic": 8,
         "no arithmetic at all (1+2+8)": 11, "loads, stores and the phases around the stages only (15)": 15}
if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "C5"
    only_build = "--build" in sys.argv
    base = None
    for name, mask in MASKS.items():
        lib = os.path.join(build.variant_dir(), "libpayne_hip_chip%d.so" % mask)
        if "--rebuild" in sys.argv or not os.path.exists(lib):       # (twins built elsewhere and shipped without their objects are used as they are)
            lib = build.build_variant("chip%d" % mask, ["-DPAYNE_EXP_CHIP=%d" % mask])
        if only_build:
            continue
        env = dict(os.environ, PAYNE_HIP_LIB=lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--steps", "5", "--warmup", "2", "--repeats", "3",
                              "--no-cpu-baseline", "--no-e2e", "--no-also", "--unchecked"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            print(name, "FAILED", out.stderr[-400:])
            continue
        t = d["kernels_us"]["post"]
        base = t if base is None else base
        print("%-60s post %8.1f us   (%+8.1f)" % (name, t, t - base), flush=True)
