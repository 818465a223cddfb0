"""The phases around the on-chip stages (C5 / C32k): twins with -DPAYNE_EXP_SKIP=<mask> (16: no observed-grid loop) and
-DPAYNE_EXP_CHIP=15 (stages do nothing between their loads and stores), alone and together."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from thepayne_amd import build  # noqa: E402
TW = {"full": [], "no observed-grid loop": ["-DPAYNE_EXP_SKIP=16"], "stages empty": ["-DPAYNE_EXP_CHIP=15"],
      "stages empty, no observed-grid loop": ["-DPAYNE_EXP_CHIP=15", "-DPAYNE_EXP_SKIP=16"]}
if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "C5"
    base = None
    for i, (name, flags) in enumerate(TW.items()):
        lib = os.path.join(build.variant_dir(), "libpayne_hip_ab2_%d.so" % i)
        if "--rebuild" in sys.argv or not os.path.exists(lib):
            lib = build.build_variant("ab2_%d" % i, flags)
        if "--build" in sys.argv:
            continue
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--steps", "5", "--warmup", "2", "--repeats", "3",
                              "--no-cpu-baseline", "--no-e2e", "--no-also", "--unchecked"], env=dict(os.environ, PAYNE_HIP_LIB=lib), capture_output=True, text=True)
        try:
            t = json.loads(out.stdout.strip().splitlines()[-1])["kernels_us"]["post"]
        except Exception:
            print(name, "FAILED", out.stderr[-300:]); continue
        base = t if base is None else base
        print("%-44s post %8.1f us   (%+8.1f)" % (name, t, t - base), flush=True)
