#!/usr/bin/env python
"""Where do payne_dense_dma3f_kernel's ~1 300 cycles per k-step go?  (GPU box)

Stamped twins of the library with pieces of the kernel compiled out (-DPAYNE_EXP_D3F=bits; results are garbage, only the
stamps count): 1 = no split arithmetic (the loaded bits stored as they are), 2 = three of the six matrix products,
4 = the weights' loads all hit one line, and combinations.  Prints the k-step medians of each twin.

    python tools/exp/d3f_ablate.py [bits ...]        # default: 0 1 2 4 7
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from thepayne_amd import build  # noqa: E402

bits = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 7]
for b in bits:
    lib = build.build_variant("d3f%d" % b, ["-DPAYNE_STAMPS", "-DPAYNE_EXP_D3F=%d" % b])
    env = dict(os.environ, STAMP_LIB=lib, PAYNE_DIAG_UNCHECKED="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "post_stamps.py")], env=env, capture_output=True, text=True)
    lines = [l for l in res.stdout.splitlines() if "k-step" in l or "prologue" in l or "epilogue" in l or "whole workgroup median" in l]
    print("== PAYNE_EXP_D3F=%d" % b)
    print("\n".join(lines[-13:]) if lines else res.stdout[-1500:] + res.stderr[-1500:])
