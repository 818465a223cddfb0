#!/usr/bin/env python
"""Where does payne_post_kernel spend its cycles?  (GPU box; diagnostic build)

Builds libpayne_hip_diag.so (-DPAYNE_STAMPS), runs one C2 batch and prints the median
cycles between consecutive phase barriers.  The diagnostic build's run time is not a
benchmark number: read the SHARES.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from thepayne_amd import build, _lib  # noqa: E402

path = build.build_diag()
os.environ["PAYNE_HIP_LIB"] = path
from thepayne_amd import synth, nnio  # noqa: E402
from thepayne_amd.engine import PayneEngine  # noqa: E402
from helpers import theta_full, yst_problem  # noqa: E402

cfgname = sys.argv[1] if len(sys.argv) > 1 else "C2"
raw, obs, flux, eflux = yst_problem(cfgname)
B = synth.CONFIGS[cfgname]["batch"]
eng = PayneEngine(nnio.normalize_spec_net(raw), obs=(obs, flux, eflux), b_max=B)
th = eng._theta(theta_full(synth.draw_candidates(B, seed=1)), eng.ncols)
eng.lnlike_batch(th)
eng.torch.cuda.synchronize()
st = np.zeros((B, 64), dtype=np.uint64)
fn = eng.lib.payne_diag_post_stamps
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
fn.restype = C.c_int
for rep in range(3):
    rc = fn(eng._ctx, th.data_ptr(), B, st.ctypes.data)
    assert rc == 0, eng.lib.payne_last_error(eng._ctx)
n = int(st[0, 0])
d = np.diff(st[:, 1:n + 1].astype(np.int64), axis=1)
med = np.median(d, axis=0)
names = ["setup+load"]
npass = 4 if cfgname == "C2" else None
print("stamps per block:", n, " total median cycles:", int(np.median(st[:, n].astype(np.int64) - st[:, 1].astype(np.int64))))
for i, m in enumerate(med):
    print("phase %2d  median %8d cycles  (p10 %8d  p90 %8d)" % (i, m, np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
