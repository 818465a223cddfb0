"""Cycle stamps of payne_post_big_kernel (diagnostic build) for the 65 536-pixel configuration: BATCH candidates,
PAYNE_BIG_TILED=0/1.  Prints the median cycles of every barrier interval."""
import ctypes as C, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from thepayne_amd import build
os.environ["PAYNE_HIP_LIB"] = os.environ.get("STAMP_LIB") or build.build_diag()
from thepayne_amd import synth, nnio
from thepayne_amd.engine import PayneEngine
cfg = synth.CONFIGS["C5"]
B = int(os.environ.get("BATCH", "256"))
net = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=300, seed=0)
obs = synth.obs_grid(net["wavelength"], cfg["nobs"], inset=0.0005, relative=True)
if os.environ.get("REVERSE"):          # descending wavelengths: the observed grid as a phase of its own (no chip_conv_obs)
    obs = obs[::-1].copy()
eng = PayneEngine(nnio.normalize_spec_net(net), obs=(obs, np.ones(len(obs)), np.full(len(obs), 0.01)), b_max=B)
th7 = synth.draw_candidates(B, seed=1)
th = np.full((B, 12), np.nan); th[:, 0:6] = th7[:, 0:6]; th[:, 7] = th7[:, 6]
t = eng._theta(th, eng.ncols)
eng.lnlike_batch(t); eng.torch.cuda.synchronize()
ROW = int(eng.lib.payne_diag_stamp_row())
st = np.zeros((B, ROW), dtype=np.uint64)
fn = eng.lib.payne_diag_post_stamps
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]; fn.restype = C.c_int
for _ in range(2):
    assert fn(eng._ctx, t.data_ptr(), B, st.ctypes.data) == 0
n = int(st[0, 0])
d = np.diff(st[:, 1:n + 1].astype(np.int64), axis=1)
med = np.median(d, axis=0)
print("phases", n - 1, "total median", int(med.sum()))
print(" ".join("%d" % (m / 1000) for m in med), "(k cycles)")
