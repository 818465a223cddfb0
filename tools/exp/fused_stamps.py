"""Cycle stamps of payne_dense_fused_kernel (PAYNE_V_DENSE_FUSED, diagnostic build, C2): per workgroup
0 entry | 1 ticket known | 2 hidden tile computed | 3 tile published (stores drained, barrier) | 4 row block complete (poll) |
5 acquire done | 6 first operand stage landed | 7 k-loop done | 15 end.  Prints medians by role."""
import ctypes as C, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from thepayne_amd import build
os.environ["PAYNE_HIP_LIB"] = os.environ.get("STAMP_LIB") or build.build_diag()
from thepayne_amd import synth, nnio
from thepayne_amd.engine import PayneEngine
cfg = synth.CONFIGS["C2"]
B = cfg["batch"]
raw = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], seed=0)
obs = synth.obs_grid(raw["wavelength"], cfg["nobs"])
eng = PayneEngine(nnio.normalize_spec_net(raw), obs=(obs, np.ones(len(obs)), np.full(len(obs), 0.01)), b_max=B, variant=32768)
th7 = synth.draw_candidates(B, seed=1)
th = np.full((B, eng.ncols), np.nan); th[:, 0:6] = th7[:, 0:6]; th[:, 7] = th7[:, 6]
t = eng._theta(th, eng.ncols)
for _ in range(5):
    eng.lnlike_batch(t)
eng.torch.cuda.synchronize()
fh = eng.lib.payne_diag_hidden_stamps
fh.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]; fh.restype = C.c_int
NB = 256
hs = np.zeros((NB, 16), dtype=np.uint64)
for rep in range(3):
    assert fh(eng._ctx, t.data_ptr(), B, hs.ctypes.data, -NB) == 0
h = hs.astype(np.int64)
hid = h[:, 2] > 0
print("workgroups: %d with a hidden tile, %d without" % (hid.sum(), (~hid).sum()))
def med(sel, a, b): return int(np.median(h[sel, b] - h[sel, a]))
print("with a hidden tile:")
for a, b, name in [(0, 1, "ticket (atomic + barrier)"), (1, 2, "hidden tile"), (2, 3, "publish (stores drained + barrier)"), (3, 4, "wait for the row block"),
                   (4, 5, "acquire + barrier"), (5, 6, "first stages land"), (6, 7, "k-loop"), (7, 15, "stores"), (0, 15, "WHOLE")]:
    print("  %-36s %7d" % (name, med(hid, a, b)))
print("without:")
for a, b, name in [(0, 1, "ticket (atomic + barrier)"), (1, 4, "wait for the row block"), (4, 5, "acquire + barrier"), (5, 6, "first stages land"),
                   (6, 7, "k-loop"), (7, 15, "stores"), (0, 15, "WHOLE")]:
    print("  %-36s %7d" % (name, med(~hid, a, b)))
