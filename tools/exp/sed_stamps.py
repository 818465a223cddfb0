"""Cycle stamps of the photometric tiles inside the hidden-layer launch (diagnostic build, C3):
0 entry | 1 requests issued | 2 theta row -> labels | 3 operands in LDS | 4 layer 1 | 6 layer 2 | 5 end; GEMM tiles: 0 entry, 5 end."""
import ctypes as C, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from thepayne_amd import build
os.environ["PAYNE_HIP_LIB"] = os.environ.get("STAMP_LIB") or build.build_diag()
from thepayne_amd import synth, nnio
from thepayne_amd.engine import PayneEngine
cfg = synth.CONFIGS["C3"]
B = cfg["batch"]
raw = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], seed=0)
obs = synth.obs_grid(raw["wavelength"], cfg["nobs"])
phot = synth.make_phot_nets()
eng = PayneEngine(nnio.normalize_spec_net(raw), obs=(obs, np.ones(len(obs)), np.full(len(obs), 0.01)), phot=phot,
                  obs_phot=synth.c3_obs_phot(phot["filters"]), photscale=True, b_max=B)
th9 = synth.draw_candidates_c3(B, seed=1)
th = np.full((B, eng.ncols), np.nan); th[:, 0:6] = th9[:, 0:6]; th[:, 7] = th9[:, 6]; th[:, 8] = th9[:, 7]; th[:, 10] = th9[:, 8]
t = eng._theta(th, eng.ncols)
fh = eng.lib.payne_diag_hidden_stamps
fh.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]; fh.restype = C.c_int
NB = 512
hs = np.zeros((NB, 16), dtype=np.uint64)
for rep in range(3):
    assert fh(eng._ctx, t.data_ptr(), B, hs.ctypes.data, NB) == 0
h = hs.astype(np.int64)
gemm = h[:160]; sed = h[162:][h[162:, 5] > 0]
t0 = h[h[:, 0] > 0, 0].min()
print("GEMM tiles: start %d..%d, end median %d max %d (cycles after the first start)" % (gemm[:, 0].min() - t0, gemm[:, 0].max() - t0, np.median(gemm[:, 5]) - t0, gemm[:, 5].max() - t0))
print("SED tiles : %d, start %d..%d, end median %d max %d" % (len(sed), sed[:, 0].min() - t0, sed[:, 0].max() - t0, np.median(sed[:, 5]) - t0, sed[:, 5].max() - t0))
for a, b, name in [(0, 1, "requests issued"), (1, 2, "theta row -> labels (pow/log10)"), (2, 3, "operands in LDS"), (3, 4, "layer 1"), (4, 6, "layer 2"), (6, 5, "layer 3 + magnitude")]:
    d = sed[:, b] - sed[:, a]
    print("  %-34s median %6d  p90 %6d" % (name, np.median(d), np.percentile(d, 90)))
print("  whole tile median %d" % np.median(sed[:, 5] - sed[:, 0]))
