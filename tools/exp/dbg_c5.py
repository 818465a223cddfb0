import ctypes as C, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from thepayne_amd import build
if os.environ.get("DIAG", "1") == "1":
    os.environ["PAYNE_HIP_LIB"] = os.path.join(ROOT, "thepayne_amd", "libpayne_hip_diag.so")
from thepayne_amd import synth, nnio
from thepayne_amd.engine import PayneEngine
cfg = synth.CONFIGS["C5"]
B = int(os.environ.get("BATCH", "64"))
net = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=300, seed=0)
obs = synth.obs_grid(net["wavelength"], cfg["nobs"], inset=0.0005, relative=True)
flux = np.ones(len(obs)); eflux = np.full(len(obs), 0.01)
eng = PayneEngine(nnio.normalize_spec_net(net), obs=(obs, flux, eflux), b_max=B)
th7 = synth.draw_candidates(B, seed=1)
th = np.full((B, 12), np.nan); th[:, 0:6] = th7[:, 0:6]; th[:, 7] = th7[:, 6]
t = eng._theta(th, eng.ncols)
print("lnlike...", flush=True)
l = eng.lnlike_batch(t); eng.torch.cuda.synchronize(); print("ok", float(l[0]), flush=True)
if os.environ.get("DIAG", "1") == "1":
    st = np.zeros((B, 64), dtype=np.uint64)
    fn = eng.lib.payne_diag_post_stamps
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]; fn.restype = C.c_int
    print("stamps...", flush=True)
    rc = fn(eng._ctx, t.data_ptr(), B, st.ctypes.data); print("rc", rc, flush=True)
    n = int(st[0, 0]); print("n", n)
    d = np.diff(st[:, 1:n + 1].astype(np.int64), axis=1)
    for i, m in enumerate(np.median(d, axis=0)):
        print("phase %2d  median %8d cycles" % (i, m))
