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

from thepayne_amd import build, _lib  # noqa: E402

# STAMP_LIB: a stamped twin built by hand (build.build_variant(tag, ['-DPAYNE_STAMPS', ...])) instead of the diagnostic build
path = os.environ.get("STAMP_LIB") or build.build_diag()
os.environ["PAYNE_HIP_LIB"] = path
from thepayne_amd import synth, nnio  # noqa: E402
from thepayne_amd.engine import PayneEngine  # noqa: E402


def theta_full(theta7):
    """[B,7] sampled spectroscopic vectors -> [B, 12] ABI rows (NaN = absent)."""
    theta7 = np.atleast_2d(theta7)
    out = np.full((len(theta7), 12), np.nan)
    out[:, 0:6] = theta7[:, 0:6]
    out[:, 7] = theta7[:, 6]
    return out


def _norm(net):
    return net["_norm"] if "_norm" in net else nnio.normalize_spec_net(net)


def problem(cfg_name):
    """Synthetic net + observed spectrum made with the engine itself (a timing tool: no oracle here)."""
    cfg = synth.CONFIGS["C2" if cfg_name == "LinNet300" else cfg_name]
    if cfg_name == "LinNet300":              # the reference's default network at full width on the C2 shape (bench.py make_problem)
        raw_ = synth.make_torch_net("LinNet", npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=(300, 300, 300), seed=21)
        net = dict(raw_, _norm=nnio.normalize_spec_net(raw_, "LinNet"))
    else:
        net = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=300, seed=0)
    obs = synth.obs_grid(net["wavelength"], cfg["nobs"])
    e0 = PayneEngine(_norm(net), obs=(obs,), b_max=1)
    T = synth.TRUTH
    truth = theta_full(np.array([[T["Teff"], T["logg"], T["feh"], T["afe"], T["vrad"], T["vrot"], T["inst_R"]]]))
    clean = e0.predict_batch(truth, stage=2, fwhm_R=True).cpu().numpy()[0].astype(np.float64)
    e0.close()
    flux = clean + np.random.default_rng(0).normal(0, 0.01, len(obs))
    return net, obs, flux, np.full(len(obs), 0.01)


cfgname = sys.argv[1] if len(sys.argv) > 1 else "C2"
raw, obs, flux, eflux = problem(cfgname)
B = int(os.environ.get("STAMP_BATCH", synth.CONFIGS["C2" if cfgname == "LinNet300" else cfgname]["batch"]))
eng = PayneEngine(_norm(raw), obs=(obs, flux, eflux), b_max=B, variant=int(os.environ.get("STAMP_VARIANT", "0")))
th = eng._theta(theta_full(synth.draw_candidates(B, seed=1)), eng.ncols)
eng.lnlike_batch(th)
eng.torch.cuda.synchronize()
ROW = int(eng.lib.payne_diag_stamp_row())
st = np.zeros((B, ROW), dtype=np.uint64)
fn = eng.lib.payne_diag_post_stamps
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
fn.restype = C.c_int
for rep in range(3):
    rc = fn(eng._ctx, th.data_ptr(), B, st.ctypes.data)
    assert rc == 0, eng.lib.payne_last_error(eng._ctx)
n = int(st[0, 0])
whole = st[:, ROW - 1].astype(np.int64) - st[:, 1].astype(np.int64)
print("whole kernel per candidate (first stamp -> kernel end): median %d cycles" % np.median(whole))
if os.environ.get("PAYNE_DIAG_SPARSE"):
    sys.exit(0)
d = np.diff(st[:, 1:n + 1].astype(np.int64), axis=1)
med = np.median(d, axis=0)
names = ["setup+load"]
npass = 4 if cfgname == "C2" else None
print("stamps per block:", n, " total median cycles:", int(np.median(st[:, n].astype(np.int64) - st[:, 1].astype(np.int64))))
for i, m in enumerate(med):
    print("phase %2d  median %8d cycles  (p10 %8d  p90 %8d)" % (i, m, np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))

# ---- hidden-layer kernel: stamps 0 entry | 1 loads issued | 2 B tile + labels in LDS | 3 A tile (fused layer 0) |
#      4 MFMA done | 5 end
try:
    fh = eng.lib.payne_diag_hidden_stamps
    fh.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    fh.restype = C.c_int
    NB = 256
    hs = np.zeros((NB, 16), dtype=np.uint64)
    for rep in range(3):
        assert fh(eng._ctx, th.data_ptr(), B, hs.ctypes.data, NB) == 0
    used = hs[:, 5] > 0
    h = hs[used].astype(np.int64)
    print("hidden kernel: %d workgroups stamped" % used.sum())
    # (the counter is per XCD: stamps of different workgroups are not comparable, only the differences inside one)
    order = [0, 1, 2, 6, 3, 4, 5]          # slot 6 = first layer done (added later, between 2 and 3)
    for (a_, b_), name in zip(zip(order[:-1], order[1:]), ["loads issued", "labels in LDS", "first layer computed",
                                                           "weight tile stored", "MFMA done", "end"]):
        dlt = h[:, b_] - h[:, a_]
        print("  %-26s median %6d  p90 %6d" % (name, np.median(dlt), np.percentile(dlt, 90)))
    print("  whole workgroup median %d cycles" % np.median(h[:, 5] - h[:, 0]))
except AttributeError:
    pass

# ---- output-layer kernel (streaming 64x64x32): 0 entry | 1 first tiles in LDS | 2.. end of k-step | 15 end
try:
    NB = 512
    ds = np.zeros((NB, 16), dtype=np.uint64)
    for rep in range(3):
        assert fh(eng._ctx, th.data_ptr(), B, ds.ctypes.data, -NB) == 0
    used = ds[:, 15] > 0
    h = ds[used].astype(np.int64)
    print("output-layer kernel: %d workgroups stamped" % used.sum())
    nk = int((h[0, 2:15] > 0).sum())
    print("  prologue (first tiles)    median %6d" % np.median(h[:, 1] - h[:, 0]))
    for k in range(nk):
        print("  k-step %2d                 median %6d  p90 %6d" % (k, np.median(h[:, 2 + k] - h[:, 1 + k]), np.percentile(h[:, 2 + k] - h[:, 1 + k], 90)))
    print("  epilogue                  median %6d" % np.median(h[:, 15] - h[:, 1 + nk]))
    print("  whole workgroup median %d" % np.median(h[:, 15] - h[:, 0]))
except Exception as e:
    print("dense stamps failed:", e)
try:
    whole = h[:, 15] - h[:, 0]
    print("  whole workgroup p10 %d p50 %d p90 %d max %d" % tuple(np.percentile(whole, [10, 50, 90, 100])))
except Exception as e:
    print("dense stamps (2) failed:", e)
