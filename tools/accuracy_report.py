"""Measured accuracy of the HIP path against the golden vectors frozen from the reference (GPU box):
max |d lnL| over the 512 C2 prior draws and max |d flux| over the getspec grid.  (Reads tests/golden only.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from thepayne_amd import synth, nnio                         # noqa: E402
from thepayne_amd.engine import PayneEngine                  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
g = np.load(os.path.join(G, "g4_lnlike_c2.npz"))
cfg = synth.CONFIGS["C2"]
net = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], seed=0)
eng = PayneEngine(nnio.normalize_spec_net(net), obs=(g["obs_wave"], g["obs_flux"], g["obs_eflux"]), b_max=512)
th = np.full((512, eng.ncols), np.nan)
th[:, 0:6] = g["theta"][:, 0:6]
th[:, 7] = g["theta"][:, 6]
got = eng.lnlike_batch(th).cpu().numpy()
ref = g["lnlike"]
ok = np.isfinite(ref)
print("C2, 512 prior draws: max |d lnL| = %.3g (|lnL| %.0f .. %.0f), max relative %.3g; NaN pattern equal: %s"
      % (np.abs(got[ok] - ref[ok]).max(), np.abs(ref[ok]).min(), np.abs(ref[ok]).max(),
         (np.abs(got[ok] - ref[ok]) / np.abs(ref[ok])).max(), np.array_equal(np.isnan(got), np.isnan(ref))))
g2 = np.load(os.path.join(G, "g2_getspec.npz"))
net2 = synth.make_yst_net(npix=1024, H=64, seed=5, line_depth=0.3)
e2 = PayneEngine(nnio.normalize_spec_net(net2), obs=(g2["obs_wave"],), b_max=64)
rows = g2["theta_rows"]
t2 = np.full((len(rows), e2.ncols), np.nan)
t2[:, 0:4] = g2["labels"]
t2[:, 4], t2[:, 5], t2[:, 7] = rows[:, 0], rows[:, 1], 2.355 * rows[:, 2]
f = e2.predict_batch(t2, stage=2).cpu().numpy().astype(np.float64)
r = g2["final"]
print("getspec grid (%d settings x %d pixels): max |d flux| = %.3g; NaN pattern equal: %s"
      % (len(rows), f.shape[1], np.nanmax(np.abs(f - r)), np.array_equal(np.isnan(f), np.isnan(r))))
