"""Joint spectrum + photometry likelihood step against spectrum only (GPU box): the photometric networks as a launch of their own
(variant 8192, PAYNE_V_SED_OWN_LAUNCH) and as extra workgroups of the first dense launch (default).  A third form, the
photometric kernel on a side stream beside the dense layers (fork / join by events), took 62.6 us against 51.7 us in line -- two
cross-queue dependencies cost more than the kernel they would hide -- and was dropped."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))     # the repository this file sits in
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from thepayne_amd import synth, nnio
from thepayne_amd.engine import PayneEngine, highav_coefficients
cfg = synth.CONFIGS["C2"]
raw = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], seed=0)
obs = synth.obs_grid(raw["wavelength"], cfg["nobs"])
phot = synth.make_phot_nets()
obs_phot = {f: (5.0 + 0.1 * i, 0.05) for i, f in enumerate(phot["filters"])}
B = 512
for with_phot, variant in ((False, 0), (True, 0)):
    eng = PayneEngine(nnio.normalize_spec_net(raw), obs=(obs, np.ones(len(obs)), np.full(len(obs), 0.01)),
                      phot=phot if with_phot else None, obs_phot=obs_phot if with_phot else None, photscale=True, b_max=B, variant=variant)
    th7 = synth.draw_candidates(B, seed=1)
    th = np.full((B, eng.ncols), np.nan); th[:, 0:6] = th7[:, 0:6]; th[:, 7] = th7[:, 6]
    if with_phot:
        th[:, eng.phot_off] = 0.0; th[:, eng.phot_off + 2] = 0.5
    t = eng._theta(th, eng.ncols)
    for _ in range(20): eng.lnlike_batch(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): eng.lnlike_batch(t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 300
    lnl = eng.lnlike_batch(t).cpu().numpy()
    print("phot" if with_phot else "spec", "variant", variant, "lnl[:2]", lnl[:2], "us/step %.2f" % (dt * 1e6), "filters", len(phot["filters"]) if with_phot else 0, "H", phot.get("H") if isinstance(phot, dict) else None)
    eng.close()
