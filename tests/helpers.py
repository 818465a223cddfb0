"""Shared builders for the tests: synthetic problems + oracle/engine glue."""
import numpy as np

import oracle as O
from thepayne_amd import synth

SPEC_PARS = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Inst_R']


def theta_full(theta7, npoly=0, pc=None, phot=None, vmic=None):
    """[B,7] sampled spectroscopic vectors -> [B, 8+npoly+4] ABI rows (NaN = absent)."""
    theta7 = np.atleast_2d(theta7)
    B = len(theta7)
    out = np.full((B, 8 + npoly + 4), np.nan)
    out[:, 0:6] = theta7[:, 0:6]
    out[:, 7] = theta7[:, 6]
    if vmic is not None:
        out[:, 6] = vmic
    if npoly:
        out[:, 8:8 + npoly] = pc
    if phot is not None:
        out[:, 8 + npoly:8 + npoly + phot.shape[1]] = phot
    return out


def yst_problem(cfg_name, H=300, seed=0, noise_seed=0, line_depth=0.02):
    cfg = synth.CONFIGS[cfg_name]
    net = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=H, seed=seed, line_depth=line_depth)
    obs = synth.obs_grid(net["wavelength"], cfg["nobs"])
    T = synth.TRUTH
    _, clean = O.getspec(net, Teff=T["Teff"], logg=T["logg"], feh=T["feh"], afe=T["afe"], rad_vel=T["vrad"],
                         rot_vel=T["vrot"], vmic=np.nan, inst_R=2.355 * T["inst_R"], outwave=obs)
    rng = np.random.default_rng(noise_seed)
    flux = clean + rng.normal(0, 0.01, len(obs))
    eflux = np.full(len(obs), 0.01)
    return net, obs, flux, eflux


def lnl_tol(ref):
    """|dlnL| <= 2e-6 |lnL| + 5e-3."""
    return 2e-6 * np.abs(ref) + 5e-3
