"""Seeded synthetic inputs for the hot path.

There is no network for trained ANNs or survey spectra, so benchmarks and
tests use random-init networks of the reference's architectures on the
reference's wavelength-grid construction (geometric grid, 3 px per sigma-R,
Payne/utils/readc3k.py:441-447).  Host-only numpy; no oracle, no GPU.
"""
import numpy as np

SPEC_LABEL_MIN = np.array([3500.0, 0.0, -2.5, -0.2])
SPEC_LABEL_MAX = np.array([8000.0, 5.5, 0.5, 0.6])

CONFIGS = {
    # name: (npix, lambda0, R_fwhm, nobs, batch)
    "C2": dict(npix=4096, lam0=5150.0, R=32000.0, nobs=3600, batch=512),
    # C3: C2 + photometry in 7 filters (Bessell_BVRI, 2MASS_JHKs), H = 64 sigmoid nets (make_phot_nets seed 1),
    # observed magnitudes 5.0 +- 0.05, the `photscale` parametrisation: log(A) U[-3, 7], Av U[0, 1]
    "C3": dict(npix=4096, lam0=5150.0, R=32000.0, nobs=3600, batch=512, phot=True),
    # not a BASELINE config: C2's grid cut to a length that is not a power of two (what a trained network has: the reference builds
    # its grid from a wavelength range and a resolution, Payne/utils/readc3k.py:441-447, and resamples to 2^k itself)
    "C2r": dict(npix=3600, lam0=5150.0, R=32000.0, nobs=3200, batch=512),
    "C5": dict(npix=65536, lam0=4000.0, R=100000.0, nobs=60000, batch=2048),
    # not a BASELINE config: spectra between the LDS-resident kernel (<= 16 384 points) and C5 -- an R ~ 60k grid of 32 768 pixels
    # (4500-4861 AA), 30 000 observed pixels, 1024 candidates (the reference's own demo spectrum has 25 600 pixels, demo/runPayne.py:43-50)
    "C32k": dict(npix=32768, lam0=4500.0, R=60000.0, nobs=30000, batch=1024),
    "tiny": dict(npix=256, lam0=5150.0, R=32000.0, nobs=200, batch=16),
    "small": dict(npix=1024, lam0=5150.0, R=32000.0, nobs=900, batch=32),
}


def ann_wavelength(npix, lam0, R_fwhm):
    """Geometric grid lam_i = lam0 (1 + 1/(3 Rsigma))^i with Rsigma = R*2.3548."""
    Rsig = R_fwhm * 2.3548
    return lam0 * (1.0 + 1.0 / (3.0 * Rsig)) ** np.arange(npix), Rsig


def make_yst_net(npix=4096, lam0=5150.0, R_fwhm=32000.0, H=300, seed=0, D=4, line_depth=0.02):
    """Random YST1 network: D -> H -> H -> npix, fp32 weights, keys as in the
    reference's HDF5 file (Payne/predict/ystpred.py:22-38)."""
    rng = np.random.default_rng(seed)
    wave, Rsig = ann_wavelength(npix, lam0, R_fwhm)
    net = {
        "kind": "YST1",
        "w_array_0": rng.normal(0, 0.5, (H, D)).astype(np.float32),
        "b_array_0": rng.normal(0, 0.1, H).astype(np.float32),
        "w_array_1": rng.normal(0, np.sqrt(1.0 / H), (H, H)).astype(np.float32),
        "b_array_1": rng.normal(0, 0.1, H).astype(np.float32),
        "w_array_2": rng.normal(0, line_depth / np.sqrt(H), (npix, H)).astype(np.float32),
        "b_array_2": (0.95 + 0.02 * rng.normal(0, 1, npix)).astype(np.float32),
        "x_min": SPEC_LABEL_MIN[:D].copy() if D <= 4 else np.append(SPEC_LABEL_MIN, 0.5),
        "x_max": SPEC_LABEL_MAX[:D].copy() if D <= 4 else np.append(SPEC_LABEL_MAX, 2.5),
        "wavelength": wave,
        "resolution": float(Rsig),
    }
    return net


def make_cont_net(npix=600, lam_lo=5145.0, lam_hi=5250.0, H=32, seed=11, D=4):
    """Random continuum network in the same YST1 container (``Cnnpath`` of ystpred.PayneSpecPredict,
    Payne/predict/ystpred.py:81-85): a smooth positive F_nu continuum on its own, coarser, linear grid."""
    rng = np.random.default_rng(seed)
    wave = np.linspace(lam_lo, lam_hi, npix)
    x = (wave - wave.mean()) / (wave.max() - wave.min())
    net = {
        "kind": "YST1",
        "w_array_0": rng.normal(0, 0.5, (H, D)).astype(np.float32),
        "b_array_0": rng.normal(0, 0.1, H).astype(np.float32),
        "w_array_1": rng.normal(0, np.sqrt(1.0 / H), (H, H)).astype(np.float32),
        "b_array_1": rng.normal(0, 0.1, H).astype(np.float32),
        "w_array_2": (0.05 / np.sqrt(H) * np.outer(1.0 + 0.5 * x, rng.normal(0, 1, H))).astype(np.float32),
        "b_array_2": (3.0e-5 * (1.0 + 0.3 * x + 0.2 * x * x)).astype(np.float32) + np.float32(1.0e-5),
        "x_min": SPEC_LABEL_MIN[:D].copy() if D <= 4 else np.append(SPEC_LABEL_MIN, 0.5),
        "x_max": SPEC_LABEL_MAX[:D].copy() if D <= 4 else np.append(SPEC_LABEL_MAX, 2.5),
        "wavelength": wave,
        "resolution": 1000.0,
    }
    net["w_array_2"] *= np.float32(1.0e-5)          # output ~ 4e-5 +- 1e-6: positive, F_nu-like magnitudes
    return net


def make_torch_net(kind, npix=1024, lam0=5150.0, R_fwhm=32000.0, H=(64, 48, 32), seed=0, D=4):
    """Random LinNet / SMLP state dict with the reference's key names
    (Payne/train/NNmodels.py:58-63, 92-168)."""
    rng = np.random.default_rng(seed)
    wave, Rsig = ann_wavelength(npix, lam0, R_fwhm)
    H1, H2, H3 = H
    net = {"kind": kind, "xmin": SPEC_LABEL_MIN[:D].copy(), "xmax": SPEC_LABEL_MAX[:D].copy(),
           "wavelength": wave, "resolution": float(Rsig)}

    def lin(name, nout, nin, wscale, bmean=0.0, bscale=0.1):
        net[name + ".weight"] = rng.normal(0, wscale, (nout, nin)).astype(np.float32)
        net[name + ".bias"] = (bmean + bscale * rng.normal(0, 1, nout)).astype(np.float32)

    if kind == "LinNet":
        dims = [(H1, D), (H1, H1), (H2, H1), (H2, H2), (H3, H2)]
        for i, (no, ni) in enumerate(dims, start=1):
            lin("lin%d" % i, no, ni, 2.0 / np.sqrt(ni))
        lin("lin6", npix, H3, 0.1 / np.sqrt(H3), bmean=0.9, bscale=0.02)
    elif kind == "SMLP":
        dims = [(H1, D), (H2, H1), (H3, H2)]
        for i, (no, ni) in zip((0, 2, 4), dims):
            lin("features.%d" % i, no, ni, 1.0 / np.sqrt(ni))
        lin("features.6", npix, H3, 0.05 / np.sqrt(H3), bmean=0.95, bscale=0.02)
    else:
        raise ValueError(kind)
    return net


PHOT_FILTERS = ['Bessell_B', 'Bessell_V', 'Bessell_R', 'Bessell_I', '2MASS_J', '2MASS_H', '2MASS_Ks']
PHOT_LABEL_MIN = np.array([2500.0, -1.0, -4.0, -0.2, 0.0, 2.0])
PHOT_LABEL_MAX = np.array([20000.0, 5.5, 0.5, 0.6, 5.0, 5.0])


def make_phot_nets(filters=PHOT_FILTERS, H=64, seed=1):
    """Random per-filter 6->H->H->1 sigmoid nets stacked like ``fastANN``
    (Payne/predict/photANN.py:97-106): w1[F,H,6] b1[F,H,1] w2[F,H,H] b2[F,H,1]
    w3[F,1,H] b3[F,1,1], fp32; xmin/xmax fp64 [6]."""
    rng = np.random.default_rng(seed)
    F = len(filters)
    return {
        "filters": list(filters),
        "w1": rng.normal(0, 1.0, (F, H, 6)).astype(np.float32),
        "b1": rng.normal(0, 0.3, (F, H, 1)).astype(np.float32),
        "w2": rng.normal(0, 1.0 / np.sqrt(H), (F, H, H)).astype(np.float32),
        "b2": rng.normal(0, 0.3, (F, H, 1)).astype(np.float32),
        "w3": rng.normal(0, 1.0 / np.sqrt(H), (F, 1, H)).astype(np.float32),
        "b3": rng.normal(0, 0.5, (F, 1, 1)).astype(np.float32),
        "xmin": PHOT_LABEL_MIN.copy(),
        "xmax": PHOT_LABEL_MAX.copy(),
    }


TRUTH = dict(Teff=5770.0, logg=4.44, feh=0.0, afe=0.0, vrad=10.0, vrot=3.0, inst_R=28800.0)


def obs_grid(wave, nobs, inset=3.0, relative=False):
    """Observed wavelength grid: linspace inset from the ANN grid's ends."""
    if relative:
        return np.linspace(wave[0] * (1 + inset), wave[-1] * (1 - inset), nobs)
    return np.linspace(wave[0] + inset, wave[-1] - inset, nobs)


def demo_priordict():
    """Prior ranges mirroring demo/runPayne.py:123-141."""
    return {
        'Teff': {'pv_uniform': [4000.0, 8000.0]},
        'log(g)': {'pv_uniform': [4.0, 5.5]},
        '[Fe/H]': {'pv_uniform': [-0.1, 0.1]},
        '[a/Fe]': {'pv_uniform': [-0.1, 0.1]},
        'Vrad': {'pv_uniform': [9.0, 11.0]},
        'Vrot': {'pv_uniform': [0.0, 5.0]},
        'Inst_R': {'pv_tgaussian': [25000.0, 37000.0, 28800.0, 1000.0]},
    }


def c3_obs_phot(filters=PHOT_FILTERS):
    """C3's observed photometry: every filter 5.0 +- 0.05 mag."""
    return {f: (5.0, 0.05) for f in filters}


def c3_priordict():
    """C2's priors + the photometric block of C3 (photscale): log(A) U[-3, 7], Av U[0, 1]."""
    d = demo_priordict()
    d['log(A)'] = {'pv_uniform': [-3.0, 7.0]}
    d['Av'] = {'pv_uniform': [0.0, 1.0]}
    return d


def draw_candidates_c3(B, seed=1):
    """theta[B, 9] = C2's seven columns + (log(A), Av) drawn through c3_priordict's boxes from the same stream."""
    th7 = draw_candidates(B, seed=seed)
    u = np.random.default_rng(seed + 7919).uniform(size=(B, 2))
    return np.column_stack([th7, -3.0 + 10.0 * u[:, 0], u[:, 1]])


def draw_candidates(B, seed=1, ndim=7):
    """theta[B, 7] = (Teff, logg, FeH, aFe, Vrad, Vrot, Inst_R) drawn through
    the demo priors from u ~ U(0,1), seed fixed."""
    from scipy.stats import truncnorm
    rng = np.random.default_rng(seed)
    u = rng.uniform(size=(B, ndim))
    th = np.empty((B, 7))
    th[:, 0] = 4000.0 + 4000.0 * u[:, 0]
    th[:, 1] = 4.0 + 1.5 * u[:, 1]
    th[:, 2] = -0.1 + 0.2 * u[:, 2]
    th[:, 3] = -0.1 + 0.2 * u[:, 3]
    th[:, 4] = 9.0 + 2.0 * u[:, 4]
    th[:, 5] = 5.0 * u[:, 5]
    a, b = (25000.0 - 28800.0) / 1000.0, (37000.0 - 28800.0) / 1000.0
    th[:, 6] = truncnorm.ppf(u[:, 6], a, b, loc=28800.0, scale=1000.0)
    return th
