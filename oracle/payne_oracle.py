"""numpy fp64 restatement of the reference hot path (see oracle/__init__.py).

Reference paths are relative to /root/reference.  This is a checker, written
for clarity, one theta per call like the reference; it is never the product.
"""
import numpy as np
from scipy.special import j1 as _bessel_j1

__all__ = [
    "CKMS", "SIGMA_TO_FWHM", "C_KMS_DOPPLER",
    "leaky_relu", "yst_encode", "yst_forward", "torchnet_forward", "ann_forward",
    "mask_range", "resample_pow2", "taper_vsini", "taper_gauss", "fft_convolve",
    "smooth_vsini", "smooth_R", "smooth_lsf", "smoothspec_offpath", "smoothspec_fft", "getspec", "polycalc", "genspec",
    "chi2_spec", "chi2_spec_loop", "fastann_forward", "highav_offset", "sed_mags",
    "genphot", "genphot_scaled", "OracleLikelihood", "lnprobfn",
]

# Payne/utils/smoothing.py:16-17 -- the smoothing code's own speed of light
CKMS = 2.998e5
SIGMA_TO_FWHM = 2.355
# Payne/predict/ystpred.py:11-12 -- scipy.constants.c/1000, used for the Doppler shift only
C_KMS_DOPPLER = 299792.458


# --------------------------------------------------------------------------
# ANN forward passes
# --------------------------------------------------------------------------
def leaky_relu(z):
    """Payne/predict/ystpred.py:41-45."""
    return z * (z > 0) + 0.01 * z * (z < 0)


def yst_encode(net, x):
    """Payne/predict/ystpred.py:47-50 : (x-xmin)/(xmax-xmin) - 0.5."""
    x = np.array(x, dtype=np.float64)
    return (x - net["x_min"]) / (net["x_max"] - net["x_min"]) - 0.5


def yst_forward(net, x):
    """YST1 ``Net.eval`` (Payne/predict/ystpred.py:52-58).

    ``net`` holds w_array_{0,1,2}, b_array_{0,1,2}, x_min, x_max (x_min/x_max
    already carrying the Teff*1000 fix of ystpred.py:76-79).  fp32 weights are
    promoted by the fp64 labels, so the arithmetic is fp64.
    """
    xs = yst_encode(net, x)
    h1 = net["w_array_0"] @ xs + net["b_array_0"]
    h2 = net["w_array_1"] @ leaky_relu(h1) + net["b_array_1"]
    return net["w_array_2"] @ leaky_relu(h2) + net["b_array_2"]


def _sigmoid32(a):
    return (1.0 / (1.0 + np.exp(-a))).astype(np.float32)


def torchnet_forward(net, x):
    """``ANN.eval`` for LinNet / SMLP (Payne/predict/predictspec.py:61-74,
    Payne/train/NNmodels.py:92-168): encode in fp64 numpy, cast to fp32,
    fp32 forward (torch), fp32 output."""
    x = np.asarray(x, dtype=np.float64)
    xs = ((x - net["xmin"]) / (net["xmax"] - net["xmin"]) - 0.5).astype(np.float32)
    kind = net["kind"]
    if kind == "LinNet":
        h = xs
        for i in range(1, 6):
            h = _sigmoid32(net["lin%d.weight" % i] @ h + net["lin%d.bias" % i])
        return (net["lin6.weight"] @ h + net["lin6.bias"]).astype(np.float32)
    if kind == "SMLP":
        h = xs
        for i in (0, 2, 4):
            z = net["features.%d.weight" % i] @ h + net["features.%d.bias" % i]
            h = np.where(z > 0, z, np.float32(0.01) * z).astype(np.float32)
        return (net["features.6.weight"] @ h + net["features.6.bias"]).astype(np.float32)
    raise ValueError(kind)


def ann_forward(net, x):
    if net.get("kind", "YST1") == "YST1":
        return yst_forward(net, x)
    return torchnet_forward(net, x)


# --------------------------------------------------------------------------
# smoothing (the two branches of smoothspec the path uses)
# --------------------------------------------------------------------------
def mask_range(wave, width, outwave, nsigma_pad=20.0):
    """``mask_wave`` non-linear branch (Payne/utils/smoothing.py:631-647).

    Returns the boolean mask ``wlim0 < wave < wlim1`` (strict).  With
    ``outwave is None`` the limits are [0, inf] scaled, i.e. everything.
    """
    if outwave is not None:
        wlim = np.array([outwave.min(), outwave.max()])
    else:
        wlim = np.array([0.0, np.inf])
    wlim = wlim * (1 + nsigma_pad / width * np.array([-1, 1]))
    return (wave > wlim[0]) & (wave < wlim[1])


def resample_pow2(wave, spec):
    """``resample_wave`` log branch (smoothing.py:649-668)."""
    wmin, wmax = wave.min(), wave.max()
    nnew = int(2.0 ** (np.ceil(np.log2(len(wave)))))
    w = np.exp(np.linspace(np.log(wmin), np.log(wmax), nnew))
    return w, np.interp(w, wave, spec)


def taper_vsini(n, dv, sigma):
    """Fourier taper of ``smooth_fft_vsini`` (smoothing.py:610-620)."""
    ss = np.fft.rfftfreq(n, d=dv)
    ss[0] = 0.01
    ub = 2.0 * np.pi * sigma * ss
    sb = _bessel_j1(ub) / ub - 3 * np.cos(ub) / (2 * ub ** 2) + 3.0 * np.sin(ub) / (2 * ub ** 3)
    sb[0] = 1.0
    return sb


def taper_gauss(n, dv, sigma):
    """Fourier taper of ``smooth_fft`` (smoothing.py:597-601); computed
    before the ss[0] hack so the DC gain is exactly 1."""
    ss = np.fft.rfftfreq(n, d=dv)
    return np.exp(-2 * (np.pi ** 2) * (sigma ** 2) * (ss ** 2))


def fft_convolve(spec, taper):
    """rfft * taper -> irfft: circular convolution, no padding
    (smoothing.py:602-608 / 622-629)."""
    return np.fft.irfft(np.fft.rfft(spec) * taper)


def _grid_dv(w):
    """smoothing.py:276-279: dv = ckms * median(diff(ln w))."""
    return CKMS * np.median(np.diff(np.log(w)))


def smooth_vsini(wave, spec, vrot):
    """smoothspec(..., smoothtype='vsini', outwave=None, inres=0.0)
    (smoothing.py:93-100, 131-169, 293-312).  Output on ``wave``."""
    mask = mask_range(wave, CKMS / vrot, None)          # all True
    w, s = wave[mask], np.nan_to_num(spec[mask], nan=1.0)
    sigma = np.sqrt(vrot ** 2 - 0.0 ** 2)
    wr, sr = resample_pow2(w, s)
    conv = fft_convolve(sr, taper_vsini(len(sr), _grid_dv(wr), sigma))
    return np.interp(wave, wr, conv, left=np.nan, right=np.nan)


def smooth_R(wave, spec, Rsigma, outwave, R_ann, return_parts=False):
    """smoothspec(..., smoothtype='R', outwave=obs, inres=R_ann)
    (smoothing.py:103-115, 131-169, 252-291)."""
    sigma_out = CKMS / Rsigma
    inres = CKMS / R_ann
    mask = mask_range(wave, Rsigma, outwave)
    w, s = wave[mask], np.nan_to_num(spec[mask], nan=1.0)
    if outwave is None:                              # smoothing.py:140-141: back onto the input grid
        outwave = wave
    with np.errstate(invalid="ignore"):
        sigma = np.sqrt(sigma_out ** 2 - inres ** 2)
    wr, sr = resample_pow2(w, s)
    conv = fft_convolve(sr, taper_gauss(len(sr), _grid_dv(wr), sigma))
    out = np.interp(outwave, wr, conv, left=np.nan, right=np.nan)
    if return_parts:
        first = int(np.argmax(mask)) if mask.any() else -1
        return out, dict(first=first, count=int(mask.sum()), nfft=len(sr))
    return out


def _trapz(y, x):
    """numpy.trapz (renamed trapezoid in numpy 2): sum of (x[k+1] - x[k]) (y[k] + y[k+1]) / 2."""
    return np.sum(np.diff(x) * (y[1:] + y[:-1]) / 2.0)


def smoothspec_offpath(wave, spec, resolution, outwave=None, smoothtype='vel', fftsmooth=True, **kw):
    """The branches of ``smoothspec`` (Payne/utils/smoothing.py:19-169) that the sampler's path does not take:
    ``fftsmooth=False`` -> smooth_vel (:171-210) / smooth_wave (:339-393) / smooth_lsf (:435-480), and smoothtype
    'lambda' with FFTs -> smooth_wave_fft (:395-433).  Units, mask and nan_to_num as smoothspec does them (:89-138)."""
    wave, spec = np.asarray(wave, dtype=np.float64), np.asarray(spec, dtype=np.float64)
    inres = kw.get('inres', 0)
    if smoothtype in ('vel', 'R'):
        sigma = float(resolution) if smoothtype == 'vel' else CKMS / float(resolution)
        width, linear = CKMS / sigma, False
        if smoothtype == 'R' and 'inres' in kw:
            inres = CKMS / kw['inres']
    elif smoothtype == 'lambda':
        sigma, width, linear = resolution, resolution, True
    elif smoothtype == 'lsf':
        sigma, width, linear = resolution, 100, True
    else:
        raise ValueError(smoothtype)
    if outwave is not None:
        wlim = np.array([np.min(outwave), np.max(outwave)], dtype=np.float64)
    else:
        wlim = np.array([kw.get('min_wave_smooth', 0), kw.get('max_wave_smooth', np.inf)], dtype=np.float64)
    wlim = wlim + 20.0 * width * np.array([-1, 1]) if linear else wlim * (1 + 20.0 / width * np.array([-1, 1]))
    mask = (wave > wlim[0]) & (wave < wlim[1])
    w, s = wave[mask], np.nan_to_num(spec[mask], nan=1.0)
    if outwave is None:
        outwave = wave
    outwave = np.asarray(outwave, dtype=np.float64)
    nsigma = kw.get('nsigma', 10)
    if smoothtype == 'lsf':                                          # smooth_lsf
        if sigma is None:
            return np.interp(outwave, w, s)
        sig = np.interp(outwave, wave, resolution)
        dw = np.gradient(w)
        kernel = outwave[:, None] - w[None, :]
        kernel = (1 / (sig * np.sqrt(np.pi * 2))[:, None] * np.exp(-kernel ** 2 / (2 * sig[:, None] ** 2)) * dw[None, :])
        kernel = kernel / kernel.sum(axis=1)[:, None]
        return np.dot(kernel, s)
    if smoothtype == 'lambda' and fftsmooth:                         # smooth_wave_fft
        nnew = int(2.0 ** (np.ceil(np.log2(len(w)))))
        wg = np.linspace(w.min(), w.max(), nnew)
        sg = np.interp(wg, w, s)
        sig = np.sqrt(sigma ** 2 - inres ** 2)
        dwg = np.median(np.diff(wg))
        return np.interp(outwave, wg, fft_convolve(sg, taper_gauss(len(sg), dwg, sig)))
    if smoothtype == 'lambda':                                       # smooth_wave
        if inres <= 0:
            sq = sigma ** 2
        elif kw.get('in_vel', False):
            sq = sigma ** 2 - (w / inres) ** 2
        else:
            sq = sigma ** 2 - inres ** 2
        if np.any(sq < 0):
            raise ValueError("Desired wavelength sigma is lower than the value possible for this input spectrum.")
        se = np.sqrt(sq)
        xs = lambda wo: (w - wo) / se                                # noqa: E731
    else:                                                            # smooth_vel
        se = np.sqrt(sigma ** 2 - inres ** 2) / CKMS
        lnw = np.log(w)
        xs = lambda wo: (np.log(wo) - lnw) / se                      # noqa: E731
    flux = np.zeros(len(outwave))
    for i, wo in enumerate(outwave):
        x = xs(wo)
        _s = s
        if nsigma > 0:
            good = np.abs(x) < nsigma
            x, _s = x[good], s[good]
        f = np.exp(-0.5 * x ** 2)
        with np.errstate(all="ignore"):
            flux[i] = _trapz(f * _s, x) / _trapz(f, x)
    return flux


def smoothspec_fft(wave, spec, resolution, outwave=None, smoothtype='vel', inres=None, min_wave_smooth=0, max_wave_smooth=np.inf):
    """``smoothspec`` with ``fftsmooth=True`` for smoothtype 'vel' / 'R' / 'vsini' / 'lsf' in general (Payne/utils/smoothing.py:19-169):
    any ``outwave``, ``inres``, and -- with ``outwave=None`` -- the input restricted by ``min_wave_smooth`` / ``max_wave_smooth`` with the
    result back on all of ``wave`` (:131-141).  smooth_vel_fft :252-291, smooth_vsini_fft :293-312, smooth_lsf_fft :482-586."""
    wave, spec = np.asarray(wave, dtype=np.float64), np.asarray(spec, dtype=np.float64)
    if smoothtype in ('vel', 'vsini'):
        sigma, width, linear = float(resolution), CKMS / float(resolution), False
        inres_v = 0.0 if inres is None else float(inres)
    elif smoothtype == 'R':
        sigma, width, linear = CKMS / float(resolution), float(resolution), False
        inres_v = 0.0 if inres is None else CKMS / float(inres)             # :110-115
    elif smoothtype == 'lsf':
        sigma, width, linear = np.asarray(resolution, dtype=np.float64), 100, True
    else:
        raise ValueError(smoothtype)
    if outwave is not None:                                                 # mask_wave, :631-647
        outwave = np.asarray(outwave, dtype=np.float64)
        wlim = np.array([outwave.min(), outwave.max()])
    else:
        wlim = np.squeeze(np.array([min_wave_smooth, max_wave_smooth])).astype(np.float64)
    wlim = wlim + 20.0 * width * np.array([-1, 1]) if linear else wlim * (1 + 20.0 / width * np.array([-1, 1]))
    mask = (wave > wlim[0]) & (wave < wlim[1])
    w, s = wave[mask], np.nan_to_num(spec[mask], nan=1.0)
    if outwave is None:
        outwave = wave
    if smoothtype == 'lsf':                                                 # :143-147, 482-586 (the vector on the INPUT grid, masked with it)
        sig = sigma[mask]
        dw = np.gradient(w)
        cdf = np.cumsum(dw / sig)
        cdf /= cdf.max()
        x_per_sigma = np.nanmedian(np.gradient(cdf) / (dw / sig))
        nx = int(2 ** np.ceil(np.log2(2 / x_per_sigma)))
        lam = np.interp(np.linspace(0, 1, nx), cdf, w)
        conv = fft_convolve(np.interp(lam, w, s), taper_gauss(nx, 1.0 / nx, x_per_sigma))
        return np.interp(outwave, lam, conv)
    with np.errstate(invalid="ignore"):
        sig = np.sqrt(sigma ** 2 - inres_v ** 2)
    wr, sr = resample_pow2(w, s)
    taper = taper_vsini(len(sr), _grid_dv(wr), sig) if smoothtype == 'vsini' else taper_gauss(len(sr), _grid_dv(wr), sig)
    return np.interp(outwave, wr, fft_convolve(sr, taper), left=np.nan, right=np.nan)


def smooth_lsf(wave, spec, disparr, outwave, pix_per_sigma=2):
    """smoothspec(..., smoothtype='lsf', fftsmooth=True) -> smooth_lsf_fft
    (Payne/utils/smoothing.py:125-128, 131-151, 482-586): wavelength-dependent Gaussian LSF of dispersion
    ``disparr`` (same units as ``wave``, one value per input pixel) by warping to the coordinate in which the
    LSF has constant width.  Returns the smoothed spectrum on ``outwave`` (np.interp: clamped, no NaN)."""
    wlim = np.array([outwave.min(), outwave.max()]) + 20.0 * 100 * np.array([-1, 1])      # mask_wave, linear, width=100
    mask = (wave > wlim[0]) & (wave < wlim[1])
    w, s, sigma = wave[mask], np.nan_to_num(spec[mask], nan=1.0), disparr[mask]
    dw = np.gradient(w)
    cdf = np.cumsum(dw / sigma)
    cdf /= cdf.max()
    sigma_per_pixel = dw / sigma
    x_per_pixel = np.gradient(cdf)
    x_per_sigma = np.nanmedian(x_per_pixel / sigma_per_pixel)
    N = pix_per_sigma / x_per_sigma
    nx = int(2 ** np.ceil(np.log2(N)))
    x = np.linspace(0, 1, nx)
    dx = 1.0 / nx
    lam = np.interp(x, cdf, w)
    newspec = np.interp(lam, w, s)
    conv = fft_convolve(newspec, taper_gauss(nx, dx, x_per_sigma))          # smooth_fft(dx, newspec, x_per_sigma)
    return np.interp(outwave, lam, conv)


def getspec(net, Teff=5770.0, logg=4.44, feh=0.0, afe=0.0, rad_vel=None, rot_vel=None,
            vmic=None, inst_R=None, outwave=None, return_stages=False, cnet=None):
    """``PayneSpecPredict.getspec`` (Payne/predict/ystpred.py:119-277 ==
    Payne/predict/predictspec.py:136-294), scalar ``inst_R`` branch.

    ``cnet``: the optional continuum network (``Cnnpath``).
    ``inst_R`` is the sigma-based R handed straight to smoothspec (the 2.355
    factor belongs to genspec).  ``None`` for rad_vel/rot_vel/inst_R means
    "kwarg absent".  Stages: raw ANN, after vsini, on the output grid.
    """
    labels = [Teff, logg, feh, afe]
    if vmic is not None and np.isfinite(vmic):       # ystpred.py:172-187
        labels.append(vmic)
    modspec = ann_forward(net, labels)
    raw = modspec
    modwave = net["wavelength"]
    if cnet is not None:                             # continuum network, ystpred.py:191-209
        modcont = ann_forward(cnet, labels)
        modcontwave = cnet["wavelength"]
        modcont = modcont * (C_KMS_DOPPLER / ((modcontwave * 1E-8) ** 2.0))      # F_nu -> F_lambda
        modcont = modcont / np.nanmedian(modcont)
        modspec = modspec * np.interp(modwave, modcontwave, modcont, right=np.nan, left=np.nan)

    if rot_vel is not None and rot_vel != 0.0:       # ystpred.py:211-224
        modspec = smooth_vsini(modwave, modspec, rot_vel)
        modspec[0] = modspec[1]
        modspec[-1] = modspec[-2]
    after_rot = modspec

    if rad_vel is not None and rad_vel != 0.0:       # ystpred.py:226-232
        modwave = modwave * (1.0 + (rad_vel / C_KMS_DOPPLER))

    smoothed = False
    if inst_R is not None:                           # ystpred.py:233-246
        if outwave is not None:
            outwave = np.array(outwave)
        if isinstance(inst_R, float):
            if inst_R > 0.0:
                smoothed = True
                modspec = smooth_R(modwave, modspec, inst_R, outwave, net["resolution"])
        else:                                        # LSF: dispersion (AA) per output pixel, ystpred.py:248-269
            smoothed = True
            inst_R = np.asarray(inst_R, dtype=np.float64)
            disparr = np.interp(modwave, outwave, inst_R) if outwave is not None else inst_R
            assert len(disparr) == len(modwave), "Length of LSF vector not equal to input wavelength"
            modspec = smooth_lsf(modwave, modspec, disparr, outwave if outwave is not None else modwave)
    if (not smoothed) and (outwave is not None):     # ystpred.py:271-272
        modspec = np.interp(outwave, modwave, modspec, right=np.nan, left=np.nan)
    if outwave is not None:
        modwave = outwave
    if return_stages:
        return modwave, modspec, dict(raw=raw, after_rot=after_rot)
    return modwave, modspec


def polycalc(coef, inwave):
    """Payne/fitting/fitutils.py:11-20."""
    x = inwave - inwave.min()
    x = 2.0 * (x / x.max()) - 1.0
    return np.polynomial.chebyshev.chebval(x, coef)


def genspec(net, pars, outwave=None, modpoly=False, cnet=None):
    """``GenMod.genspec`` (Payne/fitting/genmod.py:58-108), carbon off; ``cnet``: the continuum
    network GenMod._initspecnn(Cnnpath=...) hands to PayneSpecPredict (genmod.py:28-32)."""
    Teff, logg, FeH, aFe, radvel, rotvel, vmic, inst_R = pars[:8]
    polycoef = pars[8:] if modpoly else pars[8:-1]
    if isinstance(inst_R, float):                    # genmod.py:82-85
        inst_R = 2.355 * inst_R
    wave, flux = getspec(net, Teff=Teff, logg=logg, feh=FeH, afe=aFe, rad_vel=radvel,
                         rot_vel=rotvel, vmic=vmic, inst_R=inst_R, outwave=outwave, cnet=cnet)
    if modpoly:                                      # genmod.py:103-106
        flux = flux * polycalc(polycoef, wave)
    return wave, flux


def chi2_spec(model, obs, err):
    """Payne/fitting/likelihood.py:95-97 (python loop there; same sum)."""
    return np.sum(((model - obs) ** 2.0) / (err ** 2.0))


def chi2_spec_loop(model, obs, err):
    """Payne/fitting/likelihood.py:95-97 literally: a Python list built pixel by pixel, then np.sum.
    Same value as chi2_spec to rounding; this is the form bench.py's cpu_baseline times."""
    return np.sum([((m - o) ** 2.0) / (s ** 2.0) for m, o, s in zip(model, obs, err)])


# --------------------------------------------------------------------------
# photometry
# --------------------------------------------------------------------------
def fastann_forward(phot, x):
    """``fastANN.eval`` (Payne/predict/photANN.py:118-131): stacked per-filter
    6->H->H->1 sigmoid nets; input scaling WITHOUT the -0.5 offset."""
    xp = ((np.atleast_2d(x) - phot["xmin"]) / (phot["xmax"] - phot["xmin"])).T
    sig = lambda a: 1.0 / (1 + np.exp(-a))
    a1 = sig(np.matmul(phot["w1"], xp) + phot["b1"])
    a2 = sig(np.matmul(phot["w2"], a1) + phot["b2"])
    return np.squeeze(np.matmul(phot["w3"], a2) + phot["b3"])


def highav_offset(coef, av, rv):
    """``highAv.getAvaprox`` (Payne/predict/highred.py:19-21); ``coef`` is
    [F,5] = a1,b1,a2,b2,c2 (NaN rows for filters missing from the table)."""
    a1, b1, a2, b2, c2 = np.asarray(coef, dtype=np.float64).T
    return a1 + b1 * av * (a2 + b2 * rv + c2 * rv ** 2.0)


def sed_mags(phot, logt, logg, feh, afe, av=0.0, rv=3.1, logl=None, dist=None, logA=None):
    """``FastPayneSEDPredict.sed`` (Payne/predict/predictsed.py:75-103)."""
    if av < 5.0:
        BC = fastann_forward(phot, [10.0 ** logt, logg, feh, afe, av, rv])
    else:
        BC0 = fastann_forward(phot, [10.0 ** logt, logg, feh, afe, 0.0, 3.1])
        BC = np.atleast_1d(BC0) - highav_offset(phot["hiav"], av, rv)
    if (logl is not None) and (dist is not None):
        return -2.5 * logl + 4.74 - BC + (5.0 * np.log10(dist) - 5.0)
    if logA is not None:
        return 5.0 * logA - 10.0 * (logt - np.log10(5770.0)) - 0.26 - BC
    raise IOError("cannot understand input pars into sed function")


def genphot(phot, pars):
    """``GenMod.genphot`` (Payne/fitting/genmod.py:110-155); Rv fixed 3.1."""
    Teff, logg, FeH, aFe, logR, Dist, Av = pars[:7]
    logTeff = np.log10(Teff)
    logL = 2.0 * logR + 4.0 * (logTeff - np.log10(5770.0))
    return sed_mags(phot, logTeff, logg, FeH, aFe, av=Av, rv=3.1, logl=logL, dist=Dist)


def genphot_scaled(phot, pars):
    """``GenMod.genphot_scaled`` (Payne/fitting/genmod.py:157-187)."""
    Teff, logg, FeH, aFe, logA, Av = pars[:6]
    return sed_mags(phot, np.log10(Teff), logg, FeH, aFe, av=Av, rv=3.1, logA=logA)


# --------------------------------------------------------------------------
# likelihood object + lnprobfn
# --------------------------------------------------------------------------
_SPEC_NAMES = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Vmic', 'Inst_R']


class OracleLikelihood(object):
    """``likelihood`` (Payne/fitting/likelihood.py:5-117) over plain arrays.

    fitpars_i  : ordered names of the sampled parameters (likelihood.py:35-40)
    obs_phot   : ordered {filter: (mag, err)} or None
    """

    def __init__(self, net, obs_wave, obs_flux, obs_eflux, fitpars_i, fixedpars=None,
                 modpoly=False, phot=None, obs_phot=None, photscale=False, spec=True, pixel_loop=False):
        self.net = net
        self.pixel_loop = pixel_loop          # chi^2 by the reference's per-pixel Python loop (timing baseline)
        self.obs_wave = None if obs_wave is None else np.asarray(obs_wave, dtype=np.float64)
        self.obs_flux = obs_flux
        self.obs_eflux = obs_eflux
        self.fitpars_i = list(fitpars_i)
        self.fixedpars = dict(fixedpars or {})
        self.modpoly = modpoly
        self.phot = phot
        self.obs_phot = obs_phot
        self.photscale = photscale
        self.spec_bool = spec
        self.phot_bool = phot is not None
        self.ndim = len(self.fitpars_i)
        self.parsdict = {}

    def split(self, pars):
        """likelihood.py:42-75: sampled vector -> (specpars, photpars)."""
        self.parsdict = {pp: vv for pp, vv in zip(self.fitpars_i, pars)}
        self.parsdict.update(self.fixedpars)
        specpars = photpars = None
        if self.spec_bool:
            specpars = [self.parsdict[pp] if pp in self.parsdict else np.nan for pp in _SPEC_NAMES]
            if self.modpoly:
                specpars = specpars + [self.parsdict[pp] for pp in self.fitpars_i if 'pc' in pp]
        if self.phot_bool:
            photpars = [self.parsdict[pp] for pp in ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]']]
            if 'log(A)' in self.fitpars_i:
                photpars = photpars + [self.parsdict['log(A)']]
            else:
                photpars = photpars + [self.parsdict['log(R)'], self.parsdict['Dist']]
            photpars = photpars + [self.parsdict['Av']]
            photpars = photpars + [self.parsdict['Rv'] if 'Rv' in self.fitpars_i else None]
        return specpars, photpars

    def lnlike(self, specpars=None, photpars=None):
        """likelihood.py:84-117."""
        specchi2 = sedchi2 = 0.0
        if self.spec_bool:
            _, modflux = genspec(self.net, specpars, outwave=self.obs_wave, modpoly=self.modpoly)
            specchi2 = (chi2_spec_loop if self.pixel_loop else chi2_spec)(modflux, self.obs_flux, self.obs_eflux)
        if self.phot_bool:
            mags = genphot_scaled(self.phot, photpars) if self.photscale else genphot(self.phot, photpars)
            mags = np.atleast_1d(mags)
            sedchi2 = np.sum([((mags[i] - mo[0]) ** 2.0) / (mo[1] ** 2.0)
                              for i, mo in enumerate(self.obs_phot.values())])
        return -0.5 * (specchi2 + sedchi2)

    def lnlikefn(self, pars):
        pars = [float(p) for p in pars]               # np.float64 is a float; keep the isinstance path
        specpars, photpars = self.split(pars)
        return self.lnlike(specpars=specpars, photpars=photpars)


def lnprobfn(pars, likeobj, lnpriorfn=None):
    """Payne/fitting/fitstar.py:647-659."""
    lnlike = likeobj.lnlikefn(pars)
    if lnlike == -np.inf:
        return -np.inf
    lnprior = 0.0 if lnpriorfn is None else lnpriorfn(likeobj.parsdict)
    if lnprior == -np.inf:
        return -np.inf
    return lnprior + lnlike
