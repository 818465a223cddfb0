"""Shared implementation of the two spectral predictors.

``Payne.predict.ystpred.PayneSpecPredict`` (numpy YST1 net) and
``Payne.predict.predictspec.PayneSpecPredict`` (torch LinNet/SMLP) have the same
``getspec`` body (Payne/predict/ystpred.py:119-277 == predictspec.py:136-294); here
both are one class whose arithmetic runs in the HIP engine.  New: ``getspec_batch``.
"""
import numpy as np

from .. import nnio
from ..engine import PayneEngine

speedoflight = 299792.458            # scipy.constants.c / 1000 (ystpred.py:11-12)


def native_grid_edges(grid, flux):
    """smoothspec(outwave=None) interpolates the convolved spectrum from the resampled grid
    exp(linspace(ln wmin, ln wmax, n)) back onto the input grid itself with left = right = NaN
    (smoothing.py:140-141, 283-288): an end pixel is NaN whenever exp(log(w)) rounds to the inside of w.
    The kernel works in ln(lambda) and cannot see that rounding; the same numpy expressions decide here."""
    lo, hi = np.exp(np.log(grid.min())), np.exp(np.log(grid.max()))
    flux[(grid < lo) | (grid > hi)] = np.nan
    return flux


class SpecANN(object):
    """The emulator object the reference exposes as ``.anns``: ``Net``
    (ystpred.py:18-58) / ``ANN`` (predictspec.py:29-74).  ``eval(labels)`` returns
    the raw ANN spectrum on ``wavelength``."""

    def __init__(self, nnpath, NNtype, b_max=256, device=None, variant=0):
        self.nnpath = nnpath
        self.NNtype = NNtype
        self.net = nnio.load_spec_net(nnpath, NNtype)
        self.xmin = self.net["xmin"]
        self.xmax = self.net["xmax"]
        self.wavelength = self.net["wavelength"]
        self.resolution = self.net["resolution"]
        self.engine = PayneEngine(self.net, b_max=b_max, device=device, variant=variant)
        self.n_labels = self.engine.n_labels

    def _theta(self, labels):
        labels = np.atleast_2d(np.asarray(labels, dtype=np.float64))
        if labels.shape[1] != self.n_labels:
            raise ValueError("this ANN takes %d labels, got %d" % (self.n_labels, labels.shape[1]))
        th = np.full((labels.shape[0], self.engine.ncols), np.nan)
        th[:, 0:4] = labels[:, 0:4]
        if self.n_labels == 5:
            th[:, 6] = labels[:, 4]
        th[:, 4:6] = 0.0
        return th

    def eval(self, x):
        x = np.asarray(x, dtype=np.float64)
        out = self.engine.predict_batch(self._theta(x), stage=0).cpu().numpy()
        out = out.astype(np.float64) if self.NNtype in ("YST1", "YST2") else out
        return out[0] if x.ndim == 1 else out


class ContANN(object):
    """``.Canns``: the continuum network (``Net(Cnnpath)``, ystpred.py:81-85).  It lives in the spectral
    network's context: from here on every spectrum that context produces past the raw ANN stage is multiplied
    by the normalised continuum (ystpred.py:191-209)."""

    def __init__(self, nnpath, NNtype, spec_ann):
        self.nnpath = nnpath
        self.net = nnio.load_spec_net(nnpath, NNtype, rescale_teff=False)   # Canns is used as stored (ystpred.py:81-85)
        self.xmin, self.xmax = self.net["xmin"], self.net["xmax"]
        self.wavelength = self.net["wavelength"]
        self.resolution = self.net["resolution"]
        self._spec = spec_ann
        spec_ann.engine.set_continuum(self.net)

    def eval(self, x):
        x = np.asarray(x, dtype=np.float64)
        out = self._spec.engine.predict_batch(self._spec._theta(x), stage=4).cpu().numpy().astype(np.float64)
        return out[0] if x.ndim == 1 else out


class PayneSpecPredict(object):
    """Predict spectra from a Payne-learned ANN (drop-in for the reference class)."""

    default_NNtype = "YST1"

    def __init__(self, nnpath=None, **kwargs):
        self.NN = {}
        if nnpath is None:
            raise IOError("no default ANN ships with this build (the reference's data/ is empty too, "
                          "README.md:45); pass nnpath=")
        self.nnpath = nnpath
        self.NNtype = kwargs.get('NNtype', self.default_NNtype)
        self.anns = SpecANN(self.nnpath, self.NNtype, b_max=kwargs.get('b_max', 256), device=kwargs.get('device'),
                            variant=kwargs.get('variant', 0))      # (variant: kernel variants of include/payne_hip.h, for tests)
        self.Cnnpath = kwargs.get('Cnnpath', None)
        self.Canns = None
        if self.Cnnpath is not None:                    # ystpred.py:81-85
            self.Canns = ContANN(self.Cnnpath, self.NNtype, self.anns)
        self._bound = None

    # -- reference API -----------------------------------------------------------
    def predictspec(self, labels):
        """Raw ANN flux for [Teff, log(g), [Fe/H], [alpha/Fe] (, vmic)] (ystpred.py:84-99)."""
        return self.anns.eval(labels)

    def predictcont(self, labels):
        """Continuum network output for the same labels (ystpred.py:101-117)."""
        if self.Canns is None:
            raise AttributeError("no continuum network (pass Cnnpath=)")
        return self.Canns.eval(labels)

    @staticmethod
    def _labels_from_kwargs(kwargs):
        """Aliases and defaults of ystpred.py:132-165."""
        if 'Teff' in kwargs:
            teff = kwargs['Teff']
        elif 'logt' in kwargs:
            teff = 10.0 ** kwargs['logt']
        else:
            teff = 5770.0
        logg = kwargs['log(g)'] if 'log(g)' in kwargs else kwargs.get('logg', 4.44)
        feh = kwargs['[Fe/H]'] if '[Fe/H]' in kwargs else kwargs.get('feh', 0.0)
        afe = 0.0
        for k in ('[alpha/Fe]', '[a/Fe]', 'aFe', 'afe'):
            if k in kwargs:
                afe = kwargs[k]
                break
        return teff, logg, feh, afe

    def _bind(self, outwave):
        key = (len(outwave), float(outwave[0]), float(outwave[-1]), hash(outwave.tobytes()))
        if self._bound != key:
            self.anns.engine.set_obs(outwave)
            self._bound = key

    def getspec(self, **kwargs):
        """(wave, flux) for one parameter set; kwargs as the reference's getspec."""
        self.inputdict = {}
        teff, logg, feh, afe = self._labels_from_kwargs(kwargs)
        self.inputdict.update(teff=teff, logg=logg, feh=feh, afe=afe)
        vmic = kwargs.get('vmic', np.nan)
        vmic = vmic if (vmic is not None and np.isfinite(vmic)) else np.nan
        self.inputdict['vmic'] = vmic
        outwave = kwargs.get('outwave', None)
        rot_vel = kwargs.get('rot_vel', 0.0)
        rad_vel = kwargs.get('rad_vel', 0.0)
        has_R = 'inst_R' in kwargs
        inst_R = kwargs.get('inst_R', np.nan)
        eng = self.anns.engine
        if has_R and not isinstance(inst_R, float):
            # LSF case: inst_R is the dispersion (AA) at each output pixel (ystpred.py:248-269); anything that
            # is not a Python float lands here in the reference too (np.float64 is one, np.float32 / int are not)
            lsf = np.atleast_1d(np.asarray(inst_R, dtype=np.float64))
            modwave = self.anns.wavelength
            if rad_vel != 0.0:
                modwave = modwave * (1.0 + (rad_vel / speedoflight))
            grid = np.ascontiguousarray(outwave, dtype=np.float64) if outwave is not None else np.ascontiguousarray(modwave)
            if outwave is None:
                assert len(lsf) == len(modwave), 'Length of LSF vector not equal to input wavelength'
            elif len(lsf) != len(grid):
                raise ValueError("fp and xp are not of the same length.")      # np.interp(modwave, outwave, inst_R)
            self._bind(grid)
            th = np.full((1, eng.ncols), np.nan)
            th[0, :8] = [teff, logg, feh, afe, rad_vel, rot_vel, vmic, np.nan]
            eng.set_lsf(lsf)
            try:
                flux = eng.predict_batch(th, stage=2).cpu().numpy()[0].astype(np.float64)
            finally:
                eng.set_lsf(None)
            return grid, flux
        th = np.full((1, eng.ncols), np.nan)
        th[0, :8] = [teff, logg, feh, afe, rad_vel, rot_vel, vmic, inst_R if has_R else np.nan]
        modwave = self.anns.wavelength
        if rad_vel != 0.0:
            modwave = modwave * (1.0 + (rad_vel / speedoflight))
        if outwave is None:
            if has_R and inst_R > 0.0:
                # smoothspec(outwave=None) -> output on the (shifted) model grid itself
                self._bind(np.ascontiguousarray(modwave))
                return modwave, native_grid_edges(modwave, eng.predict_batch(th, stage=2).cpu().numpy()[0].astype(np.float64))
            return modwave, eng.predict_batch(th, stage=1).cpu().numpy()[0].astype(np.float64)
        outwave = np.ascontiguousarray(outwave, dtype=np.float64)
        self._bind(outwave)
        return outwave, eng.predict_batch(th, stage=2).cpu().numpy()[0].astype(np.float64)

    def smoothspec(self, wave, spec, sigma, outwave=None, **kwargs):
        """``smoothspec(wave, spec, resolution=sigma, outwave=outwave, **kwargs)`` (ystpred.py:279-281 ->
        Payne/utils/smoothing.py:19-169) on the GPU.  The FFT branches the sampler's path uses -- smoothtype 'vsini'
        (outwave=None), 'vel' and 'R' (optional ``inres``), 'lsf' (dispersion vector on ``wave``) -- run through the
        likelihood's own kernels; ``fftsmooth=False`` (smooth_vel / smooth_wave / smooth_lsf) and smoothtype
        'lambda' run through payne_smooth_direct (csrc/k_smooth.hip)."""
        smoothtype = kwargs.get('smoothtype', 'vel')
        if smoothtype not in ('vel', 'vsini', 'R', 'lambda', 'lsf'):
            raise NotImplementedError("smoothtype=%r (the reference knows 'vel', 'vsini', 'R', 'lambda', 'lsf')" % (smoothtype,))
        if smoothtype == 'lambda' or (not kwargs.get('fftsmooth', True) and smoothtype != 'vsini'):
            return self._smoothspec_direct(wave, spec, sigma, outwave, smoothtype, dict(kwargs))
        wave = np.ascontiguousarray(wave, dtype=np.float64)
        spec = np.asarray(spec, dtype=np.float64)
        ckms = 2.998e5                                                       # smoothing.py:16
        inres = kwargs.get('inres', None)
        r_in = np.inf
        # What smoothspec does before it calls the smoothing function is argument preparation and stays here (as in
        # _smoothspec_direct): with outwave=None, min_wave_smooth / max_wave_smooth restrict the INPUT (mask_wave, smoothing.py:131-135,
        # 631-647) and the result comes back on all of `wave` (:140-141), NaN outside the kept range.
        wlo, whi = kwargs.get('min_wave_smooth', 0), kwargs.get('max_wave_smooth', np.inf)
        limited = outwave is None and (wlo != 0 or whi != np.inf)
        ow = None if outwave is None else np.ascontiguousarray(outwave, dtype=np.float64)

        def masked(width, linear, by_out):
            """wave / spec inside mask_wave's limits: from outwave (by_out) or from min / max_wave_smooth."""
            wlim = np.array([ow.min(), ow.max()]) if by_out else np.squeeze(np.array([wlo, whi])).astype(np.float64)
            wlim = wlim + 20.0 * float(width) * np.array([-1, 1]) if linear else wlim * (1 + 20.0 / width * np.array([-1, 1]))
            m = (wave > wlim[0]) & (wave < wlim[1])
            if m.sum() < 8:
                raise ValueError("fewer than 8 input pixels inside the smoothing limits")
            return m
        lsf_wave = None
        if smoothtype == 'vsini':
            sig = float(sigma)
            sig_eff = float(np.sqrt(sig ** 2 - float(inres or 0.0) ** 2))       # smooth_vsini_fft, smoothing.py:296-297 (NaN if negative)
            if ow is None and not limited:
                th_R, th_rot, stage = np.nan, sig_eff, 1                       # onto the model grid itself (what getspec calls)
            else:
                # onto another grid: the result is interpolated from the stage's own resampled grid (:308-311), after mask_wave
                m = masked(ckms / sig, False, ow is not None)
                grid_out = wave if ow is None else ow
                wave, spec = np.ascontiguousarray(wave[m]), spec[m]
                th_R, th_rot, stage = np.nan, sig_eff, 4
        elif smoothtype in ('vel', 'R'):
            if smoothtype == 'vel':                                          # sigma in km/s -> R_sigma; inres in km/s
                th_R = ckms / float(sigma)
                r_in = (ckms / inres) if inres else np.inf
            else:                                                            # smoothing.py:103-115
                th_R = float(sigma)
                r_in = float(inres) if inres is not None else np.inf
            th_rot, stage = 0.0, 2
            grid_out = wave if ow is None else ow
            if limited:
                m = masked(th_R, False, False)
                wave, spec = np.ascontiguousarray(wave[m]), spec[m]
        elif smoothtype == 'lsf':
            th_R, th_rot, stage = np.nan, 0.0, 2
            lsf_vec = np.atleast_1d(np.asarray(sigma, dtype=np.float64))
            grid_out = wave if ow is None else ow
            if limited:
                m = masked(100, True, False)
                wave, spec, lsf_vec = np.ascontiguousarray(wave[m]), spec[m], lsf_vec[m]
            if limited or (ow is not None and not np.array_equal(ow, wave)):
                lsf_wave = wave                                              # the vector is a function of the INPUT wavelengths (smoothing.py:145-147)
        else:
            raise NotImplementedError("smoothtype=%r is not built (have 'vsini', 'vel', 'R', 'lsf')" % (smoothtype,))
        eng = self._smooth_engine(wave, r_in)
        if stage >= 2:
            eng.set_obs(np.ascontiguousarray(grid_out))
        th = np.full((1, eng.ncols), np.nan)
        th[0, :8] = [5000.0, 4.0, 0.0, 0.0, 0.0, th_rot, np.nan, th_R]
        if smoothtype == 'lsf':
            eng.set_lsf(lsf_vec, wave=lsf_wave)
        out = eng.smooth_batch(spec[None, :], th, stage=stage).cpu().numpy()[0].astype(np.float64)
        if smoothtype in ('vel', 'R', 'vsini') and stage >= 2:
            # np.interp(outwave, exp(linspace(ln wmin, ln wmax, n)), conv, left = right = NaN): an output pixel that coincides with an
            # end of the (kept) input grid is NaN whenever exp(log(w)) rounds to the inside of w; the kernel works in ln(lambda) and
            # cannot see that rounding -- the same numpy expressions decide here (everything further out is NaN already)
            lo, hi = np.exp(np.log(wave.min())), np.exp(np.log(wave.max()))
            out[(grid_out < lo) | (grid_out > hi)] = np.nan
        return out

    def _smoothspec_direct(self, wave, spec, resolution, outwave, smoothtype, kw):
        """The branches of smoothspec behind payne_smooth_direct: what the reference does before it calls the smoothing
        function (units, mask, nan_to_num: smoothing.py:89-138) is argument preparation and stays here; the smoothing
        itself runs on the GPU."""
        import ctypes as C
        from .. import _lib
        lib = _lib.load()
        dev = self._device_index()
        ckms = 2.998e5                                                       # smoothing.py:16
        wave = np.ascontiguousarray(wave, dtype=np.float64)
        spec = np.asarray(spec, dtype=np.float64)
        fft = kw.get('fftsmooth', True)
        inres = kw.get('inres', 0)

        def run(kind, w, s, ow, sig, inres=0.0, in_vel=False, nsigma=10.0):
            w, s = np.ascontiguousarray(w, dtype=np.float64), np.ascontiguousarray(s, dtype=np.float64)
            ow = np.ascontiguousarray(ow, dtype=np.float64)
            sig = None if sig is None else np.ascontiguousarray(np.atleast_1d(sig), dtype=np.float64)
            out = np.empty(len(ow))
            if len(w) < 2:
                # a mask that leaves fewer than two input pixels (an output grid outside the input's range): the reference's
                # quadratures divide trapz over an empty or one-point set by itself (0 / 0 = NaN); np.interp raises on an
                # empty set of sample points and returns the one sample otherwise
                if kind == _lib.SMOOTH_INTERP:
                    if len(w) == 0:
                        raise ValueError("array of sample points is empty")
                    return np.full(len(ow), float(s[0]))
                return np.full(len(ow), np.nan)
            rc = lib.payne_smooth_direct(dev, kind, w.ctypes.data, s.ctypes.data, len(w), ow.ctypes.data, len(ow),
                                         None if sig is None else sig.ctypes.data, 0 if sig is None else len(sig),
                                         float(inres), int(bool(in_vel)), float(nsigma), out.ctypes.data)
            if rc == _lib.E_SIGMA:                                            # smoothing.py:381-383
                raise ValueError("Desired wavelength sigma is lower than the value possible for this input spectrum.")
            if rc == _lib.E_INVALID:
                raise ValueError("payne_smooth_direct: invalid arguments (kind %d, %d input pixels, %d sigma values)"
                                 % (kind, len(w), 0 if sig is None else len(sig)))
            if rc != 0:
                raise RuntimeError("payne_smooth_direct failed (%d)" % rc)
            return out

        if smoothtype in ('vel', 'R'):                                       # smoothing.py:89-115
            sig = float(resolution) if smoothtype == 'vel' else ckms / float(resolution)
            width, linear = ckms / sig, False
            if smoothtype == 'R' and 'inres' in kw:
                inres = ckms / kw['inres']
        elif smoothtype == 'lambda':                                         # :117-124
            sig, width, linear = resolution, resolution, True
        else:                                                                # 'lsf', :126-129
            sig, width, linear = resolution, 100, True
        # mask_wave (:631-647)
        if outwave is not None:
            ow = np.asarray(outwave, dtype=np.float64)
            wlim = np.array([ow.min(), ow.max()])
        else:
            wlim = np.squeeze(np.array([kw.get('min_wave_smooth', 0), kw.get('max_wave_smooth', np.inf)])).astype(np.float64)
        if linear:
            wlim = wlim + 20.0 * float(width) * np.array([-1, 1])
        else:
            wlim = wlim * (1 + 20.0 / width * np.array([-1, 1]))
        mask = (wave > wlim[0]) & (wave < wlim[1])
        w, s = wave[mask], np.nan_to_num(spec[mask], nan=1.0)                # :133-138
        ow = wave if outwave is None else np.asarray(outwave, dtype=np.float64)
        nsig = kw.get('nsigma', 10)
        if smoothtype == 'lsf':                                              # :142-153 (fftsmooth False)
            if resolution is None:
                if kw.get('lsf') is None:
                    return run(_lib.SMOOTH_INTERP, w, s, ow, None)
                sig_out = np.asarray(kw['lsf'](ow, **{k: v for k, v in kw.items() if k not in ('lsf', 'smoothtype', 'fftsmooth')}))
            else:
                sig_out = run(_lib.SMOOTH_INTERP, wave, np.asarray(resolution, dtype=np.float64), ow, None)   # np.interp(outwave, wave, resolution)
            return run(_lib.SMOOTH_LSF_DIRECT, w, s, ow, sig_out)
        if smoothtype == 'lambda':
            if fft:
                return run(_lib.SMOOTH_WAVE_FFT, w, s, ow, float(sig), inres=inres or 0.0)
            return run(_lib.SMOOTH_WAVE_DIRECT, w, s, ow, float(sig), inres=inres or 0.0, in_vel=kw.get('in_vel', False),
                       nsigma=nsig)
        return run(_lib.SMOOTH_VEL_DIRECT, w, s, ow, sig, inres=inres or 0.0, nsigma=nsig)

    def _smooth_engine(self, wave, r_in):
        """A context whose model grid is ``wave`` (a two-layer dummy network: only the broadening stages run)."""
        key = (len(wave), float(wave[0]), float(wave[-1]), hash(wave.tobytes()), float(r_in))
        cache = self.__dict__.setdefault('_smooth_cache', {})
        if key not in cache:
            if len(cache) >= 4:
                cache.pop(next(iter(cache))).close()
            n = len(wave)
            net = {"layers": [(np.zeros((4, 1), np.float32), np.zeros(4, np.float32), 0),
                              (np.zeros((n, 4), np.float32), np.ones(n, np.float32), 0)],
                   "xmin": np.array([0.0]), "xmax": np.array([1.0]), "wavelength": wave, "resolution": float(r_in)}
            cache[key] = PayneEngine(net, b_max=8, device=self._device_index())
        return cache[key]

    def _device_index(self):
        return self.anns.engine.device.index

    # -- new: batch API ------------------------------------------------------------
    def getspec_batch(self, theta8, outwave, stage=2):
        """theta8[B, 8] = Teff, logg, FeH, aFe, Vrad, Vrot, Vmic, inst_R (sigma-based R as
        in getspec) -> flux[B, len(outwave)] (fp32 device tensor)."""
        eng = self.anns.engine
        theta8 = np.atleast_2d(theta8)
        th = np.full((theta8.shape[0], eng.ncols), np.nan)
        th[:, :8] = theta8
        if stage >= 2:
            self._bind(np.ascontiguousarray(outwave, dtype=np.float64))
        return eng.predict_batch(th, stage=stage)
