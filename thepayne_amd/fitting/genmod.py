"""Model generation: mirrors Payne/fitting/genmod.py (GenMod).

One ``PayneEngine`` carries the spectral emulator, the observed grid and the photometric
emulator of a fit; ``genspec`` / ``genphot`` / ``genphot_scaled`` evaluate one parameter
list through it (API parity, B = 1) and the ``*_batch`` methods are what the batched
likelihood uses."""
import numpy as np

from .. import nnio
from ..engine import PayneEngine
from .fitutils import polycalc

speedoflight = 299792.458        # km/s, the Doppler constant of getspec (ystpred.py:232)

# ---- contexts and networks kept between fits ----------------------------------------------------------------------------
# A complete C2 fit is ~35 ms of sampling; building its context (weights uploaded, the output layer restated in the frequency
# domain on the host, tables) was 19 ms more and reading the network file 4 ms, for every star of a multi-star job
# (tools/fit_stars.py).  The reference reads its network once per process too (genmod.py:15-32 at likelihood set-up) -- here the
# CONTEXT of a finished fit is kept as well: when the last holder of a fit's engine lets go (the GenMod that built it, a
# DeviceProposer walking on it), the open context goes into a small pool keyed by everything it was built from (network file
# identity, kind, device, batch size, blaze order, kernel variant); the next fit with the same key takes it and only binds its
# own observed spectrum (payne_ctx_set_obs, which also drops an LSF vector).  Fits with photometry or a continuum network, and
# networks handed over as dictionaries, build their own context as before.
_ENGINE_POOL = {}
_ENGINE_POOL_MAX = 2
_NET_CACHE = {}
_NET_CACHE_MAX = 2


def _file_key(path):
    import os
    try:
        st = os.stat(path)
    except (OSError, TypeError, ValueError):
        return None
    return (os.path.abspath(path), st.st_mtime_ns, st.st_size)


def _pool_put(key):
    def put(eng):
        idle = _ENGINE_POOL.setdefault(key, [])
        if len(idle) >= _ENGINE_POOL_MAX:
            return False
        idle.append(eng)
        return True
    return put


def drop_idle_engines():
    """Close every context kept for re-use (tests; before a process forks or exits)."""
    for idle in list(_ENGINE_POOL.values()):
        while idle:
            try:
                idle.pop().close()
            except Exception:
                pass
    _ENGINE_POOL.clear()


import atexit  # noqa: E402
atexit.register(drop_idle_engines)


class GenMod(object):
    def __init__(self, *arg, **kwargs):
        self.verbose = kwargs.get('verbose', False)
        self.device = kwargs.get('device', None)
        self.b_max = kwargs.get('b_max', 512)
        self.variant = int(kwargs.get('variant', 0))      # payne_opts.variant (PAYNE_V_* bits; 0 = default kernels)
        self._spec_net = None
        self._phot = None
        self._obs = None
        self._obs_phot = None
        self._npoly = 0
        self._photscale = False
        self._engine = None
        self.filterarray = None

    # -- configuration (engine is (re)built lazily from these) -------------------
    def _initspecnn(self, nnpath=None, **kwargs):
        """genmod.py:15-32: NNtype 'YST1' -> ystpred layout, anything else -> predictspec."""
        self.NNtype = kwargs.get('NNtype', 'YST1')
        self._spec_key = _file_key(nnpath) if isinstance(nnpath, str) else None
        ck = (self._spec_key, self.NNtype)
        if self._spec_key is not None and ck in _NET_CACHE:
            self._spec_net = _NET_CACHE[ck]
        else:
            self._spec_net = nnio.load_spec_net(nnpath, self.NNtype)
            if self._spec_key is not None:
                while len(_NET_CACHE) >= _NET_CACHE_MAX:
                    _NET_CACHE.pop(next(iter(_NET_CACHE)))
                _NET_CACHE[ck] = self._spec_net
        Cnnpath = kwargs.get('Cnnpath', None)              # genmod.py:28-32 (the reference's likelihood never passes it)
        self._cont_net = nnio.load_spec_net(Cnnpath, self.NNtype, rescale_teff=False) if Cnnpath is not None else None
        self._let_go()

    def _initphotnn(self, filterarray, nnpath=None):
        """genmod.py:35-43."""
        from ..predict.predictsed import _ALLFILTERS
        self.filterarray = list(filterarray) if filterarray is not None else list(_ALLFILTERS)
        self._phot = nnio.load_phot_nets(self.filterarray, nnpath)
        self._let_go()

    def configure(self, obs=None, obs_phot=None, npoly=0, photscale=False):
        """Bind the data side of the fit (observed spectrum / magnitudes, blaze order)."""
        self._obs, self._obs_phot, self._npoly, self._photscale = obs, obs_phot, int(npoly), bool(photscale)
        self._let_go()
        self._pred = None

    def new_engine(self):
        """Another context with the same networks and data (own workspaces: a second batch can be in flight)."""
        eng = PayneEngine(self._spec_net, obs=self._obs, phot=self._phot, obs_phot=self._obs_phot,
                          npoly=self._npoly, photscale=self._photscale, b_max=self.b_max, device=self.device,
                          variant=self.variant)
        if getattr(self, "_cont_net", None) is not None:
            eng.set_continuum(self._cont_net)
        return eng

    def _pool_key(self):
        """What a context of this fit was built from, or None when it is not one the pool keeps (see the module's head)."""
        if getattr(self, "_spec_key", None) is None or self._phot is not None or getattr(self, "_cont_net", None) is not None:
            return None
        if self._obs is None or len(self._obs) != 3:
            return None
        return (self._spec_key, self.NNtype, self.device, int(self.b_max), int(self._npoly), bool(self._photscale), int(self.variant))

    @property
    def engine(self):
        if self._engine is None or not self._engine.is_open():
            if self._engine is not None:
                self._let_go()
            key, eng = self._pool_key(), None
            idle = _ENGINE_POOL.get(key) if key is not None else None
            while idle and eng is None:
                cand = idle.pop()
                try:
                    if cand.is_open():
                        cand.set_obs(*self._obs)
                        eng = cand
                except Exception:                           # a context that cannot take this spectrum: build a fresh one
                    cand.close()
            if eng is None:
                eng = self.new_engine()
            if key is not None:
                eng._on_idle = _pool_put(key)
            eng.hold()
            self._engine = eng
        return self._engine

    def _let_go(self):
        eng, self._engine = getattr(self, "_engine", None), None
        if eng is not None:
            try:
                eng.drop()
            except Exception:
                pass

    def __del__(self):
        self._let_go()

    # -- reference API (one parameter list) --------------------------------------
    def genspec(self, pars, outwave=None, verbose=False, modpoly=False, carbon_bool=False):
        """(wave, flux) for pars = [Teff, logg, FeH, aFe, Vrad, Vrot, Vmic, Inst_R, pc...]
        (genmod.py:58-108).  Inst_R is FWHM-based; the 2.355 factor is applied on the GPU."""
        pars = list(pars)
        eng = self.engine
        lsf = None
        if not isinstance(pars[7], float):                 # LSF vector: handed to getspec as is (genmod.py:82-85)
            lsf = np.atleast_1d(np.asarray(pars[7], dtype=np.float64))
            pars[7] = np.nan
        polycoef = pars[8:] if modpoly else []
        if modpoly and len(polycoef) != eng.npoly:
            raise ValueError("engine configured for %d blaze coefficients, got %d" % (eng.npoly, len(polycoef)))

        def row(e):
            th = np.full((1, e.ncols), np.nan)
            th[0, :8] = pars[:8]
            if modpoly:
                th[0, 8:8 + e.npoly] = polycoef
            return th
        # the grid the spectrum comes out on: outwave, or the Doppler-shifted model grid (getspec, ystpred.py:232-272)
        if outwave is not None:
            grid = np.ascontiguousarray(outwave, dtype=np.float64)
        else:
            rv = pars[4]
            grid = eng.wavelength * (1.0 + rv / speedoflight) if rv != 0.0 else eng.wavelength
            grid = np.ascontiguousarray(grid, dtype=np.float64)
            smoothed = lsf is not None or (isinstance(pars[7], float) and pars[7] > 0.0)
            if not smoothed:                               # no instrumental stage: the broadened ANN spectrum as is
                flux = eng.predict_batch(row(eng), stage=1, fwhm_R=True).cpu().numpy()[0].astype(np.float64)
                if modpoly:
                    flux = flux * polycalc(polycoef, grid)
                return grid, flux
        use = eng
        if eng.nobs != len(grid) or not np.array_equal(getattr(eng, "obs_wave", None), grid):
            use = self._prediction_engine(grid)            # another grid than the fit's: its own context
        if lsf is not None:
            use.set_lsf(lsf)
        try:
            flux = use.predict_batch(row(use), stage=3 if modpoly else 2, fwhm_R=True).cpu().numpy()[0].astype(np.float64)
        finally:
            if lsf is not None:
                use.set_lsf(None)
        if outwave is None and lsf is None:
            from ..predict._spec import native_grid_edges
            flux = native_grid_edges(grid, flux)
        return grid, flux

    def _prediction_engine(self, grid):
        """A context of its own for genspec on grids other than the fit's observed grid (the fit's
        context keeps its observed flux bound); re-bound when the grid changes."""
        key = (len(grid), float(grid[0]), float(grid[-1]), hash(grid.tobytes()))
        if getattr(self, "_pred", None) is None:
            self._pred = PayneEngine(self._spec_net, obs=(grid,), npoly=self._npoly, b_max=1, device=self.device)
            if getattr(self, "_cont_net", None) is not None:
                self._pred.set_continuum(self._cont_net)
            self._pred_key = key
        elif self._pred_key != key:
            self._pred.set_obs(grid)
            self._pred_key = key
        return self._pred

    def _phot_theta(self, pars, scaled):
        eng = self.engine
        th = np.full((1, eng.ncols), np.nan)
        th[0, 0:4] = pars[0:4]
        off = eng.phot_off
        if scaled:
            th[0, off], th[0, off + 2] = pars[4], pars[5]
        else:
            th[0, off], th[0, off + 1], th[0, off + 2] = pars[4], pars[5], pars[6]
        return th

    def _sed_rows(self, theta_full, scaled):
        """theta rows -> the 9 sed() kwargs per row (genmod.py:110-187)."""
        eng = self.engine
        off = eng.phot_off
        th = np.asarray(theta_full, dtype=np.float64)
        logt = np.log10(th[:, 0])
        rows = np.full((th.shape[0], 9), np.nan)
        rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3] = logt, th[:, 1], th[:, 2], th[:, 3]
        rows[:, 4], rows[:, 5] = th[:, off + 2], 3.1
        if scaled:
            rows[:, 8] = th[:, off]
        else:
            rows[:, 6] = 2.0 * th[:, off] + 4.0 * (logt - np.log10(5770.0))
            rows[:, 7] = th[:, off + 1]
        return rows

    def genphot(self, pars, rvfree=False, verbose=False):
        """{filter: mag} from [Teff, logg, FeH, aFe, logR, Dist, Av(, Rv)] (genmod.py:110-155)."""
        rows = self._sed_rows(self._phot_theta(pars, False), False)
        if rvfree:
            rows[:, 5] = pars[7]
        sed = self.engine.sed_batch(rows).cpu().numpy()[0]
        return {ff: m for m, ff in zip(sed, self.filterarray)}

    def genphot_scaled(self, pars, rvfree=False, verbose=False):
        """{filter: mag} from [Teff, logg, FeH, aFe, logA, Av(, Rv)] (genmod.py:157-187)."""
        rows = self._sed_rows(self._phot_theta(pars, True), True)
        if rvfree:
            rows[:, 5] = pars[6]
        sed = self.engine.sed_batch(rows).cpu().numpy()[0]
        return {ff: m for m, ff in zip(sed, self.filterarray)}
