"""The likelihood object: mirrors Payne/fitting/likelihood.py.

``lnlikefn(pars)`` keeps the reference's scalar contract (one parameter vector ->
float, side effect ``self.parsdict``) by running a batch of one; ``lnlike_batch`` is
the new entry the batched sampler drives: theta[B, ndim] -> lnL[B] in one set of
kernel launches."""
import numpy as np

from .genmod import GenMod
from ..engine import THETA_SPEC_COLS


class likelihood(object):
    def __init__(self, fitargs, fitpars, runbools, **kwargs):
        self.verbose = kwargs.get('verbose', True)
        self.fitargs = fitargs
        self.spec_bool, self.phot_bool, self.modpoly_bool, self.photscale_bool, self.carbon_bool = runbools[:5]
        self.fixedpars = self.fitargs['fixedpars']
        self.GM = GenMod(device=kwargs.get('device', None), b_max=kwargs.get('b_max', 512),
                         variant=kwargs.get('variant', 0))
        if self.spec_bool:
            self.GM._initspecnn(nnpath=fitargs['specANNpath'], NNtype=self.fitargs['NNtype'],
                                carbon_bool=self.carbon_bool)
        if self.phot_bool:
            self.GM._initphotnn(self.fitargs['obs_phot'].keys(), nnpath=fitargs['photANNpath'])
        self.fitpars_i = [pp for pp in fitpars[0] if fitpars[1][pp]]          # likelihood.py:35-40
        self.ndim = len(self.fitpars_i)
        self.pc_names = [pp for pp in self.fitpars_i if 'pc' in pp]
        npoly = len(self.pc_names) if self.modpoly_bool else 0
        obs = None
        if self.spec_bool:
            obs = (np.ascontiguousarray(fitargs['obs_wave_fit'], dtype=np.float64),
                   np.ascontiguousarray(fitargs['obs_flux_fit'], dtype=np.float64),
                   np.ascontiguousarray(fitargs['obs_eflux_fit'], dtype=np.float64))
        obs_phot = None
        if self.phot_bool:
            obs_phot = {k: (v[0], v[1]) for k, v in self.fitargs['obs_phot'].items()}
        self.GM.configure(obs=obs, obs_phot=obs_phot, npoly=npoly, photscale=self.photscale_bool)
        self.parsdict = {}
        self._colmap = None

    # -- sampled vector -> ABI theta rows ----------------------------------------
    def _columns(self):
        """(theta column, source) pairs: source is an index into the sampled vector or a
        fixed value.  Column order of include/payne_hip.h == specpars + photpars of
        likelihood.py:50-72."""
        if self._colmap is not None:
            return self._colmap
        eng = self.GM.engine
        off = eng.phot_off
        col_of = {n: i for i, n in enumerate(THETA_SPEC_COLS)}
        for i, n in enumerate(self.pc_names):
            col_of[n] = 8 + i
        col_of.update({'log(A)': off, 'log(R)': off, 'Dist': off + 1, 'Av': off + 2, 'Rv': off + 3})
        idx, fixed = [], []
        for j, name in enumerate(self.fitpars_i):
            if name in col_of:
                idx.append((col_of[name], j))
        for name, val in self.fixedpars.items():
            if name == 'Inst_R' and np.ndim(val) > 0:
                # a fixed LSF vector (dispersion per observed pixel): genspec hands it to getspec as is
                # (genmod.py:82-85 -> ystpred.py:248-269); it lives in the engine, not in theta
                eng.set_lsf(np.asarray(val, dtype=np.float64))
            elif name in col_of:
                fixed.append((col_of[name], float(val)))
        self._colmap = (eng.ncols, idx, fixed)
        return self._colmap

    def theta_rows(self, theta):
        theta = np.atleast_2d(np.asarray(theta, dtype=np.float64))
        ncols, idx, fixed = self._columns()
        full = np.full((theta.shape[0], ncols), np.nan)
        for c, j in idx:
            full[:, c] = theta[:, j]
        for c, v in fixed:
            full[:, c] = v
        return full

    # -- batch entry ---------------------------------------------------------------
    def lnlike_batch(self, theta, as_numpy=True):
        """theta[B, ndim] (sampled-vector order) -> lnL[B]."""
        out = self.GM.engine.lnlike_batch(self.theta_rows(theta))
        return out.cpu().numpy() if as_numpy else out

    # -- reference API -------------------------------------------------------------
    def lnlikefn(self, pars):
        """likelihood.py:42-82."""
        self.parsdict = {pp: vv for pp, vv in zip(self.fitpars_i, pars)}
        for kk in self.fixedpars.keys():
            self.parsdict[kk] = self.fixedpars[kk]
        return float(self.lnlike_batch(np.asarray(pars, dtype=np.float64)[None, :])[0])

    def lnlike(self, specpars=None, photpars=None):
        """likelihood.py:84-117 with explicit parameter lists."""
        eng = self.GM.engine
        th = np.full((1, eng.ncols), np.nan)
        if self.spec_bool:
            th[0, :8] = specpars[:8]
            if self.modpoly_bool:
                th[0, 8:8 + eng.npoly] = specpars[8:8 + eng.npoly]
        if self.phot_bool:
            th[0, 0:4] = photpars[0:4]
            off = eng.phot_off
            if self.photscale_bool:
                th[0, off], th[0, off + 2] = photpars[4], photpars[5]
            else:
                th[0, off], th[0, off + 1], th[0, off + 2] = photpars[4], photpars[5], photpars[6]
        return float(eng.lnlike_batch(th).cpu().numpy()[0])
