"""Fit orchestration: mirrors Payne/fitting/fitstar.py (FitPayne, lnprobfn).

``FitPayne().run(inputdict=...)`` takes the reference's nested ``inputdict`` unchanged
(schema: Payne/fitting/fitstar.py:31-37, 71-194, 262-271) and returns the sampler
object.  The sampling loop is the reference's ``_runsampler`` (fitstar.py:260-463) with
dynesty replaced by the batched driver in ``thepayne_amd.sampler``: prior transform,
likelihood and ln-prior are evaluated for a whole queue of proposals per call, the
likelihood on the GPU.  The text output keeps the reference's format (one row per dead
point, header ``Iter <pars> log(lk) log(vol) log(wt) h nc log(z) delta(log(z))``).
"""
import sys
from datetime import datetime

import ctypes as C

import numpy as np

from .fitutils import airtovacuum
from ..sampler import NestedSampler

__all__ = ["FitPayne", "lnprobfn", "lnprob_batch"]

_ALL_FITPARS = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Vmic', 'Inst_R',
                'log(R)', 'Dist', 'log(A)', 'Av', 'Rv', 'CarbonScale']


def lnprobfn(pars, likeobj, priorobj):
    """Scalar posterior callable handed to a sampler (fitstar.py:647-659)."""
    lnlike = likeobj.lnlikefn(pars)
    if lnlike == -np.inf:
        return -np.inf
    lnprior = priorobj.lnpriorfn(likeobj.parsdict)
    if lnprior == -np.inf:
        return -np.inf
    return lnprior + lnlike


def lnprob_batch(theta, likeobj, priorobj):
    """theta[B, ndim] -> lnprior + lnlike [B]; -inf where either is -inf."""
    theta = np.atleast_2d(theta)
    lnp = priorobj.lnprior_batch(theta)
    lnl = likeobj.lnlike_batch(theta)
    out = lnp + lnl
    out[(lnl == -np.inf) | (lnp == -np.inf)] = -np.inf
    return out


class FitPayne(object):
    def __init__(self, **kwargs):
        from .likelihood import likelihood
        from .prior import prior
        self.prior = prior
        self.likelihood = likelihood
        self.device = kwargs.get('device', None)

    # ---------------------------------------------------------------- inputdict
    def run(self, *args, **kwargs):
        self.verbose = kwargs.get('verbose', True)
        if 'inputdict' not in kwargs:
            print('NO USER DEFINED INPUT DICT, NOTHING TO FIT!')
            raise IOError
        inputdict = kwargs['inputdict']
        self.priordict = inputdict.get('priordict', {})
        self.output = inputdict.get('output', 'Test.dat')
        self.samplerdict = inputdict.get('sampler', {})
        fa = self.fitargs = {}
        self.spec_bool = self.phot_bool = self.modpoly_bool = self.photscale_bool = self.carbon_bool = False
        self.fitpars = list(_ALL_FITPARS)
        self.fitpars_bool = {pp: False for pp in self.fitpars}

        if 'spec' in inputdict:
            spec = inputdict['spec']
            if self.verbose:
                print('... fitting spectrum')
            self.spec_bool = True
            fa['obs_wave'] = np.asarray(spec['obs_wave'])
            fa['obs_flux'] = np.asarray(spec['obs_flux'])
            fa['obs_eflux'] = np.asarray(spec['obs_eflux'])
            fa['specANNpath'] = inputdict.get('specANNpath', None)
            fa['NNtype'] = inputdict.get('NNtype', 'LinNet')
            keep = slice(None)
            if 'wave_minmax' in spec:
                fa['wave_minmax'] = spec['wave_minmax']
                keep = (fa['obs_wave'] >= spec['wave_minmax'][0]) & (fa['obs_wave'] <= spec['wave_minmax'][1])
            for k in ('wave', 'flux', 'eflux'):
                fa['obs_%s_fit' % k] = fa['obs_' + k][keep]
            if spec.get('convertair', True):
                fa['obs_wave_fit'] = airtovacuum(fa['obs_wave_fit'])
            on = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Inst_R']
            if fa['NNtype'] == 'YST2':
                on.append('Vmic')
            for pp in on:
                self.fitpars_bool[pp] = True
            if spec.get('modpoly', False):
                self.modpoly_bool = True
                if 'blaze_coeff' in self.priordict:
                    self.polycoefarr = self.priordict['blaze_coeff']
                    self.polyorder = len(self.polycoefarr)
                else:
                    self.polyorder = spec['polyorder'] + 1 if 'polyorder' in spec else 3
                    self.polysigma = spec.get('polysigma', 1.0) if 'polyorder' in spec else 1.0
                    self.polycoefarr = [[0.0, self.polysigma] for _ in range(self.polyorder)]
                    self.priordict['blaze_coeff'] = self.polycoefarr
                if self.verbose:
                    print('... Fitting a Blaze function with polyoder: {0}'.format(self.polyorder))
                fa['norm_polyorder'] = self.polyorder
                for ii in range(self.polyorder):
                    self.fitpars.append('pc_{}'.format(ii))
                    self.fitpars_bool['pc_{}'.format(ii)] = True
                span = fa['obs_wave_fit'] - fa['obs_wave_fit'].min()
                fa['obs_wave_fit_norm'] = 2.0 * (span / span.max()) - 1.0

        if 'phot' in inputdict:
            fa['photANNpath'] = inputdict.get('photANNpath', None)
            self.phot_bool = True
            for pp in ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Av']:
                self.fitpars_bool[pp] = True
            fa['obs_phot'] = {kk: inputdict['phot'][kk] for kk in inputdict['phot'].keys()}
            self.photscale_bool = inputdict.get('photscale', False)
            if self.photscale_bool:
                self.fitpars_bool['log(A)'] = True
            else:
                self.fitpars_bool['log(R)'] = True
                self.fitpars_bool['Dist'] = True
            self.Rvfree_bool = inputdict.get('Rvfree', False)
            if self.Rvfree_bool:
                self.fitpars_bool['Rv'] = True

        fa['fixedpars'] = {}
        for kk, spec in self.priordict.items():
            if isinstance(spec, dict) and 'fixed' in spec:
                fa['fixedpars'][kk] = spec['fixed']
                self.fitpars_bool[kk] = False

        return self({'fitargs': fa, 'fitpars': [self.fitpars, self.fitpars_bool], 'sampler': self.samplerdict,
                     'priordict': self.priordict,
                     'runbools': [self.spec_bool, self.phot_bool, self.modpoly_bool, self.photscale_bool,
                                  self.carbon_bool]})

    def __call__(self, indicts):
        return self.run_dynesty(indicts)

    # ---------------------------------------------------------------- sampling
    def run_dynesty(self, indicts):
        fitargs, fitpars = indicts['fitargs'], indicts['fitpars']
        samplerdict, runbools = indicts['sampler'], indicts['runbools']
        self.ndim = sum(1 for pp in fitpars[0] if fitpars[1][pp])
        # limits of the library's fixed-size records (include/payne_hip.h), reported here and not from inside the
        # first likelihood batch; the reference itself has none
        from .._lib import PAYNE_MAX_DIM, PAYNE_MAX_POLY
        npoly = fitargs.get('norm_polyorder', 0) if runbools[2] else 0
        if npoly > PAYNE_MAX_POLY:
            raise ValueError("blaze polynomial with %d coefficients: this build carries at most %d (PAYNE_MAX_POLY); "
                             "lower spec['polyorder'] / shorten priordict['blaze_coeff']" % (npoly, PAYNE_MAX_POLY))
        if self.ndim > PAYNE_MAX_DIM:
            raise ValueError("%d sampled parameters: this build's sampler records carry at most %d (PAYNE_MAX_DIM)"
                             % (self.ndim, PAYNE_MAX_DIM))
        self.priorobj = self.prior(fitargs, indicts['priordict'], fitpars, runbools)
        nlive = samplerdict.get('npoints', 200)
        kind = samplerdict.get('samplertype', 'Static')
        widest = 2 * nlive if kind == 'Dynamic' else nlive        # the dynamic sampler's batches carry 2 x npoints
        self.likeobj = self.likelihood(fitargs, fitpars, runbools, device=self.device,
                                       b_max=max(64, widest, int(samplerdict.get('queue_size', widest))))
        if kind == 'Static' and samplerdict.get('use_dynesty', False):
            return self._rundynesty(samplerdict)
        if kind == 'Static':
            return self._runsampler(samplerdict)
        if kind == 'Dynamic':
            return self._rundysampler(samplerdict)
        print('Did not understand sampler type, return nothing')
        return None

    def _proposer(self, samplerdict, k_max):
        """Prior transform, ln-prior and random-walk proposals on the GPU when the fit's priors can be
        expressed there (everything but derived-quantity priors); None = host path."""
        if not samplerdict.get('device_proposals', True):
            return None
        try:
            from ..sampler.device import DeviceProposer
            return DeviceProposer(self.likeobj, self.priorobj, k_max=k_max)
        except NotImplementedError:
            return None

    def _initoutput(self, parnames):
        self.outff = open(self.output, 'w')
        self.outff.write('Iter ')
        for pp in parnames:
            self.outff.write('{} '.format(pp))
        self.outff.write('log(lk) log(vol) log(wt) h nc log(z) delta(log(z))')
        self.outff.write('\n')

    def _row(self, it, vstar, results):
        """One output row.  The reference writes ``likeobj.parsdict`` of the LAST evaluated
        point (fitstar.py:348); with batched evaluation that is
        meaningless, so the dead point ``vstar`` itself is written."""
        pars = {pp: vv for pp, vv in zip(self.likeobj.fitpars_i, vstar)}
        pars.update(self.fitargs_fixed)
        (loglstar, logvol, logwt, h, nc, logz, delta_logz) = results
        self.outff.write('{0} '.format(it))
        self.outff.write(' '.join([str(pars[q]) for q in self.parnames]))
        self.outff.write(' {0} {1} {2} {3} {4} {5} {6} '.format(loglstar, logvol, logwt, h, nc, logz, delta_logz))
        self.outff.write('\n')

    def _rows(self, it0, rec):
        """The rows of one chunk of dead points, same text as ``_row`` one by one.  Formatted by the library
        (payne_format_rows: Python's str() of every value, from std::to_chars) when every fixed parameter is a
        plain number; in Python otherwise."""
        names, fixed = self.likeobj.fitpars_i, self.fitargs_fixed
        col = {pp: j for j, pp in enumerate(names)}
        m = len(rec["logl"])
        fmt = self._row_formatter()
        if fmt is not None and m:
            lib, is_int, src = fmt
            M = np.empty((m, len(is_int)))
            M[:, 0] = np.arange(it0, it0 + m)
            for j, (kind, val) in enumerate(src):
                M[:, 1 + j] = rec["v"][:, val] if kind == 'v' else val
            k = 1 + len(src)
            for j, key in enumerate(("logl", "logvol", "logwt", "h", "nc", "logz", "delta_logz")):
                M[:, k + j] = rec[key]
            if len(self._row_buf) < 48 * M.size + 16:
                self._row_buf = C.create_string_buffer(48 * M.size + 16)
            n = lib.payne_format_rows(M.ctypes.data, m, M.shape[1], is_int.ctypes.data, self._row_buf, len(self._row_buf))
            if n >= 0:
                self.outff.write(self._row_buf.raw[:n].decode('ascii'))
                return
        V = rec["v"].tolist()
        tail = zip(rec["logl"].tolist(), rec["logvol"].tolist(), rec["logwt"].tolist(), rec["h"].tolist(),
                   rec["nc"].tolist(), rec["logz"].tolist(), rec["delta_logz"].tolist())
        fixed_s = {q: str(fixed[q]) for q in self.parnames if q in fixed}   # a fixed entry overrides (pars.update)
        lines = []
        for i, (v, t) in enumerate(zip(V, tail)):
            pars = ' '.join([fixed_s[q] if q in fixed_s else str(v[col[q]]) for q in self.parnames])
            lines.append('{0} {1} {2} {3} {4} {5} {6} {7} {8} \n'.format(it0 + i, pars, *t))
        self.outff.write(''.join(lines))

    def _row_formatter(self):
        """(library, integer-column flags, source of every parameter column) or None (Python formatting)."""
        if getattr(self, "_row_fmt_for", None) is self.parnames:
            return self._row_fmt
        self._row_fmt_for, self._row_fmt, self._row_buf = self.parnames, None, b""
        try:
            from .. import _lib
            lib = _lib.load()
        except Exception:
            return None
        names, fixed = self.likeobj.fitpars_i, self.fitargs_fixed
        col = {pp: j for j, pp in enumerate(names)}
        src, ints = [], [1]
        for q in self.parnames:
            if q in fixed:
                v = fixed[q]
                if isinstance(v, bool) or not isinstance(v, (int, float, np.integer, np.floating)):
                    return None                              # an LSF vector, a string, ...: str() of it in Python
                src.append(('c', float(v)))
                ints.append(1 if isinstance(v, (int, np.integer)) else 0)
            else:
                src.append(('v', col[q]))
                ints.append(0)
        ints += [0, 0, 0, 0, 1, 0, 0]
        self._row_fmt = (lib, np.array(ints, dtype=np.int32), src)
        return self._row_fmt

    def _runsampler(self, samplerdict):
        npoints = samplerdict.get('npoints', 200)
        bound = samplerdict.get('samplerbounds', 'multi')
        samplemethod = samplerdict.get('samplemethod', 'unif')
        delta_logz_final = samplerdict.get('delta_logz_final', 0.01)
        flushnum = samplerdict.get('flushnum', 10)
        numwalks = samplerdict.get('walks', 25)
        maxiter = samplerdict.get('maxiter', sys.maxsize)
        maxcall = samplerdict.get('maxcall', sys.maxsize)
        seed = samplerdict.get('seed', None)
        starttime = datetime.now()
        if self.verbose:
            print('Static batched nested sampler w/ {0} sampler, {1} walks, {2} number of samples, Ndim = {3}, '
                  'and w/ stopping criteria of dlog(z) = {4}: {5}'.format(
                      samplemethod, numwalks, npoints, self.ndim, delta_logz_final, starttime))
            print('Max Iter: {0} / Max Call: {1}'.format(maxiter, maxcall))
        sys.stdout.flush()
        # prior transform, ln-prior and the random-walk proposals run on the GPU when the fit's priors
        # can be expressed there (everything but derived-quantity priors); host path otherwise
        self.proposer = proposer = self._proposer(samplerdict, max(npoints, int(samplerdict.get('queue_size', npoints))))
        sampler = NestedSampler(
            lnprob_batch, self.priorobj.priortrans_batch, self.ndim, proposer=proposer,
            logl_args=[self.likeobj, self.priorobj], nlive=npoints, bound=bound, sample=samplemethod,
            bootstrap=samplerdict.get('bootstrap', 0), walks=numwalks, slices=samplerdict.get('slices', 5),
            batched=True, queue_size=samplerdict.get('queue_size', npoints),
            pipeline=samplerdict.get('pipeline'),      # None: with device proposals the turn between two queues on the device ('device'), else queues launched ahead from the host's turn
            rstate=np.random.default_rng(seed))
        self.parnames = list(self.likeobj.fitpars_i) + list(self.fitargs['fixedpars'].keys())
        self.fitargs_fixed = dict(self.fitargs['fixedpars'])
        self._initoutput(self.parnames)
        ncall, nit = 0, -1
        t_iter = datetime.now()
        print('Start Sampling @ {}'.format(t_iter))
        # dead points arrive a consumed queue at a time (arrays); rows keep the reference's text format
        for rec in sampler.sample_chunks(dlogz=delta_logz_final, maxiter=maxiter, maxcall=maxcall):
            m = len(rec["logl"])
            self._rows(nit + 1, rec)
            nc_chunk = int(rec["nc"].sum())
            ncall += nc_chunk
            first, nit = nit + 1, nit + m
            if (first // flushnum) != ((nit + 1) // flushnum) or first == 0 or nit >= maxiter:
                self.outff.flush()
                if self.verbose:
                    now = datetime.now()
                    logzvar = float(rec["logzvar"][-1])
                    logzerr = np.sqrt(logzvar) if logzvar > 0. else np.nan
                    sys.stdout.write("\riter: {0:d} | nc: {1:d} | ncall: {2:d} | eff(%): {3:6.3f} | "
                                     "logz: {4:6.3f} +/- {5:6.3f} | loglk: {6:6.3f} | dlogz: {7:6.3f} > {8:6.3f}   | "
                                     "mean(time):  {9:7.5f} | time: {10} \n".format(
                                         nit, int(rec["nc"][-1]), ncall, float(rec["eff"][-1]), float(rec["logz"][-1]),
                                         logzerr, float(rec["logl"][-1]), float(rec["delta_logz"][-1]), delta_logz_final,
                                         (now - t_iter).total_seconds() / max(1, nc_chunk), now))
                    sys.stdout.flush()
            t_iter = datetime.now()
        nit = max(nit, 0)
        rec = sampler.add_live_points_chunk()          # (fitstar.py:410-413: the remaining live points, rows numbered from nit as there)
        self._rows(nit, rec)
        ncall += int(rec["nc"].sum())
        self.outff.close()
        if self.verbose:
            sys.stdout.write('\n')
            print('RUN TIME: {0}'.format(datetime.now() - starttime))
        return sampler

    def _rundynesty(self, samplerdict):
        """The reference's own loop (fitstar.py:260-463) around REAL dynesty, with the GPU behind a pool-shaped
        adapter (sampler/pool.py): dynesty maps its queue of proposals over ``pool`` and every set of simultaneous
        likelihood requests becomes one batch.  Needs dynesty (not part of this image): ``sampler['use_dynesty']``."""
        import dynesty
        from ..sampler.pool import BatchPool
        npoints = samplerdict.get('npoints', 200)
        delta_logz_final = samplerdict.get('delta_logz_final', 0.01)
        flushnum = samplerdict.get('flushnum', 10)
        maxiter = samplerdict.get('maxiter', sys.maxsize)
        maxcall = samplerdict.get('maxcall', sys.maxsize)
        self.pool = pool = BatchPool(self.likeobj, self.priorobj)
        sampler = dynesty.NestedSampler(
            pool.lnprob, pool.prior_transform, self.ndim, nlive=npoints,
            bound=samplerdict.get('samplerbounds', 'multi'), sample=samplerdict.get('samplemethod', 'unif'),
            bootstrap=samplerdict.get('bootstrap', 0), walks=samplerdict.get('walks', 25),
            slices=samplerdict.get('slices', 5), first_update={'min_ncall': -np.inf, 'min_eff': np.inf},
            pool=pool, queue_size=min(pool.size, int(samplerdict.get('queue_size', pool.size))),
            use_pool={'prior_transform': True, 'loglikelihood': True, 'propose_point': False, 'update_bound': False})
        self.parnames = list(self.likeobj.fitpars_i) + list(self.fitargs['fixedpars'].keys())
        self.fitargs_fixed = dict(self.fitargs['fixedpars'])
        self._initoutput(self.parnames)
        ncall, nit = 0, 0
        for it, results in enumerate(sampler.sample(dlogz=delta_logz_final, maxiter=maxiter, maxcall=maxcall)):
            (worst, ustar, vstar, loglstar, logvol, logwt, logz, logzvar,
             h, nc, worst_it, boundidx, bounditer, eff, delta_logz) = results[:15]
            self._row(it, vstar, (loglstar, logvol, logwt, h, nc, logz, delta_logz))
            ncall += nc
            nit = it
            if (it % flushnum) == 0 or it == maxiter:
                self.outff.flush()
                if self.verbose:
                    self._progress(nit, nc, ncall, eff, logz, logzvar, delta_logz, delta_logz_final, 0.0)
        for it2, results in enumerate(sampler.add_live_points()):
            (worst, ustar, vstar, loglstar, logvol, logwt, logz, logzvar,
             h, nc, worst_it, boundidx, bounditer, eff, delta_logz) = results[:15]
            self._row(nit + it2, vstar, (loglstar, logvol, logwt, h, nc, logz, delta_logz))
        self.outff.close()
        return sampler

    def _progress(self, nit, nc, ncall, eff, logz, logzvar, delta_logz, delta_logz_final, mean_time):
        logzerr = np.sqrt(logzvar) if logzvar >= 0. else np.nan
        sys.stdout.write("\riter: {0:d} | nc: {1:d} | ncall: {2:d} | eff(%): {3:6.3f} | "
                         "logz: {4:6.3f} +/- {5:6.3f} | dlogz: {6:6.3f} > {7:6.3f}   | mean(time):  {8:7.5f} | time: {9} \n"
                         .format(nit, nc, ncall, eff, logz, logzerr, delta_logz, delta_logz_final, mean_time,
                                 datetime.now()))
        sys.stdout.flush()

    def _rundysampler(self, samplerdict):
        """The reference's dynamic-sampling driver (fitstar.py:466-645): a baseline static run, then
        batches of 2 x npoints particles placed by the weight function until the stopping function
        fires, each merged into the saved run.  Output rows keep the reference's text: during the
        batches the evidence columns repeat the baseline run's last values (fitstar.py:601-603)."""
        from ..sampler.dynamic import DynamicNestedSampler
        npoints = samplerdict.get('npoints', 200)
        delta_logz_final = samplerdict.get('delta_logz_final', 0.01)
        flushnum = samplerdict.get('flushnum', 10)
        maxiter = samplerdict.get('maxiter', sys.maxsize)
        maxbatch = samplerdict.get('maxbatch', sys.maxsize)
        samplemethod = samplerdict.get('samplemethod', 'unif')
        seed = samplerdict.get('seed', None)
        starttime = datetime.now()
        if self.verbose:
            print('Start Dynamic batched nested sampler w/ {0} sampler, {1} number of samples, Ndim = {2}, '
                  'and w/ stopping criteria of dlog(z) = {3}: {4}'.format(
                      samplemethod, npoints, self.ndim, delta_logz_final, starttime))
        sys.stdout.flush()
        queue = max(2 * npoints, int(samplerdict.get('queue_size', 2 * npoints)))
        self.proposer = proposer = self._proposer(samplerdict, queue)
        dy_sampler = DynamicNestedSampler(
            lnprob_batch, self.priorobj.priortrans_batch, self.ndim, logl_args=[self.likeobj, self.priorobj],
            bound=samplerdict.get('samplerbounds', 'multi'), sample=samplemethod,
            update_interval=samplerdict.get('update_interval', 0.6), bootstrap=samplerdict.get('bootstrap', 0),
            walks=samplerdict.get('walks', 25), slices=samplerdict.get('slices', 5), batched=True,
            queue_size=queue, proposer=proposer,
            rstate=np.random.default_rng(seed))
        self.parnames = list(self.likeobj.fitpars_i) + list(self.fitargs['fixedpars'].keys())
        self.fitargs_fixed = dict(self.fitargs['fixedpars'])
        self._initoutput(self.parnames)
        ncall = nit = 0
        t_iter, dt = datetime.now(), []
        logvol = logwt = h = logz = logzvar = delta_logz = eff = 0.0
        for it, results in enumerate(dy_sampler.sample_initial(nlive=npoints, dlogz=delta_logz_final, maxiter=maxiter)):
            (worst, ustar, vstar, loglstar, logvol, logwt, logz, logzvar,
             h, nc, worst_it, propidx, propiter, eff, delta_logz) = results
            ncall += nc
            nit = it
            self._row(it, vstar, (loglstar, logvol, logwt, h, nc, logz, delta_logz))
            now = datetime.now()
            dt.append((now - t_iter).total_seconds() / float(max(1, nc)))
            t_iter = now
            if (it % flushnum) == 0 or it == maxiter:
                self.outff.flush()
                if self.verbose:
                    self._progress(nit, nc, ncall, eff, logz, logzvar, delta_logz, delta_logz_final, np.mean(dt))
                dt = []
            if it == maxiter:
                break
        sys.stdout.write('\n Finished Initial Static Run {0}\n'.format(datetime.now() - starttime))
        sys.stdout.flush()
        nit += 1
        for n in range(dy_sampler.batch, min(maxiter, maxbatch)):
            res = dy_sampler.results
            res['prop'] = None
            stop, stop_vals = dy_sampler.stopping_function(res, return_vals=True)
            if stop:
                break
            logl_bounds = dy_sampler.weight_function(res)
            it2 = -1
            for it2, results2 in enumerate(dy_sampler.sample_batch(nlive_new=npoints * 2, logl_bounds=logl_bounds,
                                                                   maxiter=maxiter, save_bounds=True)):
                (worst, ustar, vstar, loglstar, nc, worst_it, propidx, propiter, eff) = results2
                ncall += nc
                self._row(nit + it2, vstar, (loglstar, logvol, logwt, h, nc, logz, delta_logz))
                now = datetime.now()
                dt.append((now - t_iter).total_seconds() / float(max(1, nc)))
                t_iter = now
                if (it2 % flushnum) == 0:
                    self.outff.flush()
                    if self.verbose:
                        self._progress(nit + it2, nc, ncall, eff, logz, logzvar, delta_logz, delta_logz_final, np.mean(dt))
                    dt = []
            nit += it2 + 1
            dy_sampler.combine_runs()
        self.outff.close()
        sys.stdout.write('\n Finished Full Dynamic Run {0}\n'.format(datetime.now() - starttime))
        sys.stdout.flush()
        return dy_sampler
