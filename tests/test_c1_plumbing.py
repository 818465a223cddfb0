"""C1 of SURVEY 8(d): plumbing without a GPU.

A ``runPayne.py``-shaped ``inputdict`` (demo/runPayne.py:60-150) goes through the
build's ``FitPayne.run`` -- argument parsing, prior object, batched static sampler,
text output -- with the CPU restatement (``oracle``) standing in for the likelihood
object, so everything around the HIP path is exercised here on CPU.  Pass criterion:
the sampler terminates and the posterior brackets the truth.  The GPU twin of this
test (the real likelihood) is tests/test_api_gpu.py::test_fitpayne_run_end_to_end.
"""
import numpy as np

import oracle as O
from thepayne_amd import synth, nnio
from thepayne_amd.fitting.fitstar import FitPayne
from helpers import yst_problem


class OracleBackedLikelihood(object):
    """Same constructor and attributes as fitting/likelihood.py's ``likelihood``
    (Payne/fitting/likelihood.py:5-40), evaluated by the oracle one row at a time."""
    calls = 0

    def __init__(self, fitargs, fitpars, runbools, **kwargs):
        self.fitargs = fitargs
        self.spec_bool, self.phot_bool, self.modpoly_bool, self.photscale_bool = runbools[:4]
        self.fixedpars = fitargs['fixedpars']
        self.fitpars_i = [pp for pp in fitpars[0] if fitpars[1][pp]]
        self.ndim = len(self.fitpars_i)
        with np.load(fitargs['specANNpath']) as z:
            net = {k: z[k] for k in z.files}
        net['kind'] = 'YST1'
        net['resolution'] = float(np.ravel(net['resolution'])[0])
        self.inner = O.OracleLikelihood(net, fitargs['obs_wave_fit'], fitargs['obs_flux_fit'],
                                        fitargs['obs_eflux_fit'], self.fitpars_i, fixedpars=self.fixedpars,
                                        modpoly=self.modpoly_bool)
        self.parsdict = {}

    def lnlikefn(self, pars):
        out = self.inner.lnlikefn(pars)
        self.parsdict = self.inner.parsdict
        type(self).calls += 1
        return out

    def lnlike_batch(self, theta):
        return np.array([self.lnlikefn(row) for row in np.atleast_2d(theta)])


def _inputdict(tmp_path, **sampler_kw):
    raw, obs, flux, eflux = yst_problem("tiny", H=16, line_depth=0.3)
    path = str(tmp_path / "yst.npz")
    nnio.save_npz(path, {k: (np.array([v]) if k == "resolution" else v) for k, v in raw.items() if k != "kind"})
    sampler = {'samplertype': 'Static', 'samplerbounds': 'multi', 'samplemethod': 'rwalk', 'npoints': 64,
               'walks': 10, 'delta_logz_final': 0.5, 'bootstrap': 0, 'flushnum': 100, 'seed': 11,
               'device_proposals': False}
    sampler.update(sampler_kw)
    return {
        'spec': {'obs_wave': obs, 'obs_flux': flux, 'obs_eflux': eflux, 'convertair': False},
        'specANNpath': path, 'NNtype': 'YST1', 'sampler': sampler,
        'priordict': synth.demo_priordict(), 'output': str(tmp_path / 'fit.dat'),
    }


def test_c1_fitpayne_run_with_cpu_restatement(tmp_path):
    F = FitPayne()
    F.likelihood = OracleBackedLikelihood
    OracleBackedLikelihood.calls = 0
    inputdict = _inputdict(tmp_path)
    sampler = F.run(inputdict=inputdict, verbose=False)
    r = sampler.results
    assert r.niter > 64 and np.isfinite(r.logz[-1])
    assert OracleBackedLikelihood.calls >= int(np.sum(r.ncall)) - 64     # every call went through the stand-in
    w = sampler.posterior_weights()
    mean = (w[:, None] * r.samples).sum(0)
    std = np.sqrt((w[:, None] * (r.samples - mean) ** 2).sum(0))
    T = synth.TRUTH
    truth = np.array([T["Teff"], T["logg"], T["feh"], T["afe"], T["vrad"], T["vrot"], T["inst_R"]])
    assert np.all(np.abs(mean - truth) < 5 * std + 1e-3 * np.abs(truth)), (mean, std, truth)
    # output: header + one row per dead point + the final live points (fitstar.py:318-340, :395-420)
    lines = open(inputdict['output']).read().splitlines()
    head = lines[0].split()
    assert head[0] == 'Iter' and head[1:8] == F.likeobj.fitpars_i
    assert head[-7:] == ['log(lk)', 'log(vol)', 'log(wt)', 'h', 'nc', 'log(z)', 'delta(log(z))']
    assert len(lines) == 1 + r.niter
    rows = np.array([[float(x) for x in ln.split()] for ln in lines[1:]])
    assert np.all(np.diff(rows[:, 8]) >= 0)                     # log(lk) never decreases
    # the run ended on the dlogz rule (fitstar.py:356-372), not on a cap
    ndead = r.niter - 64
    assert rows[ndead - 1, -1] <= 0.5 < rows[ndead - 2, -1]
    assert np.allclose(rows[:, 1:8], r.samples, rtol=1e-6, atol=1e-6)


def test_c1_fixed_parameter_and_maxcall(tmp_path):
    """A 'fixed' prior entry drops the parameter from the sampled vector and appends it to
    every output row (fitstar.py:226-231, :330-334); maxcall stops the run early."""
    F = FitPayne()
    F.likelihood = OracleBackedLikelihood
    inputdict = _inputdict(tmp_path, maxcall=3000)
    inputdict['priordict']['Vrot'] = {'fixed': 3.0}
    sampler = F.run(inputdict=inputdict, verbose=False)
    assert F.ndim == 6 and 'Vrot' not in F.likeobj.fitpars_i
    assert int(np.sum(sampler.results.ncall)) < 3000 + 2 * 64
    lines = open(inputdict['output']).read().splitlines()
    head = lines[0].split()
    assert head[1:7] == F.likeobj.fitpars_i and head[7] == 'Vrot'
    rows = np.array([[float(x) for x in ln.split()] for ln in lines[1:]])
    assert np.all(rows[:, 7] == 3.0)


def test_c1_dynamic_sampler_driver(tmp_path):
    """samplertype 'Dynamic' (fitstar.py:466-645): baseline run, then batches of 2 x npoints placed by the
    weight function; the rows of the batches follow the baseline's in the same file."""
    F = FitPayne()
    F.likelihood = OracleBackedLikelihood
    inputdict = _inputdict(tmp_path, samplertype='Dynamic', npoints=50, delta_logz_final=1.0, maxbatch=2)
    dy = F.run(inputdict=inputdict, verbose=False)
    r = dy.results
    assert dy.batch == 2 and list(r.batch_nlive) == [50, 100, 100]
    lines = open(inputdict['output']).read().splitlines()
    assert len(lines) == 1 + r.niter
    rows = np.array([[float(x) for x in ln.split()] for ln in lines[1:]])
    assert np.array_equal(rows[:, 0], np.arange(r.niter))
    # every row of the file is a sample of the merged run
    assert np.allclose(np.sort(rows[:, 8]), r.logl, rtol=1e-12, atol=1e-9)
    w = dy.posterior_weights()
    mean = (w[:, None] * r.samples).sum(0)
    std = np.sqrt((w[:, None] * (r.samples - mean) ** 2).sum(0))
    T = synth.TRUTH
    truth = np.array([T["Teff"], T["logg"], T["feh"], T["afe"], T["vrad"], T["vrot"], T["inst_R"]])
    assert np.all(np.abs(mean - truth) < 5 * std + 1e-3 * np.abs(truth)), (mean, std, truth)


def _fake_dynesty():
    """A stand-in for the dynesty module with the calls FitPayne._rundynesty makes: NestedSampler(loglike, ptform, ndim,
    nlive=, pool=, queue_size=, use_pool=, ...), .sample(dlogz=) yielding the 15-tuples, .add_live_points().  Like the
    real one it reaches the likelihood only through ``pool.map`` (first live points; the queue of evolving points)."""
    import types
    mod = types.ModuleType("dynesty")

    class NestedSampler(object):
        def __init__(self, loglike, ptform, ndim, nlive=50, pool=None, queue_size=1, walks=5, **kw):
            self.L, self.P, self.ndim, self.nlive, self.pool, self.q, self.walks = loglike, ptform, ndim, nlive, pool, queue_size, walks
            self.rng = np.random.default_rng(5)
            self.u = self.rng.random((nlive, ndim))
            self.v = np.array(pool.map(ptform, list(self.u)))
            self.logl = np.array(pool.map(loglike, list(self.v)))
            self.maps = 0

        def _evolve(self, arg):
            u, lstar, seed = arg
            r = np.random.default_rng(seed)
            v, logl, nc = self.P(u), None, 0
            for _ in range(self.walks):
                un = u + 0.02 * r.normal(size=self.ndim)
                if np.any(un <= 0) or np.any(un >= 1):
                    continue
                vn = self.P(un)
                ln = self.L(vn)
                nc += 1
                if ln > lstar:
                    u, v, logl = un, vn, ln
            return u, v, logl, nc

        def sample(self, dlogz=0.5, maxiter=10 ** 9, maxcall=10 ** 9):
            logz, logvol, h, it, queue = -1e300, 0.0, 0.0, 0, []
            while it < maxiter:
                worst = int(np.argmin(self.logl))
                lstar = self.logl[worst]
                logvol -= 1.0 / self.nlive
                logwt = lstar + logvol - np.log(self.nlive)
                logz = np.logaddexp(logz, logwt)
                dz = np.logaddexp(logz, self.logl.max() + logvol) - logz
                yield (worst, self.u[worst].copy(), self.v[worst].copy(), lstar, logvol, logwt, logz, 0.0, h, 1, it, 0, 0, 10.0, dz)
                it += 1
                if dz < dlogz:
                    return
                while True:
                    if not queue:
                        starts = self.rng.integers(0, self.nlive, size=self.q)
                        queue = list(self.pool.map(self._evolve, [(self.u[s], lstar, int(self.rng.integers(1 << 30))) for s in starts]))
                        self.maps += 1
                    u, v, logl, nc = queue.pop()
                    if logl is not None and logl > lstar:
                        self.u[worst], self.v[worst], self.logl[worst] = u, v, logl
                        break

        def add_live_points(self):
            for i in np.argsort(self.logl):
                yield (int(i), self.u[i], self.v[i], self.logl[i], 0.0, 0.0, 0.0, 0.0, 0.0, 1, 0, 0, 0, 10.0, 0.0)

    mod.NestedSampler = NestedSampler
    return mod


def test_c1_real_dynesty_shape_through_the_pool_adapter(tmp_path, monkeypatch):
    """sampler['use_dynesty']: FitPayne hands dynesty a BatchPool (sampler/pool.py) -- the `pool=` of
    Payne/fitting/fitstar.py:309-321 -- and the proposals a queue evaluates at the same time reach the likelihood
    object as batches, not one by one."""
    import sys
    monkeypatch.setitem(sys.modules, "dynesty", _fake_dynesty())
    sizes = []

    class Recording(OracleBackedLikelihood):
        def lnlike_batch(self, theta):
            sizes.append(len(np.atleast_2d(theta)))
            return super().lnlike_batch(theta)

    F = FitPayne()
    F.likelihood = Recording
    inputdict = _inputdict(tmp_path, use_dynesty=True, npoints=32, walks=4, delta_logz_final=5.0, queue_size=16, maxiter=60)
    sampler = F.run(inputdict=inputdict, verbose=False)
    assert sampler.maps >= 1 and F.pool.ncall_batches == len(sizes)
    assert sizes[0] == 32                                        # the first live points: one batch
    assert max(sizes[1:]) > 1 and np.mean(sizes[1:]) > 4         # queued walks share their batches
    lines = open(inputdict['output']).read().splitlines()
    rows = np.array([[float(x) for x in ln.split()] for ln in lines[1:]])
    assert lines[0].split()[0] == 'Iter' and np.all(np.diff(rows[:-32, 8]) >= 0)
