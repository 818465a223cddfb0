"""The dynamic nested-sampling driver (dynesty's DynamicNestedSampler contract as the
reference uses it, Payne/fitting/fitstar.py:466-645) on analytic problems (CPU, no GPU)."""
import numpy as np
import pytest

from thepayne_amd.sampler import NestedSampler
from thepayne_amd.sampler.dynamic import DynamicNestedSampler, integrate_run, live_counts

SIG = 0.05
NDIM = 3
LOGZ_TRUE = NDIM * np.log(np.sqrt(2 * np.pi) * SIG)


def loglike_batch(V):
    return -0.5 * np.sum(((V - 0.5) / SIG) ** 2, axis=1)


def ptform_batch(U):
    return U.copy()


def _dy(seed=3, **kw):
    args = dict(bound='multi', sample='rwalk', walks=20, batched=True, rstate=np.random.default_rng(seed))
    args.update(kw)
    return DynamicNestedSampler(loglike_batch, ptform_batch, NDIM, **args)


def test_merged_run_of_the_baseline_alone_reproduces_the_static_integral():
    """live_counts + integrate_run on a static run's dead points give back its own columns."""
    s = NestedSampler(loglike_batch, ptform_batch, NDIM, nlive=200, bound='single', sample='unif', batched=True,
                      rstate=np.random.default_rng(1))
    s.run_nested(dlogz=0.1)
    r = s.results
    it = np.asarray(r.samples_it)
    birth = np.where(it > 0, r.logl[np.maximum(it, 1) - 1], -np.inf)
    n = live_counts(r.logl, birth)
    ndead = r.niter - 200
    assert np.all(n[:ndead] == 200) and np.array_equal(n[ndead:], np.arange(200, 0, -1))
    logvol, logwt, logz, logzvar, h = integrate_run(r.logl, n)
    assert np.allclose(logvol, r.logvol, rtol=0, atol=1e-9)
    assert np.allclose(logwt, r.logwt, rtol=0, atol=1e-8)
    assert np.allclose(logz, r.logz, rtol=0, atol=1e-8)
    assert np.allclose(h, r.information, rtol=1e-7, atol=1e-8)
    assert np.allclose(np.sqrt(np.maximum(logzvar, 0)), r.logzerr, rtol=1e-6, atol=1e-8)


def test_reference_loop_on_a_gaussian():
    """The loop of fitstar.py:513-640: baseline, stopping/weight functions, batches, merges."""
    dy = _dy()
    npoints = 100
    first = [t for t in dy.sample_initial(nlive=npoints, dlogz=0.5)]
    assert all(len(t) == 15 for t in first) and dy.batch == 0
    res0 = dy.results
    assert res0.niter == len(first) and np.all(np.diff(res0.logl) >= 0)
    nb = 0
    for n in range(dy.batch, 6):
        res = dy.results
        res['prop'] = None
        stop, vals = dy.stopping_function(res, return_vals=True)
        assert len(vals) == 3 and np.isfinite(vals[2])
        if stop:
            break
        lo, hi = dy.weight_function(res)
        assert lo < hi and hi <= res.logl[-1]
        rows = [t for t in dy.sample_batch(nlive_new=2 * npoints, logl_bounds=(lo, hi), save_bounds=True)]
        assert all(len(t) == 9 for t in rows)
        ll = np.array([t[3] for t in rows])
        assert np.all(np.diff(ll) >= 0) and (lo == -np.inf or ll[0] > lo)
        assert ll[len(rows) - 2 * npoints] >= hi > ll[len(rows) - 2 * npoints - 1]     # ran until the threshold passed logl_max
        before = dy.results.niter
        dy.combine_runs()
        nb += 1
        r = dy.results
        assert r.niter == before + len(rows) and dy.batch == nb
        assert np.all(np.diff(r.logl) >= 0) and np.all(np.diff(r.logvol) < 0)
        # inside the batch's range the live count is the batch's own plus whatever was alive there before
        inside = (r.logl > lo) & (r.logl < hi)
        assert np.all(r.samples_n[inside] >= 2 * npoints) and r.samples_n[inside].max() > 2 * npoints
        assert r.samples_n.min() >= 1
    assert nb >= 1
    r = dy.results
    assert abs(r.logz[-1] - LOGZ_TRUE) < max(0.2, 4 * r.logzerr[-1]), (r.logz[-1], LOGZ_TRUE, r.logzerr[-1])
    w = dy.posterior_weights()
    mean = (w[:, None] * r.samples).sum(0)
    std = np.sqrt((w[:, None] * (r.samples - mean) ** 2).sum(0))
    assert np.all(np.abs(mean - 0.5) < 0.012) and np.all(np.abs(std - SIG) < 0.012), (mean, std)
    assert dy.ncall > r.niter and abs(r.eff - 100.0 * r.niter / dy.ncall) < 1e-9
    assert len(r.batch_nlive) == nb + 1 and r.batch_bounds.shape == (nb + 1, 2)


def test_batches_sharpen_the_posterior_estimate():
    """More batches -> the stopping value (posterior noise) goes down."""
    dy = _dy(seed=8, sample='unif', bound='single')
    for _ in dy.sample_initial(nlive=60, dlogz=0.5):
        pass
    v0 = dy.stopping_function(dy.results, return_vals=True)[1][2]
    for _ in range(4):
        dy.add_batch(nlive=200)
    v1 = dy.stopping_function(dy.results, return_vals=True)[1][2]
    assert v1 < v0, (v0, v1)


def test_batch_from_the_prior_and_argument_checks():
    dy = _dy(seed=2, sample='unif', bound='single')
    with pytest.raises(ValueError):
        next(dy.sample_batch(nlive_new=10, logl_bounds=(-np.inf, 0.0)))
    for _ in dy.sample_initial(nlive=50, dlogz=1.0):
        pass
    res = dy.results
    mid = float(res.logl[len(res.logl) // 3])
    rows = list(dy.sample_batch(nlive_new=40, logl_bounds=(-np.inf, mid)))
    assert rows[0][3] < mid and len(rows) >= 40
    dy.combine_runs()
    r = dy.results
    assert r.samples_n[0] == 90                       # both sets of prior draws are alive at the start
    with pytest.raises(ValueError):
        dy.combine_runs()
    with pytest.raises(ValueError):
        next(dy.sample_batch(nlive_new=10, logl_bounds=(1.0, 0.0)))
    with pytest.raises(ValueError):
        dy.weight_function(r, {'pfrac': 2.0})


def test_run_nested_stops_by_itself():
    dy = _dy(seed=4, sample='unif', bound='single')
    dy.run_nested(nlive_init=100, dlogz_init=0.5, nlive_batch=200, maxbatch=20,
                  stop_kwargs={'post_thresh': 0.08, 'n_mc': 64})
    assert 0 < dy.batch < 20
    assert dy.stopping_function(dy.results, {'post_thresh': 0.12, 'n_mc': 64})
