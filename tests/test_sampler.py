"""The batched nested-sampling driver (dynesty's contract as the reference uses it,
Payne/fitting/fitstar.py:309-338,410-413) on analytic problems (CPU, no GPU)."""
import numpy as np
import pytest

from thepayne_amd.sampler import NestedSampler

SIG = 0.05
NDIM = 3
LOGZ_TRUE = NDIM * np.log(np.sqrt(2 * np.pi) * SIG)      # unnormalised Gaussian under U[0,1]^3


def loglike_batch(V, scale=1.0):
    return -0.5 * np.sum(((V - 0.5) / (SIG * scale)) ** 2, axis=1)


def ptform_batch(U):
    return U.copy()


@pytest.mark.parametrize("method,bound", [("unif", "single"), ("rwalk", "multi"), ("unif", "none")])
def test_evidence_and_posterior_of_a_gaussian(method, bound):
    nlive = 400 if bound != "none" else 200
    s = NestedSampler(loglike_batch, ptform_batch, NDIM, nlive=nlive, bound=bound, sample=method, walks=20,
                      batched=True, rstate=np.random.default_rng(5), queue_size=nlive)
    tuples = []
    for res in s.sample(dlogz=0.05, maxcall=3_000_000):
        assert len(res) == 15
        tuples.append(res)
    worst, ustar, vstar, loglstar, logvol, logwt, logz, logzvar, h, nc, worst_it, bidx, biter, eff, dlz = tuples[-1]
    assert dlz < 0.05 and nc >= 1 and 0 < eff <= 100
    # lnL of dead points is non-decreasing, ln X decreases by ln((n+1)/n) per iteration
    ll = np.array([t[3] for t in tuples])
    assert np.all(np.diff(ll) >= 0)
    lv = np.array([t[4] for t in tuples])
    assert np.allclose(np.diff(lv), -np.log((nlive + 1.0) / nlive))
    n_added = sum(1 for _ in s.add_live_points())
    assert n_added == nlive
    r = s.results
    assert r.niter == len(tuples) + nlive and r.samples.shape == (r.niter, NDIM)
    err = max(0.15, 4 * r.logzerr[-1])
    assert abs(r.logz[-1] - LOGZ_TRUE) < err, (r.logz[-1], LOGZ_TRUE, r.logzerr[-1])
    w = s.posterior_weights()
    mean = (w[:, None] * r.samples).sum(0)
    std = np.sqrt((w[:, None] * (r.samples - mean) ** 2).sum(0))
    assert np.all(np.abs(mean - 0.5) < 0.01) and np.all(np.abs(std - SIG) < 0.012)
    summ = s.summary()
    assert summ.shape == (5 + 5 * NDIM,) and abs(summ[0] - r.logz[-1]) < 1e-12
    with pytest.raises(ValueError):
        next(s.add_live_points())


def test_scalar_callables_with_logl_args_like_dynesty():
    calls = []

    def loglike(v, scale):
        calls.append(1)
        return float(-0.5 * np.sum(((v - 0.5) / (SIG * scale)) ** 2))

    s = NestedSampler(loglike, lambda u: list(u), 2, nlive=100, bound="single", sample="unif", logl_args=[2.0],
                      rstate=np.random.default_rng(1))
    s.run_nested(dlogz=0.5)
    assert len(calls) == s.ncall
    assert abs(s.logz - 2 * np.log(np.sqrt(2 * np.pi) * 2 * SIG)) < 0.5


def test_nan_likelihood_is_never_accepted_and_limits_hold():
    def ll(V):
        out = loglike_batch(V)
        out[V[:, 0] > 0.6] = np.nan        # the reference lets NaN lnL through (likelihood.py:80); the driver must not
        return out
    s = NestedSampler(ll, ptform_batch, NDIM, nlive=100, sample="rwalk", walks=10, batched=True,
                      rstate=np.random.default_rng(2))
    n = sum(1 for _ in s.sample(dlogz=0.01, maxiter=300))
    assert n == 300
    assert np.all(s.live_v[np.isfinite(s.live_logl), 0] <= 0.6)
    with pytest.raises(NotImplementedError):
        NestedSampler(ll, ptform_batch, NDIM, sample="slice")
