"""The pool-shaped adapter for real dynesty (call site Payne/fitting/fitstar.py:309-321): a fake
sampler drives it exactly as dynesty drives a pool -- `pool.map` over the likelihood for the first live points and
over a point-evolving function (several likelihood calls each) for the queue -- and every set of simultaneous
requests must reach the likelihood as ONE batch."""
import numpy as np
import pytest

from thepayne_amd.sampler.pool import BatchPool


class _Wrapped(object):
    """dynesty's _function_wrapper: the object its pool.map receives."""

    def __init__(self, func, args=(), kwargs=None):
        self.func, self.args, self.kwargs = func, args, kwargs or {}

    def __call__(self, x):
        return self.func(x, *self.args, **self.kwargs)


def _make(size=64):
    batches = []

    def lnprob_batch(theta):
        batches.append(len(theta))
        return -0.5 * np.sum(np.asarray(theta) ** 2, axis=1)

    def priortrans_batch(u):
        return 10.0 * np.asarray(u) - 5.0

    return BatchPool(size=size, lnprob_batch=lnprob_batch, priortrans_batch=priortrans_batch), batches


def test_direct_maps_are_single_batches():
    pool, batches = _make()
    u = np.random.default_rng(0).random((40, 3))
    v = pool.map(_Wrapped(pool.prior_transform), list(u))
    np.testing.assert_allclose(np.array(v), 10.0 * u - 5.0)
    logl = pool.map(_Wrapped(pool.lnprob), v)
    np.testing.assert_allclose(logl, -0.5 * np.sum(np.array(v) ** 2, axis=1))
    assert batches == [40]
    assert pool.lnprob(v[0]) == pytest.approx(logl[0])          # a call from the main thread: a batch of one
    assert batches == [40, 1]


def test_evolving_points_side_by_side_share_batches():
    """'rwalk'-shaped work: every item walks a different number of steps; step k of all items still walking is
    one likelihood batch, and the values each item sees are its own."""
    pool, batches = _make()
    rng = np.random.default_rng(1)
    starts = rng.normal(size=(16, 3))
    nsteps = rng.integers(1, 9, size=16)

    def evolve(arg):
        x, n, seed = arg
        r = np.random.default_rng(seed)
        vals = []
        for _ in range(n):
            x = x + 0.1 * r.normal(size=3)
            vals.append(pool.lnprob(x))
        return x, vals

    out = pool.map(evolve, [(starts[i], int(nsteps[i]), i) for i in range(16)])
    for i, (x, vals) in enumerate(out):
        r = np.random.default_rng(i)
        y = starts[i]
        for k in range(int(nsteps[i])):
            y = y + 0.1 * r.normal(size=3)
            assert vals[k] == pytest.approx(-0.5 * np.sum(y ** 2), rel=1e-14)
        np.testing.assert_array_equal(x, y)
    # step k is evaluated for exactly the items with more than k steps, in one batch
    assert batches == [int(np.sum(nsteps > k)) for k in range(int(nsteps.max()))]


def test_batches_respect_the_engine_size_and_errors_propagate():
    pool, batches = _make(size=8)
    pts = np.random.default_rng(2).normal(size=(20, 2))
    assert len(pool.map(pool.lnprob, list(pts))) == 20
    assert batches == [8, 8, 4]

    def bad(arg):
        if arg == 3:
            raise ValueError("boom")
        return pool.lnprob(np.zeros(2))

    with pytest.raises(ValueError, match="boom"):
        pool.map(bad, list(range(6)))
    assert pool.map(lambda a: a + 1, [1, 2]) == [2, 3]           # a plain function: still a pool
