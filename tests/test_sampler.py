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
        NestedSampler(ll, ptform_batch, NDIM, sample="hslice", batched=True)


def test_native_bookkeeping_matches_python_loop():
    """payne_ns_consume (C++) against the same loop in Python: identical dead points, evidence and live set."""
    from thepayne_amd.build import build_lib
    build_lib()
    nd = 4

    def ll(V):
        return -0.5 * np.sum(((V - 0.5) / 0.05) ** 2, axis=1)

    runs = []
    for native in (True, False):
        S = NestedSampler(ll, lambda U: U, nd, nlive=64, bound='single', sample='rwalk', walks=10, batched=True,
                          queue_size=64, rstate=np.random.default_rng(5), native=native)
        S._native_bound = False            # same ellipsoids to the last bit, so that the chains are the same
        tuples = list(S.sample(dlogz=0.05, maxiter=700))
        tuples += list(S.add_live_points())
        runs.append((S, tuples))
    (Sn, tn), (Sp, tp) = runs
    assert len(tn) == len(tp) > 500
    for a, b in zip(tn, tp):
        assert a[0] == b[0] and a[9] == b[9] and a[10] == b[10]                 # worst, nc, worst_it
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])       # ustar, vstar
        np.testing.assert_allclose([a[i] for i in (3, 4, 5, 6, 7, 8, 13, 14)], [b[i] for i in (3, 4, 5, 6, 7, 8, 13, 14)],
                                   rtol=1e-12, atol=1e-12)
    assert np.array_equal(Sn.live_u, Sp.live_u) and Sn.ncall == Sp.ncall and Sn.it == Sp.it
    rn, rp = Sn.results, Sp.results
    assert rn.niter == len(tn) and np.allclose(rn.logz, rp.logz, rtol=1e-12)
    # analytic evidence of the Gaussian in the unit cube: ln((2 pi)^(nd/2) sigma^nd)
    assert abs(rn.logz[-1] - (0.5 * nd * np.log(2 * np.pi) + nd * np.log(0.05))) < 5 * rn.logzerr[-1] + 0.3


def test_native_and_python_records_from_an_empty_evidence_with_points_at_minus_infinity():
    """The regime the evidence guards exist for: ln Z = -1e300 at the start and a third of the first live points at lnL = -inf
    (NaN likelihoods, a prior box wider than the model).  Native and Python bookkeeping give the same records -- delta_logz = inf
    while nothing finite has accumulated, finite and equal to a few ulp afterwards."""
    from thepayne_amd.build import build_lib
    build_lib()
    nd = 3

    def ll(V):
        out = -0.5 * np.sum(((V - 0.5) / 0.1) ** 2, axis=1)
        out[V[:, 0] < 0.33] = -np.inf
        return out

    runs = []
    for native in (True, False):
        S = NestedSampler(ll, lambda U: U, nd, nlive=48, bound='single', sample='rwalk', walks=8, batched=True,
                          queue_size=48, rstate=np.random.default_rng(11), native=native)
        S._native_bound = False
        assert S.logz == -1e300 and np.isneginf(S.live_logl).sum() >= 8
        runs.append(list(S.sample(dlogz=0.1, maxiter=260)))
    tn, tp = runs
    assert len(tn) == len(tp) == 260
    n_inf = 0
    for a, b in zip(tn, tp):
        assert a[0] == b[0] and a[9] == b[9] and np.array_equal(a[1], b[1])
        assert a[3] == b[3] or (np.isneginf(a[3]) and np.isneginf(b[3]))        # loglstar of the dead point
        for i in (4, 5, 6, 7, 8, 14):
            x, y = a[i], b[i]
            if np.isinf(x) or np.isinf(y):
                assert x == y, (i, x, y)
            else:
                assert abs(x - y) <= 1e-11 * max(1.0, abs(x)), (i, x, y)
        n_inf += int(np.isinf(a[14]))
    assert n_inf >= 8 and np.isfinite(tn[-1][14]) and np.isfinite(tn[-1][6])   # the -inf points die first, then the evidence is finite


def test_a_bound_fitted_ahead_and_dropped_leaves_the_decomposition_schedule_alone():
    """_prefetch_bound fits the next bound while the GPU walks; when that fit is not used (the update was not due, or another
    queue came first) the schedule of decomposition attempts must be the serial run's: fitting has no side effects, adopting has."""
    def ll(V):
        return -0.5 * np.sum(((V - 0.5) / 0.1) ** 2, axis=1)
    S = NestedSampler(ll, lambda U: U, 3, nlive=64, bound='multi', sample='rwalk', walks=5, batched=True,
                      rstate=np.random.default_rng(3))
    S._update_bound()
    w0 = S._split_wait
    for _ in range(3):
        S._fit_bound()                         # made ahead, never adopted
        S._bound_next = S._fit_bound() + (S._cycle - 1,)     # a prefetch for another cycle: dropped by _update_bound
    assert S._split_wait == w0
    S._update_bound()
    assert S._bound_next is None and S._split_wait == (w0 - 1 if w0 > 1 else S._split_wait)
    ells, stack, split, wait = S._fit_bound()
    S._adopt_bound(ells, stack, split, wait)
    assert S._split_wait == wait


class _QueueProposer(object):
    """A host stand-in for DeviceProposer's queue calls (rwalk_queue / _begin / _end): K random-walk chains from random live
    points, isotropic steps (the bound is ignored, so a queue depends on the live set, the threshold, the scale and the seed only)."""
    def __init__(self, ll):
        self.ll, self.pending, self.begun, self.dropped = ll, None, 0, 0

    def _make(self, live_u, live_v, live_logl, K, scale, lstar, walks, seed):
        rng = np.random.default_rng(seed)
        st = rng.integers(0, len(live_logl), size=K)
        U, L = np.array(live_u)[st], np.array(live_logl)[st]
        nacc, ncall = np.zeros(K, np.int64), np.zeros(K, np.int64)
        for _ in range(walks):
            P = U + 0.03 * scale * rng.normal(size=U.shape)
            ins = np.all((P > 0) & (P < 1), axis=1)
            lp = np.where(ins, self.ll(P), -np.inf)
            ncall += ins
            ok = ins & (lp > lstar)
            U[ok], L[ok] = P[ok], lp[ok]
            nacc += ok
        mv = nacc > 0
        return U[mv], L[mv], np.maximum(1, ncall[mv]).astype(np.int32), int(nacc.sum()), int(ncall.sum()), int(ncall[~mv].sum())

    def rwalk_queue_begin(self, live_u, live_v, live_logl, K, axes_unit, ctr, ainv, scale, loglstar, walks, seed):
        assert self.pending is None
        self.begun += 1
        self.pending = self._make(live_u, live_v, live_logl, K, scale, loglstar, walks, seed)

    def rwalk_queue_end(self, qbuf):
        U, L, nc, acc, calls, idle = self.pending
        self.pending = None
        n = len(L)
        qbuf[0][:n], qbuf[1][:n], qbuf[2][:n], qbuf[3][:n] = U, U, L, nc
        return n, acc, calls, 0, idle

    def rwalk_queue(self, live_u, live_v, live_logl, K, axes_unit, ctr, ainv, scale, loglstar, walks, seed, qbuf, between=None):
        self.rwalk_queue_begin(live_u, live_v, live_logl, K, axes_unit, ctr, ainv, scale, loglstar, walks, seed)
        if between is not None:
            between()
        return self.rwalk_queue_end(qbuf)

    def lnprob_u(self, U):
        return np.array(U), self.ll(np.asarray(U))


def test_queue_launched_ahead_is_the_queue_launched_after():
    """pipeline=True launches the next queue from the state payne_ns_peek predicts BEFORE the current queue is consumed:
    with a proposer whose queues do not depend on the bound, the run is the serial run to the last bit -- every dead point,
    the evidence, the live set --, through bound updates in mid-queue, a stop on maxiter in mid-queue (the queue in flight is
    collected and dropped), a second call that resumes, and convergence."""
    from thepayne_amd.build import build_lib
    build_lib()
    nd = 3

    def ll(V):
        return -0.5 * np.sum(((V - 0.5) / 0.07) ** 2, axis=1)

    runs = []
    for pipeline in (True, False):
        prop = _QueueProposer(ll)
        S = NestedSampler(ll, lambda U: U, nd, nlive=96, bound='single', sample='rwalk', walks=8, batched=True,
                          queue_size=96, rstate=np.random.default_rng(3), proposer=prop, pipeline=pipeline, overlap_bound=False,
                          update_interval=40)
        assert S.pipeline == pipeline
        recs = list(S.sample_chunks(maxiter=333, dlogz=1e-9))             # stops inside a queue
        assert S._ahead is None and prop.pending is None                  # nothing left in flight
        recs += list(S.sample_chunks(dlogz=0.05, maxcall=400000))         # resumes, runs to convergence
        assert S._ahead is None and prop.pending is None
        tuples = list(S.add_live_points())
        runs.append((S, recs, tuples, prop))
    (Sa, ra, ta, pa), (Sb, rb, tb, pb) = runs
    assert Sa.it == Sb.it > 600 and Sa.ncall == Sb.ncall
    ca = {k: np.concatenate([r[k] for r in ra]) for k in ("worst", "u", "logl", "logz", "logwt", "nc", "delta_logz", "scale")}
    cb = {k: np.concatenate([r[k] for r in rb]) for k in ca}
    for k in ca:
        assert np.array_equal(ca[k], cb[k]), k
    assert np.array_equal(Sa.live_u, Sb.live_u) and Sa.logz == Sb.logz
    assert pa.begun >= pb.begun + 1                                       # (queues launched ahead and dropped at the two stops)
    assert abs(Sa.results.logz[-1] - (0.5 * nd * np.log(2 * np.pi) + nd * np.log(0.07))) < 5 * Sa.results.logzerr[-1] + 0.3


def test_default_loop_is_sized_by_the_proposers_k_max_not_the_queue():
    """The device turn (payne_ns_queue_dev_init) is sized by the proposer's k_max: the dynamic sampler builds its proposer with
    k_max = 2 npoints, fitstar's static one with max(npoints, queue_size).  The default loop must be the device turn only where
    nlive + k_max fits the turn kernel's 2048 slots, the host-turn loop otherwise; asked for by name it must refuse."""
    from thepayne_amd.build import build_lib
    build_lib()

    def ll(V):
        return -0.5 * np.sum(((V - 0.5) / 0.07) ** 2, axis=1)

    class _Dev(_QueueProposer):
        def __init__(self, ll, k_max):
            _QueueProposer.__init__(self, ll)
            self.k_max = k_max

        def queue_dev_launch(self, *a, **k):
            raise AssertionError("the device turn must not be chosen here")

    for nlive, queue, k_max, dev in ((512, 512, 512, True), (700, 700, 1400, False), (1100, 512, 1100, False), (1024, 1024, 1024, True)):
        S = NestedSampler(ll, lambda U: U, 3, nlive=nlive, bound='single', sample='rwalk', walks=4, batched=True, queue_size=queue,
                          rstate=np.random.default_rng(1), proposer=_Dev(ll, k_max))
        assert S._dev_turn == dev and (dev or S.pipeline), (nlive, queue, k_max)
        if not dev:
            with pytest.raises(ValueError):
                NestedSampler(ll, lambda U: U, 3, nlive=nlive, bound='single', sample='rwalk', walks=4, batched=True, queue_size=queue,
                              rstate=np.random.default_rng(1), proposer=_Dev(ll, k_max), pipeline='device')
    # and such a run works through the host-turn loop (it failed with PAYNE_E_UNSUPPORTED at the first queue before)
    prop = _Dev(ll, 1400)
    S = NestedSampler(ll, lambda U: U, 3, nlive=700, bound='single', sample='rwalk', walks=4, batched=True, queue_size=700,
                      rstate=np.random.default_rng(1), proposer=prop)
    n = sum(len(r["logl"]) for r in S.sample_chunks(maxiter=900, dlogz=1e-9))
    assert n == 900 and prop.begun >= 1


def test_peek_predicts_the_consumed_state():
    """payne_ns_peek against payne_ns_consume on the same queue: the same live points and threshold, the same number of dead
    points; ties and -inf values included; an empty and a useless queue."""
    import ctypes as C
    from thepayne_amd import _lib
    from thepayne_amd.build import build_lib
    build_lib()
    lib = _lib.load()
    rng = np.random.default_rng(12)
    n, nd = 50, 3
    A = lambda a: a.ctypes.data                                          # noqa: E731
    for case in range(4):
        lu, lv = rng.uniform(size=(n, nd)), rng.uniform(size=(n, nd))
        ll = np.round(rng.normal(size=n), 1)                              # ties
        ll[:5] = -np.inf
        nq = (0, 40, 40, 200)[case]
        qu, qv = rng.uniform(size=(nq, nd)), rng.uniform(size=(nq, nd))
        ql = np.round(rng.normal(size=nq) + (-9.0 if case == 2 else 0.3), 1)   # case 2: nothing beats the threshold
        qnc = np.ones(nq, np.int32)
        ou, ov, ol = np.empty((n, nd)), np.empty((n, nd)), np.empty(n)
        ls, m = C.c_double(-1e300), C.c_int(-1)
        assert lib.payne_ns_peek(n, nd, A(lu), A(lv), A(ll), A(qu), A(qv), A(ql), nq, A(ou), A(ov), A(ol), C.byref(ls), C.byref(m)) == 0
        cu, cv, cl, cit = lu.copy(), lv.copy(), ll.copy(), np.zeros(n, np.int32)
        st = _lib.NsState(n, nd, 1, 0, -1e300, 0.0, 0.0, 0.0, -1e300)
        cap = max(nq, 1)
        fb, ib = np.empty(cap * (2 * nd + 7)), np.empty(cap * 3, np.int32)
        f0, i0 = A(fb), A(ib)
        col = lambda j: f0 + 8 * cap * (2 * nd + j)                      # noqa: E731
        dead = _lib.NsDead(i0, f0, f0 + 8 * cap * nd, col(0), col(1), col(2), col(3), col(4), col(5), i0 + 4 * cap, i0 + 8 * cap, col(6))
        used, stop = C.c_int(0), C.c_int(0)
        # in two calls, as the sampling loop consumes a queue around a bound update
        m1 = lib.payne_ns_consume(C.byref(st), A(cu), A(cv), A(cl), A(cit), A(qu), A(qv), A(ql), A(qnc), nq, 0.0, 7, np.inf,
                                  C.byref(dead), cap, C.byref(used), C.byref(stop))
        u1 = used.value
        m2 = lib.payne_ns_consume(C.byref(st), A(cu), A(cv), A(cl), A(cit), A(qu[u1:]), A(qv[u1:]), A(ql[u1:]), A(qnc[u1:]), nq - u1,
                                  0.0, 2 ** 40, np.inf, C.byref(dead), cap, C.byref(used), C.byref(stop))
        assert m1 >= 0 and m2 >= 0 and m.value == m1 + m2
        assert np.array_equal(ou, cu) and np.array_equal(ov, cv) and np.array_equal(ol, cl)
        if m.value:
            assert ls.value == st.loglstar
        else:
            assert ls.value == -1e300 and case in (0, 2)


def test_host_bookkeeping_under_address_and_ub_sanitizers(tmp_path):
    """payne_ns_peek / payne_ns_consume / payne_ns_bound (csrc/ns_core.hpp: host code) built with g++ -fsanitize=address,undefined and
    run on random queues with ties, -inf values and an empty queue: the peek's live set equals the consumed one."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    main = tmp_path / "main.cpp"
    main.write_text(r'''
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "thepayne_amd/csrc/ns_core.hpp"
int main() {
  const int n = 37, nd = 3;
  int bad = 0;
  for (int rep = 0; rep < 40; ++rep) {
    srand(rep);
    const int nq = rep == 0 ? 0 : 1 + rand() % 90;
    std::vector<double> lu(n * nd), lv(n * nd), ll(n), qu(nq * nd + 1), qv(nq * nd + 1), ql(nq + 1);
    std::vector<int> qnc(nq + 1, 1), lit(n, 0);
    for (auto& x : lu) x = rand() / (double)RAND_MAX;
    for (auto& x : lv) x = rand() / (double)RAND_MAX;
    for (auto& x : ll) x = std::round(10.0 * rand() / RAND_MAX) / 5.0;
    ll[0] = ll[5] = -INFINITY;
    for (auto& x : qu) x = rand() / (double)RAND_MAX;
    for (auto& x : qv) x = rand() / (double)RAND_MAX;
    for (auto& x : ql) x = std::round(10.0 * rand() / RAND_MAX) / 5.0 + 0.2;
    std::vector<double> ou(n * nd), ov(n * nd), ol(n);
    double ls = -1e300; int m = -1;
    if (payne_ns_peek(n, nd, lu.data(), lv.data(), ll.data(), qu.data(), qv.data(), ql.data(), nq, ou.data(), ov.data(), ol.data(), &ls, &m)) return 2;
    payne_ns_state st{n, nd, 1, 0, -1e300, 0.0, 0.0, 0.0, -1e300};
    const int cap = nq + 1;
    std::vector<int> wi(cap), nc(cap), wit(cap);
    std::vector<double> du(cap * nd), dv(cap * nd), c0(cap), c1(cap), c2(cap), c3(cap), c4(cap), c5(cap), c6(cap);
    payne_ns_dead dead{wi.data(), du.data(), dv.data(), c0.data(), c1.data(), c2.data(), c3.data(), c4.data(), c5.data(), nc.data(), wit.data(), c6.data()};
    int used = 0, stop = 0;
    const int e = payne_ns_consume(&st, lu.data(), lv.data(), ll.data(), lit.data(), qu.data(), qv.data(), ql.data(), qnc.data(), nq, 0.0,
                                   1LL << 40, INFINITY, &dead, cap, &used, &stop);
    if (e != m) ++bad;
    for (int i = 0; i < n * nd; ++i) if (ou[i] != lu[i] || ov[i] != lv[i]) ++bad;
    for (int i = 0; i < n; ++i) if (ol[i] != ll[i]) ++bad;
    if (m > 0 && ls != st.loglstar) ++bad;
  }
  std::vector<double> u(64 * 3), ctr(4 * 3), ax(4 * 9), au(4 * 9), ai(4 * 9), lv(4);
  for (auto& x : u) x = rand() / (double)RAND_MAX;
  int ne = 0;
  if (payne_ns_bound(u.data(), 64, 3, 1.25, 1, 4, ctr.data(), ax.data(), au.data(), ai.data(), lv.data(), &ne) || ne < 1) return 3;
  std::printf("bad=%d\\n", bad);
  return bad != 0;
}''')
    exe = tmp_path / "ns_san"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", root,
                    "-I", os.path.join(root, "include"), str(main), "-o", str(exe)], check=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True)
    assert res.returncode == 0 and "bad=0" in res.stdout, (res.stdout, res.stderr[-2000:])


def test_fitpayne_bulk_rows_equal_single_rows(tmp_path):
    import io
    from thepayne_amd.fitting.fitstar import FitPayne

    class _L:
        fitpars_i = ['Teff', 'log(g)', 'Vrad']
    F = FitPayne()
    F.likeobj, F.fitargs_fixed = _L(), {'[Fe/H]': -0.25, 'Vrad': 3.0}
    F.parnames = ['Teff', 'log(g)', 'Vrad', '[Fe/H]']
    rng = np.random.default_rng(0)
    m = 7
    rec = {"v": rng.normal(size=(m, 3)) * 1e3, "logl": rng.normal(size=m), "logvol": -rng.uniform(size=m),
           "logwt": rng.normal(size=m), "h": rng.uniform(size=m), "nc": rng.integers(1, 40, m).astype(np.int32),
           "logz": rng.normal(size=m), "delta_logz": rng.uniform(size=m) * 10}
    F.outff = io.StringIO()
    F._rows(11, rec)
    bulk = F.outff.getvalue()
    F.outff = io.StringIO()
    for i in range(m):
        F._row(11 + i, rec["v"][i], (rec["logl"][i], rec["logvol"][i], rec["logwt"][i], rec["h"][i], rec["nc"][i],
                                     rec["logz"][i], rec["delta_logz"][i]))
    assert bulk == F.outff.getvalue() and bulk.count('\n') == m
    assert F._row_formatter() is not None                        # ... and the bulk text came from payne_format_rows
    # values str() writes in every notation, an integer-valued fixed parameter, and the Python fall-back (a vector)
    rec["v"][:, 0] = [0.0, -0.0, 1e-5, 1e-4, 1e15, 1e16, 123456789012345680.0]
    rec["logl"][:] = [np.nan, np.inf, -np.inf, 5e-324, 1.7976931348623157e308, -2.5e-300, 1 / 3]
    for fixed in ({'[Fe/H]': 2, 'Vrad': 3.0}, {'[Fe/H]': np.float64(0.1), 'Vrad': np.array([0.5, 0.25])}):
        F.fitargs_fixed = fixed
        F.parnames = list(F.parnames)                            # a new fit: the formatter is chosen again
        F.outff = io.StringIO()
        F._rows(0, rec)
        bulk = F.outff.getvalue()
        F.outff = io.StringIO()
        for i in range(m):
            F._row(i, rec["v"][i], (rec["logl"][i], rec["logvol"][i], rec["logwt"][i], rec["h"][i], rec["nc"][i],
                                    rec["logz"][i], rec["delta_logz"][i]))
        assert bulk == F.outff.getvalue()
        assert (F._row_formatter() is None) == isinstance(fixed['Vrad'], np.ndarray)


def test_ns_consume_stop_conditions():
    """payne_ns_consume: empty queue, emission limit, likelihood ceiling, exhausted queue with carried call counts."""
    import ctypes as C
    from thepayne_amd.build import build_lib
    from thepayne_amd import _lib
    build_lib()
    lib = _lib.load()
    L = _lib
    n, nd = 8, 2
    rng = np.random.default_rng(0)

    def fresh():
        lu = np.ascontiguousarray(rng.uniform(size=(n, nd)))
        return lu, lu.copy(), np.ascontiguousarray(np.arange(n, dtype=np.float64)), np.zeros(n, dtype=np.int32)

    def call(st, live, q, dlogz=1e-9, max_emit=100, logl_max=np.inf, cap=16):
        lu, lv, ll, lit = live
        qu, qv, ql, qnc = q
        rec = {k: np.empty(cap) for k in ("logl", "logvol", "logwt", "logz", "logzvar", "h", "delta_logz")}
        rec.update(worst=np.empty(cap, np.int32), nc=np.empty(cap, np.int32), worst_it=np.empty(cap, np.int32),
                   u=np.empty((cap, nd)), v=np.empty((cap, nd)))
        A = lambda a: a.ctypes.data                                              # noqa: E731
        dead = L.NsDead(*[A(rec[k]) for k in ("worst", "u", "v", "logl", "logvol", "logwt", "logz", "logzvar", "h", "nc",
                                              "worst_it", "delta_logz")])
        consumed, stop = C.c_int(0), C.c_int(0)
        m = lib.payne_ns_consume(C.byref(st), A(lu), A(lv), A(ll), A(lit), A(qu), A(qv), A(ql), A(qnc), len(ql), dlogz,
                                       max_emit, logl_max, C.byref(dead), cap, C.byref(consumed), C.byref(stop))
        return m, consumed.value, stop.value, rec

    def queue(vals, nc=3):
        k = len(vals)
        return (np.ascontiguousarray(rng.uniform(size=(k, nd))), np.ascontiguousarray(rng.uniform(size=(k, nd))),
                np.ascontiguousarray(np.asarray(vals, dtype=np.float64)), np.full(k, nc, dtype=np.int32))

    new_state = lambda: L.NsState(n, nd, 1, 0, -1e300, 0.0, 0.0, 0.0, -1e300)    # noqa: E731
    # empty queue: nothing emitted, nothing consumed
    st, live = new_state(), fresh()
    m, used, stop, _ = call(st, live, queue([]))
    assert (m, used, stop) == (0, 0, L.NS_QUEUE_EMPTY) and st.it == 1
    # proposals below the threshold are skipped and their calls carried over; the good ones replace the worst points
    m, used, stop, rec = call(st, live, queue([-1.0, 50.0, 0.5, 60.0]))
    assert (m, used, stop) == (2, 4, L.NS_QUEUE_EMPTY)
    assert list(rec["logl"][:2]) == [0.0, 1.0] and list(rec["nc"][:2]) == [6, 6] and list(rec["worst"][:2]) == [0, 1]
    assert st.it == 3 and st.pending_nc == 0 and live[2][0] == 50.0 and live[2][1] == 60.0
    m, used, stop, rec = call(st, live, queue([1.5]))                       # does not beat the current worst (2.0)
    assert (m, used, stop) == (0, 1, L.NS_QUEUE_EMPTY) and st.pending_nc == 3
    m, used, stop, rec = call(st, live, queue([70.0]))
    assert m == 1 and rec["nc"][0] == 6                                       # 3 carried + 3
    # emission limit and likelihood ceiling
    m, used, stop, _ = call(st, live, queue([80.0, 81.0, 82.0]), max_emit=2)
    assert (m, used, stop) == (2, 2, L.NS_LIMIT)
    m, used, stop, _ = call(st, live, queue([90.0]), logl_max=4.0)           # worst live point is 5.0 by now
    assert (m, stop) == (0, L.NS_LOGL_MAX)
    # convergence: a huge dlogz stops at once
    m, used, stop, _ = call(st, live, queue([95.0]), dlogz=1e9)
    assert (m, stop) == (0, L.NS_CONVERGED)
    # volumes shrink by ln((n+1)/n) per dead point
    assert abs(st.logvol + (st.it - 1) * np.log((n + 1.0) / n)) < 1e-12


# ---- bound='multi': two well-separated modes -----------------------------------------------------
C_A, C_B = np.array([0.25, 0.3, 0.5]), np.array([0.75, 0.7, 0.5])
LOGZ_TWO = LOGZ_TRUE + np.log(2.0)


def loglike_two_modes(V):
    a = -0.5 * np.sum(((V - C_A) / SIG) ** 2, axis=1)
    b = -0.5 * np.sum(((V - C_B) / SIG) ** 2, axis=1)
    return np.logaddexp(a, b)


@pytest.mark.parametrize("method", ["unif", "rwalk"])
def test_multi_ellipsoid_bound_on_two_modes(method):
    runs = {}
    for bound in ("single", "multi"):
        s = NestedSampler(loglike_two_modes, ptform_batch, NDIM, nlive=400, bound=bound, sample=method, walks=20,
                          batched=True, rstate=np.random.default_rng(12), queue_size=400)
        nell = []
        for _ in s.sample(dlogz=0.05):
            nell.append(len(s._ells))
        for _ in s.add_live_points():
            pass
        runs[bound] = (s, max(nell))
    s, nmax = runs["multi"]
    assert nmax >= 2 and runs["single"][1] == 1
    r = s.results
    assert abs(r.logz[-1] - LOGZ_TWO) < max(0.15, 4 * r.logzerr[-1]), (r.logz[-1], LOGZ_TWO, r.logzerr[-1])
    w = s.posterior_weights()
    in_a = np.linalg.norm(r.samples - C_A, axis=1) < np.linalg.norm(r.samples - C_B, axis=1)
    assert abs(w[in_a].sum() - 0.5) < 0.1                               # both modes carry half the mass
    if method == "unif":                                                # the union wastes far fewer draws
        assert runs["multi"][0].ncall < 0.6 * runs["single"][0].ncall


def test_ellipsoid_decomposition_is_a_cover():
    from thepayne_amd.sampler.nested import _Ell, _split_ellipsoids
    rng = np.random.default_rng(3)
    u = np.concatenate([C_A + 0.02 * rng.standard_normal((150, 3)), C_B + 0.03 * rng.standard_normal((250, 3))])
    whole = _Ell(u, 1.25)
    ells = _split_ellipsoids(u, whole, 1.25, [32])
    assert len(ells) == 2
    covered = np.any([e.dist2(u) <= 1.0 for e in ells], axis=0)
    assert covered.all()
    assert np.logaddexp(ells[0].logvol, ells[1].logvol) < whole.logvol + np.log(0.5)
    # one compact cloud is left alone
    one = _Ell(u[:150], 1.25)
    assert len(_split_ellipsoids(u[:150], one, 1.25, [32])) == 1


def test_native_bound_matches_python_bound():
    """payne_ns_bound (C++) against _Ell / _split_ellipsoids: same ellipsoids for one cloud and for two."""
    import ctypes as C
    from thepayne_amd import _lib
    from thepayne_amd.sampler.nested import _Ell, _split_ellipsoids
    lib = _lib.load()
    rng = np.random.default_rng(3)
    one = 0.5 + 0.02 * rng.standard_normal((300, 5)) @ np.diag([1, 3, 0.5, 2, 1.0])
    two = np.concatenate([C_A + 0.02 * rng.standard_normal((150, 3)), C_B + 0.03 * rng.standard_normal((250, 3))])
    for u, multi, expect in ((one, 0, 1), (one, 1, 1), (two, 1, 2), (two, 0, 1)):
        u = np.ascontiguousarray(u)
        n, nd = u.shape
        ctr, lv = np.empty((32, nd)), np.empty(32)
        ax, au, ai = np.empty((32, nd, nd)), np.empty((32, nd, nd)), np.empty((32, nd, nd))
        ne = C.c_int(0)
        rc = lib.payne_ns_bound(u.ctypes.data, n, nd, 1.25, multi, 32, ctr.ctypes.data, ax.ctypes.data, au.ctypes.data,
                                ai.ctypes.data, lv.ctypes.data, C.byref(ne))
        assert rc == 0 and ne.value == expect
        whole = _Ell(u, 1.25)
        ref = _split_ellipsoids(u, whole, 1.25, [32]) if multi else [whole]
        assert len(ref) == expect
        order = np.argsort([e.ctr[0] for e in ref])
        mine = np.argsort(ctr[:expect, 0])
        for i, j in zip(order, mine):
            e = ref[i]
            np.testing.assert_allclose(ctr[j], e.ctr, rtol=1e-12)
            np.testing.assert_allclose(ax[j], e.axes, rtol=1e-9, atol=1e-15)
            np.testing.assert_allclose(au[j], e.axes_unit, rtol=1e-9, atol=1e-15)
            np.testing.assert_allclose(ai[j], e.ainv, rtol=1e-8, atol=1e-9)
            assert abs(lv[j] - e.logvol) < 1e-9
            assert np.all((((u - ctr[j]) @ ai[j].T) ** 2).sum(axis=1)[e.dist2(u) <= 1.0] <= 1.0 + 1e-9)
    assert lib.payne_ns_bound(None, 10, 3, 1.25, 0, 1, None, None, None, None, None, None) < 0


@pytest.mark.parametrize("method", ["slice", "rslice"])
def test_slice_sampling_on_a_gaussian(method):
    s = NestedSampler(loglike_batch, ptform_batch, NDIM, nlive=300, bound='single', sample=method, slices=3,
                      batched=True, rstate=np.random.default_rng(17), queue_size=300)
    s.run_nested(dlogz=0.05)
    r = s.results
    assert abs(r.logz[-1] - LOGZ_TRUE) < max(0.15, 4 * r.logzerr[-1]), (r.logz[-1], LOGZ_TRUE, r.logzerr[-1])
    w = s.posterior_weights()
    mean = (w[:, None] * r.samples).sum(0)
    std = np.sqrt((w[:, None] * (r.samples - mean) ** 2).sum(0))
    assert np.all(np.abs(mean - 0.5) < 0.012) and np.all(np.abs(std - SIG) < 0.012), (mean, std)
    assert np.all(np.diff(r.logl[:-300]) >= 0) and 1e-4 < s.scale < 8.0
