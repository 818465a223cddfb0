"""Batched static nested sampler with dynesty's calling contract.

The reference drives ``dynesty.NestedSampler(lnprobfn, priortrans, ndim, logl_args=...,
nlive=, bound=, sample=, bootstrap=, walks=, slices=)`` and consumes
``.sample(dlogz=, maxiter=, maxcall=)`` / ``.add_live_points()`` as generators of
15-tuples (Payne/fitting/fitstar.py:309-338, :410-413).  dynesty is a third-party
dependency that is absent here and evaluates one point per call; this module provides
the subset of that contract the reference uses, re-designed around BATCHED likelihood
calls: proposals are generated ``queue_size`` at a time as lock-step chains, so every
chain step is one GPU batch (the same "queue" parallelisation dynesty applies with a
pool: a proposal made under an older likelihood threshold is kept iff it still beats the
current one).  The per-iteration bookkeeping (worst point, evidence update, replacement from
the queue) runs in C++ (``payne_ns_consume`` in include/payne_hip.h) a queue at a time;
``sample()`` replays the resulting records as dynesty's 15-tuples, ``sample_chunks()`` hands
them over as arrays for consumers that can work in bulk (FitPayne's output writer).

Algorithm (Skilling 2004/2006 static nested sampling; trapezoid evidence weights):
  * nlive points drawn from the unit cube; at iteration i the worst live point
    (loglstar) dies with ln X_i = -i ln((nlive+1)/nlive) and is replaced by a point
    drawn from the prior restricted to logl > loglstar;
  * replacement proposals: 'unif' = uniform in the (enlarged) bounding ellipsoid(s) of the
    live points in the unit cube ('single': one ellipsoid; 'multi': the live points are split
    recursively by 2-means while the children's ellipsoids hold less than half the parent's
    volume, proposals are uniform in the union; 'none': the cube);
    'rwalk' = `walks` Metropolis steps started from random live points, steps drawn
    uniformly from an ellipsoid with the covariance of the start point's cluster scaled by
    an adaptive factor (target acceptance 0.5);
  * stop when ln(1 + L_max X / Z) < dlogz, then ``add_live_points`` closes the integral.
"""
import ctypes as C
import math

import numpy as np

__all__ = ["NestedSampler", "Results"]


class Results(dict):
    __getattr__ = dict.__getitem__


def _unit_ball(rng, n, ndim):
    z = rng.standard_normal((n, ndim))
    z /= np.linalg.norm(z, axis=1)[:, None]
    return z * rng.uniform(size=(n, 1)) ** (1.0 / ndim)


class _Ell(object):
    """Bounding ellipsoid of a set of unit-cube points: {ctr + axes z, |z| <= 1}."""
    __slots__ = ("ctr", "axes", "axes_unit", "ainv", "logvol", "_cov")

    def __init__(self, u, enlarge):
        n, nd = u.shape
        self.ctr = u.mean(axis=0)
        dev = u - self.ctr
        cov = (dev.T @ dev) / max(1, n - 1)
        cov += 1e-14 * np.eye(nd) * max(1e-300, np.trace(cov) / nd)
        try:
            L = np.linalg.cholesky(cov)
            logdet = float(np.log(np.diagonal(L)).sum())
        except np.linalg.LinAlgError:
            w, Q = np.linalg.eigh(cov)
            L = Q * np.sqrt(np.clip(w, 1e-30, None))
            logdet = float(np.linalg.slogdet(L)[1])
        linv = np.linalg.inv(L)
        r2max = ((dev @ linv.T) ** 2).sum(axis=1).max()      # smallest scaled ellipsoid holding every point
        f = math.sqrt(r2max) * enlarge ** (1.0 / nd)
        self.axes = L * f
        self.axes_unit = L * math.sqrt(nd + 2.0)              # 1-sigma-ish metric for rwalk steps
        self.ainv = linv / f
        self.logvol = logdet + nd * math.log(f)               # up to the unit ball's volume
        self._cov = cov

    @classmethod
    def from_arrays(cls, ctr, axes, axes_unit, ainv, logvol):
        e = cls.__new__(cls)
        e.ctr, e.axes, e.axes_unit, e.ainv, e.logvol, e._cov = ctr, axes, axes_unit, ainv, float(logvol), None
        return e

    def dist2(self, U):
        """Squared radius of the points in this ellipsoid's coordinates (<= 1 inside)."""
        return (((U - self.ctr) @ self.ainv.T) ** 2).sum(axis=1)


def _split_ellipsoids(u, ell, enlarge, budget, vol_dec=0.5):
    """Recursive 2-means decomposition (the rule dynesty's MultiEllipsoid applies, vol_dec = 0.5):
    keep a split while the children hold less than vol_dec of the parent's volume."""
    n, nd = u.shape
    if n < 4 * nd + 2 or budget[0] <= 1:
        return [ell]
    # 2-means on the points themselves (unit-cube coordinates), started from the two halves along the
    # direction of largest spread; the assignment step is one matrix-vector product
    w = u - ell.ctr
    _, vec = np.linalg.eigh(ell._cov)
    lab = (w @ vec[:, -1]) > 0
    for _ in range(8):
        if lab.all() or not lab.any():
            return [ell]
        c0, c1 = w[~lab].mean(axis=0), w[lab].mean(axis=0)
        new = (w @ (c1 - c0)) > 0.5 * (c1 @ c1 - c0 @ c0)
        if np.array_equal(new, lab):
            break
        lab = new
    n1 = int(lab.sum())
    if min(n1, n - n1) < 2 * nd + 1:
        return [ell]
    e0, e1 = _Ell(u[~lab], enlarge), _Ell(u[lab], enlarge)
    if np.logaddexp(e0.logvol, e1.logvol) >= ell.logvol + math.log(vol_dec):
        return [ell]
    budget[0] -= 1
    return _split_ellipsoids(u[~lab], e0, enlarge, budget, vol_dec) + _split_ellipsoids(u[lab], e1, enlarge, budget, vol_dec)


MAX_ELL = 32                                                    # PAYNE_MAX_ELL of include/payne_hip.h


class NestedSampler(object):
    def __init__(self, loglikelihood, prior_transform, ndim, nlive=500, bound='multi', sample='unif',
                 logl_args=None, bootstrap=0, walks=25, slices=5, enlarge=None, rstate=None,
                 batched=False, queue_size=None, update_interval=None, first_update=None, proposer=None,
                 native=True, live_points=None, loglstar=None, overlap_bound=None, pipeline=None, **ignored):
        if sample not in ('unif', 'rwalk', 'slice', 'rslice'):
            raise NotImplementedError("sample=%r: this driver provides 'unif', 'rwalk', 'slice' and 'rslice'" % (sample,))
        if bound not in ('none', 'single', 'multi'):
            raise NotImplementedError("bound=%r: this driver provides 'none', 'single', 'multi'" % (bound,))
        self.ndim = int(ndim)
        self.nlive = int(nlive)
        self.bound = bound
        self.method = sample
        self.walks = int(walks)
        self.slices = int(slices)
        self.enlarge = 1.25 if enlarge is None else float(enlarge)
        self.rng = rstate if rstate is not None else np.random.default_rng()
        self.queue_size = int(queue_size) if queue_size else self.nlive
        args = list(logl_args or [])
        # device-side prior transform / lnprob / random walks (thepayne_amd.sampler.device.DeviceProposer)
        self.proposer = proposer
        if batched:
            self._logl = lambda V: np.asarray(loglikelihood(V, *args), dtype=np.float64)
            self._ptform = lambda U: np.asarray(prior_transform(U), dtype=np.float64)
        else:
            self._logl = lambda V: np.array([loglikelihood(v, *args) for v in V], dtype=np.float64)
            self._ptform = lambda U: np.array([prior_transform(u) for u in U], dtype=np.float64)
        # live points (dynesty's `live_points=[u, v, logl]` hands over an existing set; the dynamic sampler's
        # batches start that way, above the threshold `loglstar`)
        if live_points is not None:
            self.live_u, self.live_v, self.live_logl = [np.array(a, dtype=np.float64) for a in live_points]
            if self.live_u.shape != (self.nlive, self.ndim) or self.live_logl.shape != (self.nlive,):
                raise ValueError("live_points must be (u[nlive, ndim], v[nlive, ndim], logl[nlive])")
        else:
            self.live_u = self.rng.uniform(size=(self.nlive, self.ndim))
        if live_points is not None:
            pass
        elif self.proposer is not None:
            self.live_v, ll = self.proposer.lnprob_u(self.live_u)
            self.live_logl = np.where(np.isnan(ll), -np.inf, ll)
        else:
            self.live_v = self._ptform(self.live_u)
            self.live_logl = self._eval(self.live_v)
        self.live_u = np.ascontiguousarray(self.live_u, dtype=np.float64)
        self.live_v = np.ascontiguousarray(self.live_v, dtype=np.float64)
        self.live_logl = np.ascontiguousarray(self.live_logl, dtype=np.float64)
        self.live_it = np.zeros(self.nlive, dtype=np.int32)
        self.ncall = self.nlive if live_points is None else 0
        self.it = 1
        self.scale = 1.0
        self._pending_nc = 0
        self._clear_queue()
        self._qbuf = None
        self._ell_stack = None
        self._since_update = 0
        # native bookkeeping (C++); native=False keeps the same loop in Python (cross-check / no library)
        self._lib = None
        if native:
            from .. import _lib
            self._lib = _lib.load()
            self._L = _lib
        self._native_bound = self._lib is not None
        self._axes = None
        self._ells = []
        # overlap_bound: with random-walk proposals made on the device, the ellipsoids of the NEXT bound update are fitted on the
        # host while the GPU walks (to the live points as they are when the walk starts: one consumed queue older than an update
        # made at the loop's head would see).  For 'rwalk' the bound is only the proposal metric -- dynesty itself lets it age
        # by update_interval iterations -- so nothing but a slightly staler step shape changes; 'unif' (the bound IS the
        # sampling region) never uses it.  Default: on when the proposer can run the walk in two parts.
        self.overlap_bound = (sample == 'rwalk' and hasattr(proposer, "rwalk_queue")) if overlap_bound is None else bool(overlap_bound)
        self._bound_next = None
        # pipeline: with the whole queue made on the device, the NEXT queue is launched before the current one is consumed, from
        # the live set and the threshold the consumption will leave (payne_ns_peek: the replacements only, no evidence
        # arithmetic) -- the GPU walks while the host does the bookkeeping of the queue before, fits the bound and yields the
        # records.  The queue is the one a launch after the consumption would have made, except for the bound it steps in (one
        # consumed queue older: the proposal metric only, as under overlap_bound).  A queue launched ahead is dropped when the
        # consumption ended elsewhere (a stop condition, maxiter).  Default: on when the proposer can run the queue in two parts.
        # pipeline='device': the turn between two queues is made ON the device (payne_ns_queue_dev_*: the live set lives there, one
        # workgroup merges a queue's proposals into it, adapts the scale, raises the threshold and draws the next start points),
        # and the next queue is enqueued BEFORE the current one has finished -- the GPU goes from queue to queue without waiting for
        # the host, which consumes each queue for the evidence meanwhile.  The host's live SET stays the device's (checked every
        # queue: the device's threshold must lie in [the host's, the live minimum); outside that window the set is re-uploaded); start points are drawn from the device's ordering of
        # it, so a run is the host-turn run statistically, not to the bit.
        # Default (pipeline=None): 'device' wherever it can run -- random-walk proposals from a proposer that keeps the live set on
        # the device, native bookkeeping, nlive + queue_size <= 2048 (the turn kernel sorts the set and the queue in LDS) --, else
        # queues launched ahead from the host's turn (True), else the serial loop.  The evidence for the switch: twenty seeds of each
        # loop on the C2 fit give the same ln Z, the same posterior means and widths and no re-upload
        # (tests/test_sampler_gpu.py::test_turn_on_the_device_is_the_same_run_statistically); it is 4-5 % faster end to end.
        # (the turn kernel is sized by the PROPOSER's k_max, payne_ns_queue_dev_init: a Dynamic run builds its proposer with k_max = 2 npoints)
        dev_ok = sample == 'rwalk' and hasattr(proposer, "queue_dev_launch") and native and \
            self.nlive + max(self.queue_size, int(getattr(proposer, "k_max", self.queue_size))) <= 2048
        if pipeline is None and dev_ok:
            pipeline = 'device'
        if isinstance(pipeline, str) and pipeline == 'host':           # the same queues with the turn on the host
            pipeline = True
        self._dev_turn = isinstance(pipeline, str) and pipeline == 'device'
        if self._dev_turn:
            if not dev_ok:
                raise ValueError("pipeline='device' needs sample='rwalk', native bookkeeping, a proposer with queue_dev_launch and "
                                 "nlive + max(queue_size, proposer.k_max) <= 2048")
            pipeline = False
        self._dev_sync, self._dev_inflight, self._dev_desync = False, 0, 0
        self.pipeline = (sample == 'rwalk' and hasattr(proposer, "rwalk_queue_begin") and native) if pipeline is None else bool(pipeline)
        if self.pipeline and not (sample == 'rwalk' and hasattr(proposer, "rwalk_queue_begin") and native):
            raise ValueError("pipeline=True needs sample='rwalk', native bookkeeping and a proposer with rwalk_queue_begin / _end")
        self._use_turn = self.pipeline and hasattr(proposer, "rwalk_queue_turn")    # (end + scale + peek + begin as one native call)
        self._ahead = None                         # the queue in flight: {"it", "lstar"} the consumption must arrive at, its "seed"
        self._seed_again = None                    # the seed of a queue that was dropped: the queue made in its place takes it
        self._qbufs = [None, None]
        self._cycle = 0                            # queues filled so far
        self._last_m = 0                           # dead points the last consumed queue gave
        self._m_acc = 0
        self._split_wait = 0
        self.nbound = 1
        self.update_interval = int(update_interval) if update_interval and update_interval >= 1 else max(1, int(0.6 * self.nlive))
        # saved run
        self._chunks = []                      # dead-point records, one dict of arrays per consumed queue
        self.logz, self.logzvar, self.h, self.logvol, self.loglstar = -1e300, 0.0, 0.0, 0.0, -1e300
        if loglstar is not None:
            self.loglstar = float(loglstar)
        self.eff = 100.0
        self.added_live = False

    # ---- likelihood plumbing --------------------------------------------------------
    def _eval(self, V):
        ll = self._logl(V)
        return np.where(np.isnan(ll), -np.inf, ll)         # NaN lnL (reference lets it through) can never be accepted

    # ---- bounding ellipsoid ------------------------------------------------------------
    def _update_bound(self):
        if self.bound == 'none' and self.method == 'unif':
            self._axes = None
            return
        if self._bound_next is not None and self._bound_next[4] == self._cycle:    # fitted while the GPU walked this cycle
            ells, stack, split, wait, _ = self._bound_next
            self._bound_next = None
            self._adopt_bound(ells, stack, split, wait)
            return
        self._bound_next = None                # (a fit made ahead and not used leaves no trace: the decomposition schedule is the serial run's)
        self._adopt_bound(*self._fit_bound())

    def _prefetch_bound(self):
        """The fit of the next bound update, made on the host while the GPU walks (see overlap_bound) -- only when that update is
        expected at the head of the next loop (as many dead points as the last queue gave would reach update_interval)."""
        if self._since_update + self._last_m >= self.update_interval:
            self._bound_next = self._fit_bound() + (self._cycle,)

    def _fit_bound(self):
        u = self.live_u
        # the decomposition is tried at every update while it finds several ellipsoids, at every fourth one
        # while the live points keep forming a single cloud
        # (no side effects here: the schedule's counter moves when a fit is ADOPTED -- a fit made ahead by _prefetch_bound may be dropped)
        split, wait = False, self._split_wait
        if self.bound == 'multi':
            wait -= 1
            split = wait <= 0
        if self._native_bound:             # C++ (payne_ns_bound): same arithmetic as _Ell / _split_ellipsoids
            nd, E = self.ndim, MAX_ELL if split else 1
            ctr, lv = np.empty((E, nd)), np.empty(E)
            ax, au, ai = np.empty((E, nd, nd)), np.empty((E, nd, nd)), np.empty((E, nd, nd))
            ne = C.c_int(0)
            rc = self._lib.payne_ns_bound(u.ctypes.data, len(u), nd, float(self.enlarge), int(split), E, ctr.ctypes.data,
                                          ax.ctypes.data, au.ctypes.data, ai.ctypes.data, lv.ctypes.data, C.byref(ne))
            if rc != 0:
                raise RuntimeError("payne_ns_bound failed (%d)" % rc)
            ells = [_Ell.from_arrays(ctr[e], ax[e], au[e], ai[e], lv[e]) for e in range(ne.value)]
            stack = (ctr[:ne.value], au[:ne.value], ai[:ne.value])                  # contiguous: handed to the native queue call
        else:
            whole = _Ell(u, self.enlarge)
            ells = _split_ellipsoids(u, whole, self.enlarge, [MAX_ELL]) if split else [whole]
            stack = tuple(np.stack([getattr(e, k) for e in ells]) for k in ("ctr", "axes_unit", "ainv"))
        if split:
            wait = 1 if len(ells) > 1 else 4
        return ells, stack, split, wait

    def _adopt_bound(self, ells, stack, split, wait):
        self._split_wait = wait
        self._ell_stack = stack
        self._ax_arg = stack[1] if len(stack[1]) > 1 else stack[1][0]      # (one object per bound: the proposer remembers its address)
        self._ells = ells                  # (an update without a split attempt always follows a single-cloud result)
        e0 = self._ells[0]
        self._ctr, self._axes, self._axes_unit = e0.ctr, e0.axes, e0.axes_unit
        self.nbound += 1
        self._since_update = 0

    # ---- proposal generation: fills the queue with batched evaluations -------------------
    def _clear_queue(self):
        nd = self.ndim
        self._q_assign(np.empty((0, nd)), np.empty((0, nd)), np.empty(0), np.empty(0, dtype=np.int32))

    def _set_queue(self, U, V, ll, nc):
        self._q_assign(np.ascontiguousarray(U, dtype=np.float64), np.ascontiguousarray(V, dtype=np.float64),
                       np.ascontiguousarray(ll, dtype=np.float64), np.ascontiguousarray(nc, dtype=np.int32))

    def _q_assign(self, U, V, ll, nc):
        """The queue's four arrays and, once per queue, their addresses for the native consume call."""
        self._qU, self._qV, self._ql, self._qnc, self._qpos = U, V, ll, nc, 0
        self._q_addr = tuple(a.__array_interface__['data'][0] for a in (U, V, ll, nc))

    def _fill_queue(self):
        K, nd, rng = self.queue_size, self.ndim, self.rng
        lstar = self.loglstar
        if self._axes is None and not (self.bound == 'none' and self.method == 'unif'):
            self._update_bound()
        if self.method == 'unif':
            if self.bound == 'none':
                U = rng.uniform(size=(K, nd))
            elif len(self._ells) == 1:
                U = self._ctr + _unit_ball(rng, K, nd) @ self._axes.T
            else:                          # uniform in the union: volume-weighted choice, 1/q thinning of overlaps
                E = self._ells
                lv = np.array([e.logvol for e in E])
                pick = rng.choice(len(E), size=K, p=np.exp(lv - lv.max()) / np.exp(lv - lv.max()).sum())
                ball = _unit_ball(rng, K, nd)
                U = np.stack([e.ctr for e in E])[pick] + np.einsum('kij,kj->ki', np.stack([e.axes for e in E])[pick], ball)
                q = np.sum([e.dist2(U) <= 1.0 for e in E], axis=0)
                U = U[rng.uniform(size=K) * np.maximum(q, 1) < 1.0]
            inside = np.all((U > 0.0) & (U < 1.0), axis=1)
            U = U[inside]
            nin = len(U)
            if nin:
                if self.proposer is not None:
                    V, ll = self.proposer.lnprob_u(U)
                    ll = np.where(np.isnan(ll), -np.inf, ll)
                else:
                    V = self._ptform(U)
                    ll = self._eval(V)
                self._set_queue(U, V, ll, np.ones(nin, dtype=np.int32))
            else:
                self._clear_queue()
                self._update_bound()
            self.ncall += nin
            return
        self._cycle += 1
        self._last_m, self._m_acc = self._m_acc, 0
        if self.method == 'rwalk' and self._dev_turn:
            return self._fill_queue_dev()
        if self.method == 'rwalk' and hasattr(self.proposer, "rwalk_queue"):
            # the whole queue in one native call: start points, ellipsoid assignment, transfers, walk, selection
            # (two host buffers in turn: the queue launched ahead is collected while the one before may still hold proposals)
            self._qbufs.reverse()
            if self._qbufs[0] is None or len(self._qbufs[0][2]) < K:
                self._qbufs[0] = (np.empty((K, nd)), np.empty((K, nd)), np.empty(K), np.empty(K, dtype=np.int32))
            self._qbuf = self._qbufs[0]
            res = None
            if self._ahead is not None:                                      # launched before the last queue was consumed
                ok = self.it == self._ahead["it"] and self.loglstar == self._ahead["lstar"]
                seed, self._ahead = self._ahead["seed"], None
                if ok and self._use_turn:
                    # collected, and the next one launched from the state ITS consumption will leave, in one native call
                    seed = self._queue_seed()
                    ctr, au, ai = self._ell_stack
                    nq, acc, calls, redrawn, idle, self.scale, lnext, m = self.proposer.rwalk_queue_turn(
                        self._qbuf, self.live_u, self.live_v, self.live_logl, K, self._ax_arg, ctr, ai, self.scale, lstar,
                        self.walks, seed)
                    self.ncall += calls
                    self._pending_nc += idle
                    qU, qV, ql, qnc = self._qbuf
                    self._q_assign(qU[:nq], qV[:nq], ql[:nq], qnc[:nq])
                    self._ahead = {"it": self.it + m, "lstar": lnext if m else self.loglstar, "seed": seed}
                    return
                res = self.proposer.rwalk_queue_end(self._qbuf)              # (collected either way: the stream must drain)
                if not ok:
                    res, self._seed_again = None, seed
            if res is None:
                ctr, au, ai = self._ell_stack
                res = self.proposer.rwalk_queue(
                    self.live_u, self.live_v, self.live_logl, K, self._ax_arg, ctr, ai, self.scale, lstar,
                    self.walks, self._queue_seed(), self._qbuf,
                    **({"between": self._prefetch_bound} if self.overlap_bound and not self.pipeline else {}))
            nq, acc, calls, redrawn, idle = res
            self.ncall += calls
            frac = acc / max(1, calls + redrawn)          # a redrawn (out-of-cube) proposal counts as a rejection (dynesty)
            self.scale = min(max(self.scale * math.exp((frac - 0.5) / nd / 0.5), 1e-4), 4.0)
            self._pending_nc += idle
            qU, qV, ql, qnc = self._qbuf
            self._q_assign(qU[:nq], qV[:nq], ql[:nq], qnc[:nq])
            if self.pipeline:
                self._launch_ahead()
            return
        # rwalk / slice: K lock-step chains
        start = rng.integers(0, self.nlive, size=K)
        U, V, ll = self.live_u[start].copy(), self.live_v[start].copy(), self.live_logl[start].copy()
        if self.method in ('slice', 'rslice'):
            self._fill_queue_slice(U, V, ll)
            return
        ell, axes = None, self._axes_unit
        if len(self._ells) > 1:            # each chain steps in the metric of an ellipsoid holding its start point
            d2 = np.stack([e.dist2(U) for e in self._ells])                     # [n_ell, K]
            jitter = rng.uniform(size=d2.shape)
            ell = np.where((d2 <= 1.0).any(axis=0), np.argmax((d2 <= 1.0) * (1.0 + jitter), axis=0), np.argmin(d2, axis=0))
            axes = np.stack([e.axes_unit for e in self._ells])
        if self.proposer is not None:      # all `walks` steps of all K chains in one device call
            kw = {} if ell is None else {"ell": ell}
            U, V, ll, nacc, ncalls = self.proposer.rwalk(U, V, ll, axes, self.scale, lstar, self.walks,
                                                        int(rng.integers(0, 2 ** 62)), **kw)
            ll = np.where(np.isnan(ll), -np.inf, ll)
        else:
            nacc = np.zeros(K, dtype=np.int64)
            ncalls = np.zeros(K, dtype=np.int64)
            for _ in range(self.walks):
                if ell is None:
                    prop = U + self.scale * (_unit_ball(rng, K, nd) @ axes.T)
                else:
                    prop = U + self.scale * np.einsum('kij,kj->ki', axes[ell], _unit_ball(rng, K, nd))
                inside = np.all((prop > 0.0) & (prop < 1.0), axis=1)
                if not inside.any():
                    continue
                pv = self._ptform(prop[inside])
                pl = self._eval(pv)
                ncalls[inside] += 1
                ok = pl > lstar
                idx = np.nonzero(inside)[0][ok]
                U[idx], V[idx], ll[idx] = prop[idx], pv[ok], pl[ok]
                nacc[idx] += 1
        self.ncall += int(ncalls.sum())
        frac = nacc.sum() / max(1, ncalls.sum())
        # dynesty-like scale adaptation towards 50 % acceptance
        self.scale *= math.exp((frac - 0.5) / nd / 0.5)
        self.scale = min(max(self.scale, 1e-4), 4.0)
        moved = nacc > 0                                      # a chain that never moved is a copy of a live point
        self._pending_nc += int(ncalls[~moved].sum())
        self._set_queue(U[moved], V[moved], ll[moved], np.maximum(1, ncalls[moved]))

    def _fill_queue_dev(self, again=True):
        """pipeline='device': keep one queue enqueued behind the one being collected; the device makes the turn between them."""
        K, nd, prop = self.queue_size, self.ndim, self.proposer
        self._qbufs.reverse()
        if self._qbufs[0] is None or len(self._qbufs[0][2]) < K:
            self._qbufs[0] = (np.empty((K, nd)), np.empty((K, nd)), np.empty(K), np.empty(K, dtype=np.int32))
        self._qbuf = self._qbufs[0]
        ctr, au, ai = self._ell_stack
        if not self._dev_sync:                     # the first queue of a loop (or after a mismatch): from the host's live set
            self._dev_epoch = prop.queue_dev_init(self.live_u, self.live_v, self.live_logl, self.scale, self.loglstar)
            prop.queue_dev_launch(K, self._ax_arg, ctr, ai, self.walks, self._queue_seed(), merge=False)
            self._dev_sync, self._dev_inflight = True, 1
        while self._dev_inflight < 2:              # (the bound it steps in: the one the host holds now -- a new one goes up with it)
            prop.queue_dev_launch(K, self._ax_arg, ctr, ai, self.walks, self._queue_seed(), merge=True)
            self._dev_inflight += 1
        nq, acc, calls, redrawn, idle, sc_used, ls_used = prop.queue_dev_collect(self._qbuf)
        self._dev_inflight -= 1
        self.ncall += calls                        # (likelihood calls were made whether or not the queue is used)
        # The threshold the device walked under is the largest lnprob left OUTSIDE its live set after the merge.  The host's own
        # (the last point to die, D) can be lower: a proposal turned away after the last replacement lies between D and the live
        # minimum M.  Every threshold in [D, M) is a valid one to walk under -- proposals are tested again, against the worst live
        # point of their iteration, when they are consumed -- so that window is the test; outside it the two live SETS differ (a loop
        # that stopped in mid-queue and went on; a tie across the set's boundary) and the device starts again from the host's.
        # (Ties INSIDE the set -- the host drops the lower slot of two equal values, the device the higher id -- leave the same
        # multiset of lnprob values on both sides and are not seen here; the points themselves then differ only in which of two
        # equally likely positions stays, which no statistic of the run depends on.)
        if not (self.loglstar <= ls_used < float(self.live_logl.min())) and ls_used != self.loglstar:
            self._dev_desync += 1
            self._dev_drain()
            if not again:
                raise RuntimeError("the device's threshold (%r) is outside the host's window [%r, %r)"
                                   % (ls_used, self.loglstar, float(self.live_logl.min())))
            self._qbufs.reverse()
            return self._fill_queue_dev(again=False)
        frac = acc / max(1, calls + redrawn)
        self.scale = min(max(sc_used * math.exp((frac - 0.5) / nd / 0.5), 1e-4), 4.0)     # (what the device's turn computed too)
        self._pending_nc += idle
        qU, qV, ql, qnc = self._qbuf
        self._q_assign(qU[:nq], qV[:nq], ql[:nq], qnc[:nq])

    def _dev_drain(self):
        """Collect and discard the queues still in flight; the device's live set no longer counts."""
        K, nd = self.queue_size, self.ndim
        scratch = (np.empty((K, nd)), np.empty((K, nd)), np.empty(K), np.empty(K, dtype=np.int32))
        closed = not getattr(getattr(self.proposer, "_handle", None), "value", True)   # (closed before an abandoned generator was finalised)
        # (... or re-initialised since by another sampler on the same proposer -- the dynamic sampler shares one across its runs --:
        #  that init collected and dropped what was in flight; what is in flight now is not ours)
        if getattr(self, "_dev_epoch", None) is not None and getattr(self.proposer, "_dq_epoch", self._dev_epoch) != self._dev_epoch:
            closed = True
        while self._dev_inflight > 0:
            if not closed:                         # any other failure is an error: the C side would still count the queue as in flight
                self.ncall += self.proposer.queue_dev_collect(scratch)[2]
            self._dev_inflight -= 1
        self._dev_sync = False

    def _launch_ahead(self):
        """The next queue, launched from the state the consumption of the current one will leave (payne_ns_peek)."""
        K, nd, n = self.queue_size, self.ndim, self.nlive
        if getattr(self, "_peek", None) is None:
            self._peek = (np.empty((n, nd)), np.empty((n, nd)), np.empty(n))
        pu, pv, pl = self._peek
        lstar, m = C.c_double(self.loglstar), C.c_int(0)
        lu, lv, lll, _ = self._live_addr()
        aU, aV, al, _ = self._q_addr
        rc = self._lib.payne_ns_peek(n, nd, lu, lv, lll, aU, aV, al, len(self._ql), pu.ctypes.data, pv.ctypes.data, pl.ctypes.data,
                                     C.byref(lstar), C.byref(m))
        if rc != 0:
            raise RuntimeError("payne_ns_peek failed (%d)" % rc)
        ctr, au, ai = self._ell_stack
        seed = self._queue_seed()
        self.proposer.rwalk_queue_begin(pu, pv, pl, K, self._ax_arg, ctr, ai, self.scale, lstar.value, self.walks, seed)
        self._ahead = {"it": self.it + m.value, "lstar": lstar.value if m.value else self.loglstar, "seed": seed}

    def _queue_seed(self):
        """The next queue's seed: from the sampler's stream, or the seed of a queue that was launched ahead and dropped (so that
        a run with queues launched ahead draws what the serial run draws)."""
        seed, self._seed_again = self._seed_again, None
        return int(self.rng.integers(0, 2 ** 62)) if seed is None else seed

    def _drop_ahead(self):
        """Collect and discard a queue still in flight (the sampling loop ended before it was needed)."""
        if self._dev_turn and (self._dev_inflight or self._dev_sync):
            self._dev_drain()
        if self._ahead is not None:
            self._seed_again, self._ahead = self._ahead["seed"], None
            self._qbufs.reverse()
            if self._qbufs[0] is None:
                K, nd = self.queue_size, self.ndim
                self._qbufs[0] = (np.empty((K, nd)), np.empty((K, nd)), np.empty(K), np.empty(K, dtype=np.int32))
            try:
                self.proposer.rwalk_queue_end(self._qbufs[0])
            except Exception:                  # (a proposer closed before an abandoned generator was finalised: nothing to collect)
                pass
            self._qbufs.reverse()

    def _eval_u(self, U):
        """Unit-cube points -> (V, lnprob), one batch."""
        if self.proposer is not None:
            V, ll = self.proposer.lnprob_u(U)
            return V, np.where(np.isnan(ll), -np.inf, ll)
        V = self._ptform(U)
        return V, self._eval(V)

    def _fill_queue_slice(self, U, V, ll):
        """Slice sampling (Neal 2003) as dynesty's 'slice' / 'rslice' apply it: `slices` sweeps, each over the
        ndim principal axes of the bounding ellipsoid in random order ('rslice': `slices` random directions);
        per axis a window of one axis length is placed at random around the point, stepped out while its
        ends are above the threshold, then sampled and shrunk until a point above the threshold is found.
        The K chains run in lock step: every round each unfinished chain asks for exactly one likelihood
        value, so a round is one batch."""
        K, nd, rng, lstar = len(U), self.ndim, self.rng, self.loglstar
        if len(self._ells) > 1:
            d2 = np.stack([e.dist2(U) for e in self._ells])
            ell = np.where((d2 <= 1.0).any(axis=0), np.argmax(d2 <= 1.0, axis=0), np.argmin(d2, axis=0))
            A = np.stack([e.axes for e in self._ells])[ell]                       # [K, nd, nd], columns = axes
        else:
            A = np.broadcast_to(self._axes, (K, nd, nd))
        ncalls = np.zeros(K, dtype=np.int64)
        nexpand = ncontract = 0
        n_dir = self.slices * nd if self.method == 'slice' else self.slices
        order = np.stack([np.concatenate([rng.permutation(nd) for _ in range(self.slices)]) for _ in range(K)]) \
            if self.method == 'slice' else None
        rows = np.arange(K)
        for step in range(n_dir):
            if self.method == 'slice':
                axis = self.scale * A[rows, :, order[:, step]]                    # [K, nd]
            else:
                z = rng.standard_normal((K, nd))
                z /= np.linalg.norm(z, axis=1)[:, None]
                axis = self.scale * np.einsum('kij,kj->ki', A, z)
            r = rng.uniform(size=(K, 1))
            left, right = U - r * axis, U + (1.0 - r) * axis

            def value(P, need):
                """lnprob of the rows of P selected by `need` (-inf outside the cube, without a call)."""
                out = np.full(K, -np.inf)
                ask = need & np.all((P > 0.0) & (P < 1.0), axis=1)
                ncalls[need] += 1
                if ask.any():
                    out[ask] = self._eval_u(P[ask])[1]
                return out
            # stepping out, both ends in lock step
            grow_l = np.ones(K, dtype=bool)
            grow_r = np.ones(K, dtype=bool)
            while grow_l.any() or grow_r.any():
                if grow_l.any():
                    nexpand += int(grow_l.sum())
                    still = value(left, grow_l) > lstar
                    left = np.where((grow_l & still)[:, None], left - axis, left)
                    grow_l &= still
                if grow_r.any():
                    nexpand += int(grow_r.sum())
                    still = value(right, grow_r) > lstar
                    right = np.where((grow_r & still)[:, None], right + axis, right)
                    grow_r &= still
            # shrinkage
            todo = np.ones(K, dtype=bool)
            for _ in range(200):
                span = right - left
                prop = left + rng.uniform(size=(K, 1)) * span
                ask = todo & np.all((prop > 0.0) & (prop < 1.0), axis=1)
                ncalls[todo] += 1
                ncontract += int(todo.sum())
                pl = np.full(K, -np.inf)
                pv = None
                if ask.any():
                    pv, got = self._eval_u(prop[ask])
                    pl[ask] = got
                ok = todo & (pl > lstar)
                if ok.any():
                    sel = ok[ask]
                    U[ok], V[ok], ll[ok] = prop[ok], pv[sel], pl[ok]
                side = np.einsum('kj,kj->k', prop - U, span)
                shrink = todo & ~ok
                left = np.where((shrink & (side < 0))[:, None], prop, left)
                right = np.where((shrink & (side >= 0))[:, None], prop, right)
                todo = shrink
                if not todo.any():
                    break
        self.ncall += int(ncalls.sum())
        # dynesty's rule: keep expansions ~ twice the contractions
        self.scale = min(max(self.scale * nexpand / max(1.0, 2.0 * ncontract), 1e-4), 8.0)
        self._set_queue(U, V, ll, np.maximum(1, ncalls))

    # ---- the bookkeeping loop over one queue --------------------------------------------------
    _REC_F = ("logl", "logvol", "logwt", "logz", "logzvar", "h", "delta_logz")     # fp64 columns besides u / v
    _REC_I = ("worst", "nc", "worst_it")                                           # int32 columns

    def _records(self, m):
        """A dict of record arrays for m dead points: views of ONE fp64 block and ONE int32 block (fourteen separate allocations
        and their addresses were a third of a consume call on the host)."""
        nd = self.ndim
        fb = np.empty(m * (2 * nd + len(self._REC_F)))
        ib = np.empty(m * len(self._REC_I), np.int32)
        rec = {"u": fb[:m * nd].reshape(m, nd), "v": fb[m * nd:2 * m * nd].reshape(m, nd)}
        o = 2 * m * nd
        for k in self._REC_F:
            rec[k] = fb[o:o + m]
            o += m
        for j, k in enumerate(self._REC_I):
            rec[k] = ib[j * m:(j + 1) * m]
        rec["_base"] = (fb, ib)
        return rec

    def _live_addr(self):
        """Addresses of the four live-point arrays (asked of numpy only when one of them has been replaced: an address costs
        1.5 us through __array_interface__, more through ndarray.ctypes, and a consume call needs twelve)."""
        arrs = (self.live_u, self.live_v, self.live_logl, self.live_it)
        key = tuple(map(id, arrs))
        if getattr(self, "_live_key", None) != key:
            self._live_key, self._live_ptr, self._live_ref = key, tuple(a.__array_interface__['data'][0] for a in arrs), arrs
        return self._live_ptr

    def _consume(self, dlogz, max_emit, logl_max):
        """Walk the queue: returns (records, stop) with stop in {'queue', 'converged', 'limit', 'logl_max'}."""
        nq = len(self._ql) - self._qpos
        cap = int(min(max_emit, max(nq, 0)))
        rec = self._records(cap)
        if self._lib is not None:
            L = self._L
            st = L.NsState(self.nlive, self.ndim, int(self.it), int(self._pending_nc), self.logz, self.logzvar, self.h,
                           self.logvol, self.loglstar)
            fb, ib = rec["_base"]
            f0, i0, nd = fb.__array_interface__['data'][0], ib.__array_interface__['data'][0], self.ndim
            col = lambda j: f0 + 8 * cap * (2 * nd + j)                           # noqa: E731  (the j-th fp64 column of the block)
            dead = L.NsDead(i0, f0, f0 + 8 * cap * nd, col(0), col(1), col(2), col(3), col(4), col(5), i0 + 4 * cap,
                            i0 + 8 * cap, col(6))
            q0 = self._qpos
            consumed, stop = C.c_int(0), C.c_int(0)
            lu, lv, lll, lit = self._live_addr()
            aU, aV, al, anc = self._q_addr
            m = self._lib.payne_ns_consume(C.byref(st), lu, lv, lll, lit, aU + q0 * nd * 8, aV + q0 * nd * 8,
                                           al + q0 * 8, anc + q0 * 4, nq, float(dlogz),
                                           int(min(max_emit, 2 ** 62)), float(logl_max), C.byref(dead), cap,
                                           C.byref(consumed), C.byref(stop))
            if m < 0:
                raise RuntimeError("payne_ns_consume failed (%d)" % m)
            self._qpos += consumed.value
            self.it, self._pending_nc = int(st.it), int(st.pending_nc)
            self.logz, self.logzvar, self.h, self.logvol, self.loglstar = st.logz, st.logzvar, st.h, st.logvol, st.loglstar
            stop = ('queue', 'converged', 'limit', 'logl_max')[stop.value]
        else:
            m, stop = self._consume_py(rec, dlogz, max_emit, logl_max, cap)
        del rec["_base"]
        if m < cap:
            rec = {k: v[:m] for k, v in rec.items()}
        return rec, stop

    def _consume_py(self, rec, dlogz, max_emit, logl_max, cap):
        """The loop of payne_ns_consume (thepayne_amd/csrc/ns_core.hpp) in Python: the same recurrences, guards and records.  The
        native loop shares its exponentials between logaddexp, H's weights and the two weight terms (seven libm calls per dead
        point instead of nine), so evidence values agree to a few ulp, not to the bit; ln Z at or below -1e299 -- nothing
        finite accumulated yet, e.g. live points at -inf -- records delta_logz = inf in both."""
        n = self.nlive
        dlv = math.log((n + 1.0) / n)
        logdfac = math.log(0.5 * math.expm1(dlv))
        lae = _logaddexp
        ql, qnc, nq = self._ql, self._qnc, len(self._ql)
        m, stop = 0, 'queue'
        while True:
            worst = int(np.argmin(self.live_logl))
            lmin, lmax = float(self.live_logl[worst]), float(self.live_logl.max())
            delta = lae(self.logz, lmax + self.logvol) - self.logz if self.logz > -1e299 else np.inf
            if delta < dlogz:
                stop = 'converged'; break
            if m >= max_emit or m >= cap:
                stop = 'limit'; break
            if lmin >= logl_max:
                stop = 'logl_max'; break
            nc = self._pending_nc
            while self._qpos < nq and not (ql[self._qpos] > lmin):
                nc += int(qnc[self._qpos]); self._qpos += 1
            if self._qpos >= nq:
                self._pending_nc = nc; stop = 'queue'; break
            q = self._qpos
            nc += int(qnc[q])
            self._pending_nc = 0
            logvol = self.logvol - dlv
            logdvol = logdfac + logvol
            logwt = lae(lmin, self.loglstar) + logdvol
            logz_new = lae(self.logz, logwt)
            lz = (math.exp(self.loglstar - logz_new + logdvol) * self.loglstar if self.loglstar > -1e299 else 0.0) + \
                 (math.exp(lmin - logz_new + logdvol) * lmin if math.isfinite(lmin) else 0.0)
            h_new = lz + (math.exp(self.logz - logz_new) * (self.h + self.logz) if self.logz > -1e299 else 0.0) - logz_new
            dh = h_new - self.h
            self.h, self.logz = h_new, logz_new
            self.logzvar += dh * dlv
            self.logvol, self.loglstar = logvol, lmin
            rec["worst"][m] = worst; rec["u"][m] = self.live_u[worst]; rec["v"][m] = self.live_v[worst]
            rec["logl"][m] = lmin; rec["logvol"][m] = logvol; rec["logwt"][m] = logwt; rec["logz"][m] = self.logz
            rec["logzvar"][m] = self.logzvar; rec["h"][m] = self.h; rec["nc"][m] = nc; rec["worst_it"][m] = self.live_it[worst]
            self.live_u[worst], self.live_v[worst] = self._qU[q], self._qV[q]
            self.live_logl[worst], self.live_it[worst] = ql[q], self.it
            self._qpos += 1
            rec["delta_logz"][m] = (lae(self.logz, float(self.live_logl.max()) + self.logvol) - self.logz) if self.logz > -1e299 else np.inf
            self.it += 1
            m += 1
        return m, stop

    # ---- dynesty-contract generators ----------------------------------------------------
    def sample_chunks(self, maxiter=None, maxcall=None, dlogz=0.01, logl_max=np.inf, **ignored):
        """The sampling loop, one dict of record ARRAYS per consumed queue (keys: worst, u, v, logl, logvol,
        logwt, logz, logzvar, h, nc, worst_it, delta_logz, plus it0 / bounditer / eff)."""
        maxiter = np.inf if maxiter is None else maxiter
        maxcall = np.inf if maxcall is None else maxcall
        niter_here = 0
        try:
            while niter_here < maxiter and self.ncall < maxcall:
                if self._since_update >= self.update_interval or (self._axes is None and self.bound != 'none'):
                    self._update_bound()      # (queued proposals stay: each is still tested against the current threshold)
                if self._qpos >= len(self._ql):
                    self._fill_queue()
                room = min(maxiter - niter_here, self.update_interval - self._since_update)
                it0 = self.it
                rec, stop = self._consume(dlogz, max(1, room), logl_max)
                m = len(rec["logl"])
                if m:
                    its = np.arange(it0, it0 + m)
                    rec["it0"] = it0
                    rec["bounditer"] = np.full(m, self.nbound, dtype=np.int64)
                    rec["eff"] = 100.0 * its / self.ncall
                    rec["scale"] = np.full(m, self.scale)
                    self.eff = float(rec["eff"][-1])
                    self._since_update += m
                    self._m_acc += m
                    niter_here += m
                    self._chunks.append(rec)
                    yield rec
                if stop in ('converged', 'logl_max'):
                    break
        finally:
            self._drop_ahead()                 # (a queue launched ahead of a loop that ended: collected, never used)

    def sample(self, maxiter=None, maxcall=None, dlogz=0.01, logl_max=np.inf, **ignored):
        """dynesty's generator contract: one 15-tuple per dead point (fitstar.py:337-338)."""
        for rec in self.sample_chunks(maxiter=maxiter, maxcall=maxcall, dlogz=dlogz, logl_max=logl_max):
            for t in _tuples(rec):
                yield t

    def add_live_points_chunk(self):
        """The remaining live points as ONE chunk of record arrays (the rows dynesty's add_live_points yields one by one: same
        recurrence, same scalar arithmetic, written into one block -- 512 one-row chunks and their tuples were 8 ms of a 70 ms fit)."""
        if self.added_live:
            raise ValueError("live points were already added")
        self.added_live = True
        order = np.argsort(self.live_logl)
        logvol0 = self.logvol
        n = self.nlive
        rec = self._records(n)
        lmax = self.live_logl.max()
        logl_sorted = self.live_logl[order].tolist()
        c_logvol, c_logwt, c_logz, c_logzvar, c_h, c_dlz = ([0.0] * n for _ in range(6))
        for rank in range(n):
            logvol = logvol0 + math.log(1.0 - (rank + 1.0) / (n + 1.0))
            prev_vol = logvol0 + (math.log(1.0 - rank / (n + 1.0)) if rank > 0 else 0.0)
            logdvol = math.log(0.5) + prev_vol + math.log1p(-math.exp(logvol - prev_vol))
            ll = logl_sorted[rank]
            logwt = np.logaddexp(ll, self.loglstar) + logdvol
            logz_new = np.logaddexp(self.logz, logwt)
            lzterm = (math.exp(self.loglstar - logz_new + logdvol) * self.loglstar if self.loglstar > -1e299 else 0.0) + \
                     (math.exp(ll - logz_new + logdvol) * ll if np.isfinite(ll) else 0.0)
            h_new = lzterm + math.exp(self.logz - logz_new) * (self.h + self.logz) - logz_new
            dh = h_new - self.h
            self.h, self.logz = h_new, logz_new
            self.logzvar += dh * (prev_vol - logvol)
            self.loglstar = ll
            self.logvol = logvol
            c_logvol[rank], c_logwt[rank], c_logz[rank], c_logzvar[rank], c_h[rank] = logvol, logwt, self.logz, self.logzvar, self.h
            c_dlz[rank] = 0.0 if rank == n - 1 else np.logaddexp(self.logz, lmax + logvol) - self.logz
        rec["worst"][:] = order; rec["u"][:] = self.live_u[order]; rec["v"][:] = self.live_v[order]; rec["logl"][:] = logl_sorted
        rec["logvol"][:] = c_logvol; rec["logwt"][:] = c_logwt; rec["logz"][:] = c_logz; rec["logzvar"][:] = c_logzvar
        rec["h"][:] = c_h; rec["nc"][:] = 1; rec["worst_it"][:] = self.live_it[order]; rec["delta_logz"][:] = c_dlz
        rec["it0"] = self.it
        rec["bounditer"] = np.full(n, self.nbound, dtype=np.int64)
        rec["eff"] = np.full(n, self.eff)
        rec["scale"] = np.full(n, self.scale)
        self._chunks.append(rec)
        return rec

    def add_live_points(self):
        """Append the remaining live points (dynesty's add_live_points contract: a generator of 15-tuples)."""
        for t in _tuples(self.add_live_points_chunk()):
            yield t

    def run_nested(self, dlogz=0.01, maxiter=None, maxcall=None, add_live=True, **kw):
        for _ in self.sample(dlogz=dlogz, maxiter=maxiter, maxcall=maxcall):
            pass
        if add_live:
            for _ in self.add_live_points():
                pass

    @property
    def results(self):
        ch = self._chunks
        cat = (lambda k: np.concatenate([c[k] for c in ch])) if ch else (lambda k: np.empty(0))
        logz = cat("logz")
        samples = cat("v") if ch else np.empty((0, self.ndim))
        samples_u = cat("u") if ch else np.empty((0, self.ndim))
        return Results(nlive=self.nlive, niter=len(logz), ncall=cat("nc"), eff=self.eff,
                       samples=samples, samples_u=samples_u, samples_id=cat("worst"),
                       samples_it=cat("worst_it"), logl=cat("logl"), logvol=cat("logvol"),
                       logwt=cat("logwt"), logz=logz, logzerr=np.sqrt(np.maximum(cat("logzvar"), 0.0)),
                       information=cat("h"), scale=cat("scale"))

    def posterior_weights(self):
        r = self.results
        w = np.exp(r.logwt - r.logz[-1])
        return w / w.sum()

    def summary(self):
        """Fixed-length fp64 summary of the fit: [logZ, logZerr, niter, ncall, eff] +
        per-parameter [mean, std, q16, q50, q84]."""
        r = self.results
        w = self.posterior_weights()
        out = [r.logz[-1], r.logzerr[-1], float(r.niter), float(self.ncall), float(self.eff)]
        for j in range(self.ndim):
            x = r.samples[:, j]
            mean = float(np.sum(w * x))
            std = float(np.sqrt(max(0.0, np.sum(w * (x - mean) ** 2))))
            o = np.argsort(x)
            cw = np.cumsum(w[o])
            q = [float(x[o][min(len(x) - 1, np.searchsorted(cw, p))]) for p in (0.16, 0.5, 0.84)]
            out += [mean, std] + q
        return np.array(out, dtype=np.float64)


def _logaddexp(a, b):
    if a == b:
        return a + 0.6931471805599453
    m, d = (a, b - a) if a > b else (b, a - b)
    return m + math.log1p(math.exp(d)) if d == d else a + b


def _tuples(rec):
    """Record arrays -> dynesty's 15-tuples (worst, ustar, vstar, loglstar, logvol, logwt, logz, logzvar, h, nc,
    worst_it, boundidx, bounditer, eff, delta_logz)."""
    for i in range(len(rec["logl"])):
        yield (int(rec["worst"][i]), rec["u"][i], rec["v"][i], float(rec["logl"][i]), float(rec["logvol"][i]),
               float(rec["logwt"][i]), float(rec["logz"][i]), float(rec["logzvar"][i]), float(rec["h"][i]),
               int(rec["nc"][i]), int(rec["worst_it"][i]), 0, int(rec["bounditer"][i]), float(rec["eff"][i]),
               float(rec["delta_logz"][i]))
