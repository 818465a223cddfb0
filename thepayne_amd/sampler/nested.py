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
current one).

Algorithm (Skilling 2004/2006 static nested sampling; trapezoid evidence weights):
  * nlive points drawn from the unit cube; at iteration i the worst live point
    (loglstar) dies with ln X_i = -i ln((nlive+1)/nlive) and is replaced by a point
    drawn from the prior restricted to logl > loglstar;
  * replacement proposals: 'unif' = uniform in the (enlarged) bounding ellipsoid of the
    live points in the unit cube ('single'/'multi': one ellipsoid; 'none': the cube);
    'rwalk' = `walks` Metropolis steps started from random live points, steps drawn
    uniformly from an ellipsoid with the live points' covariance scaled by an adaptive
    factor (target acceptance 0.5);
  * stop when ln(1 + L_max X / Z) < dlogz, then ``add_live_points`` closes the integral.
"""
import math
from collections import deque

import numpy as np

__all__ = ["NestedSampler", "Results"]


class Results(dict):
    __getattr__ = dict.__getitem__


def _unit_ball(rng, n, ndim):
    z = rng.standard_normal((n, ndim))
    z /= np.linalg.norm(z, axis=1)[:, None]
    return z * rng.uniform(size=(n, 1)) ** (1.0 / ndim)


class NestedSampler(object):
    def __init__(self, loglikelihood, prior_transform, ndim, nlive=500, bound='multi', sample='unif',
                 logl_args=None, bootstrap=0, walks=25, slices=5, enlarge=None, rstate=None,
                 batched=False, queue_size=None, update_interval=None, first_update=None, proposer=None, **ignored):
        if sample not in ('unif', 'rwalk'):
            raise NotImplementedError("sample=%r: this driver provides 'unif' and 'rwalk'" % (sample,))
        if bound not in ('none', 'single', 'multi'):
            raise NotImplementedError("bound=%r: this driver provides 'none', 'single', 'multi'" % (bound,))
        self.ndim = int(ndim)
        self.nlive = int(nlive)
        self.bound = bound
        self.method = sample
        self.walks = int(walks)
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
        # live points
        self.live_u = self.rng.uniform(size=(self.nlive, self.ndim))
        if self.proposer is not None:
            self.live_v, ll = self.proposer.lnprob_u(self.live_u)
            self.live_logl = np.where(np.isnan(ll), -np.inf, ll)
        else:
            self.live_v = self._ptform(self.live_u)
            self.live_logl = self._eval(self.live_v)
        self.live_it = np.zeros(self.nlive, dtype=int)
        self.ncall = self.nlive
        self.it = 1
        self.scale = 1.0
        self._queue = deque()
        self._queue_logl_min = -np.inf
        self._since_update = 0
        self._axes = None
        self.nbound = 1
        self.update_interval = int(update_interval) if update_interval and update_interval >= 1 else max(1, int(0.6 * self.nlive))
        # saved run
        self.saved = {k: [] for k in ("id", "u", "v", "logl", "logvol", "logwt", "logz", "logzvar", "h", "nc", "it",
                                      "boundidx", "bounditer", "scale")}
        self.logz, self.logzvar, self.h, self.logvol, self.loglstar = -1e300, 0.0, 0.0, 0.0, -1e300
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
        u = self.live_u
        self._ctr = u.mean(axis=0)
        cov = np.cov(u, rowvar=False).reshape(self.ndim, self.ndim)
        cov += 1e-14 * np.eye(self.ndim) * max(1e-300, np.trace(cov) / self.ndim)
        try:
            L = np.linalg.cholesky(cov)
        except np.linalg.LinAlgError:
            w, Q = np.linalg.eigh(cov)
            L = Q * np.sqrt(np.clip(w, 1e-30, None))
        d = np.linalg.solve(L, (u - self._ctr).T)
        r2max = (d ** 2).sum(axis=0).max()                   # smallest scaled ellipsoid holding every live point
        self._axes = L * math.sqrt(r2max) * self.enlarge ** (1.0 / self.ndim)
        self._axes_unit = L * math.sqrt(self.ndim + 2.0)      # 1-sigma-ish metric for rwalk steps
        self.nbound += 1
        self._since_update = 0

    # ---- proposal generation: fills the queue with batched evaluations -------------------
    def _fill_queue(self):
        K, nd, rng = self.queue_size, self.ndim, self.rng
        lstar = self.loglstar
        if self._axes is None and not (self.bound == 'none' and self.method == 'unif'):
            self._update_bound()
        if self.method == 'unif':
            if self.bound == 'none':
                U = rng.uniform(size=(K, nd))
            else:
                U = self._ctr + _unit_ball(rng, K, nd) @ self._axes.T
            inside = np.all((U > 0.0) & (U < 1.0), axis=1)
            V = np.empty((K, nd))
            ll = np.full(K, -np.inf)
            if inside.any():
                if self.proposer is not None:
                    V[inside], lli = self.proposer.lnprob_u(U[inside])
                    ll[inside] = np.where(np.isnan(lli), -np.inf, lli)
                else:
                    V[inside] = self._ptform(U[inside])
                    ll[inside] = self._eval(V[inside])
            nin = int(inside.sum())
            self.ncall += nin
            for i in np.nonzero(inside)[0]:
                self._queue.append((U[i].copy(), V[i].copy(), ll[i], 1))
            if nin == 0:
                self._update_bound()
            return
        # rwalk: K lock-step chains
        start = rng.integers(0, self.nlive, size=K)
        U, V, ll = self.live_u[start].copy(), self.live_v[start].copy(), self.live_logl[start].copy()
        nacc = np.zeros(K, dtype=int)
        ncalls = np.zeros(K, dtype=int)
        if self.proposer is not None:      # all `walks` steps of all K chains in one device call
            U, V, ll, nacc, ncalls = self.proposer.rwalk(U, V, ll, self._axes_unit, self.scale, lstar, self.walks,
                                                        int(rng.integers(0, 2 ** 62)))
            ll = np.where(np.isnan(ll), -np.inf, ll)
        for _ in range(self.walks if self.proposer is None else 0):
            prop = U + self.scale * (_unit_ball(rng, K, nd) @ self._axes_unit.T)
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
        self.scale *= math.exp((frac - 0.5) / nd / (0.5 if frac > 0.5 else 0.5))
        self.scale = min(max(self.scale, 1e-4), 4.0)
        for i in range(K):
            if nacc[i] > 0:                                   # a chain that never moved is a copy of a live point
                self._queue.append((U[i], V[i], ll[i], max(1, int(ncalls[i]))))
            else:
                self._pending_nc = getattr(self, "_pending_nc", 0) + int(ncalls[i])

    def _new_point(self):
        nc = getattr(self, "_pending_nc", 0)
        self._pending_nc = 0
        while True:
            if not self._queue:
                self._fill_queue()
                nc += getattr(self, "_pending_nc", 0)
                self._pending_nc = 0
                continue
            u, v, ll, c = self._queue.popleft()
            nc += c
            if ll > self.loglstar:
                return u, v, ll, nc

    # ---- dynesty-contract generators ----------------------------------------------------
    def sample(self, maxiter=None, maxcall=None, dlogz=0.01, logl_max=np.inf, **ignored):
        maxiter = np.inf if maxiter is None else maxiter
        maxcall = np.inf if maxcall is None else maxcall
        dlv = math.log((self.nlive + 1.0) / self.nlive)
        niter_here = 0
        while True:
            logz_remain = self.live_logl.max() + self.logvol
            delta_logz = np.logaddexp(self.logz, logz_remain) - self.logz if self.logz > -1e299 else np.inf
            if niter_here >= maxiter or self.ncall >= maxcall or delta_logz < dlogz:
                break
            worst = int(np.argmin(self.live_logl))
            ustar, vstar = self.live_u[worst].copy(), self.live_v[worst].copy()
            loglstar_new = self.live_logl[worst]
            worst_it = self.live_it[worst]
            if loglstar_new >= logl_max:
                break
            # evidence update (trapezoid rule on ln X)
            logvol = self.logvol - dlv
            logdvol = math.log(0.5 * math.expm1(dlv)) + logvol          # 0.5 (X_{i-1} - X_i)
            logwt = np.logaddexp(loglstar_new, self.loglstar) + logdvol
            logz_new = np.logaddexp(self.logz, logwt)
            lzterm = (math.exp(self.loglstar - logz_new + logdvol) * self.loglstar if self.loglstar > -1e299 else 0.0) + \
                     (math.exp(loglstar_new - logz_new + logdvol) * loglstar_new if np.isfinite(loglstar_new) else 0.0)
            h_new = lzterm + (math.exp(self.logz - logz_new) * (self.h + self.logz) if self.logz > -1e299 else 0.0) - logz_new
            dh = h_new - self.h
            self.h, self.logz = h_new, logz_new
            self.logzvar += dh * dlv
            self.logvol = logvol
            self.loglstar = loglstar_new
            if self._since_update >= self.update_interval or (self._axes is None and self.bound != 'none'):
                self._update_bound()
                self._queue.clear()
            u, v, ll, nc = self._new_point()
            self._since_update += 1
            self.live_u[worst], self.live_v[worst], self.live_logl[worst], self.live_it[worst] = u, v, ll, self.it
            self.eff = 100.0 * self.it / self.ncall
            logz_remain = self.live_logl.max() + self.logvol
            delta_logz = np.logaddexp(self.logz, logz_remain) - self.logz
            self._save(worst, ustar, vstar, loglstar_new, logvol, logwt, nc, worst_it)
            yield (worst, ustar, vstar, loglstar_new, logvol, logwt, self.logz, self.logzvar, self.h, nc,
                   worst_it, 0, self.nbound, self.eff, delta_logz)
            self.it += 1
            niter_here += 1

    def add_live_points(self):
        """Append the remaining live points (dynesty's add_live_points contract)."""
        if self.added_live:
            raise ValueError("live points were already added")
        self.added_live = True
        order = np.argsort(self.live_logl)
        logvol0 = self.logvol
        n = self.nlive
        for rank, i in enumerate(order):
            logvol = logvol0 + math.log(1.0 - (rank + 1.0) / (n + 1.0))
            prev_vol = logvol0 + (math.log(1.0 - rank / (n + 1.0)) if rank > 0 else 0.0)
            logdvol = math.log(0.5) + prev_vol + math.log1p(-math.exp(logvol - prev_vol))
            ll = self.live_logl[i]
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
            delta_logz = 0.0 if rank == n - 1 else np.logaddexp(self.logz, self.live_logl.max() + logvol) - self.logz
            self._save(int(i), self.live_u[i].copy(), self.live_v[i].copy(), ll, logvol, logwt, 1, self.live_it[i])
            yield (int(i), self.live_u[i].copy(), self.live_v[i].copy(), ll, logvol, logwt, self.logz, self.logzvar,
                   self.h, 1, self.live_it[i], 0, self.nbound, self.eff, delta_logz)

    def run_nested(self, dlogz=0.01, maxiter=None, maxcall=None, add_live=True, **kw):
        for _ in self.sample(dlogz=dlogz, maxiter=maxiter, maxcall=maxcall):
            pass
        if add_live:
            for _ in self.add_live_points():
                pass

    def _save(self, idx, u, v, logl, logvol, logwt, nc, it):
        s = self.saved
        s["id"].append(idx); s["u"].append(u); s["v"].append(v); s["logl"].append(logl)
        s["logvol"].append(logvol); s["logwt"].append(logwt); s["logz"].append(self.logz)
        s["logzvar"].append(self.logzvar); s["h"].append(self.h); s["nc"].append(nc); s["it"].append(it)
        s["boundidx"].append(0); s["bounditer"].append(self.nbound); s["scale"].append(self.scale)

    @property
    def results(self):
        s = self.saved
        logwt = np.array(s["logwt"])
        logz = np.array(s["logz"])
        return Results(nlive=self.nlive, niter=len(s["logl"]), ncall=np.array(s["nc"]), eff=self.eff,
                       samples=np.array(s["v"]), samples_u=np.array(s["u"]), samples_id=np.array(s["id"]),
                       samples_it=np.array(s["it"]), logl=np.array(s["logl"]), logvol=np.array(s["logvol"]),
                       logwt=logwt, logz=logz, logzerr=np.sqrt(np.maximum(np.array(s["logzvar"]), 0.0)),
                       information=np.array(s["h"]), scale=np.array(s["scale"]))

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
