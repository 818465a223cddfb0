"""Batched dynamic nested sampler with the part of dynesty's ``DynamicNestedSampler``
contract the reference drives (Payne/fitting/fitstar.py:466-645):

    dy = DynamicNestedSampler(lnprobfn, priortrans, ndim, logl_args=[...], bound=, sample=,
                              update_interval=, bootstrap=, walks=, slices=)
    for results in dy.sample_initial(nlive=, dlogz=, maxiter=): ...      # 15-tuples
    for n in range(dy.batch, maxiter):
        res = dy.results; res['prop'] = None
        stop, vals = dy.stopping_function(res, return_vals=True)
        if stop: break
        logl_bounds = dy.weight_function(res)
        for results in dy.sample_batch(nlive_new=, logl_bounds=, maxiter=, save_bounds=True): ...  # 9-tuples
        dy.combine_runs()

dynesty is a third-party dependency that is absent here; this is a restatement of the
published algorithm (Higson et al. 2019, "Dynamic nested sampling"; Speagle 2020, dynesty
section 3) on top of this package's batched static sampler, so every likelihood call of
the baseline run and of each batch is a GPU batch:

  * baseline: a static run with `nlive` points, final live points appended;
  * weight function: importance = pfrac * posterior mass + (1-pfrac) * remaining evidence,
    a batch is placed where importance > maxfrac * max, padded by `pad` samples;
  * batch: `nlive_new` points are drawn from the prior restricted to logl > logl_min
    (proposals started from the particles that were alive at that threshold), then
    evolved as a static run until the threshold passes logl_max; their final live points
    are appended;
  * merge: every dead point carries the threshold it was born under, so the number of
    live particles at any logl is (#born below) - (#dead below); volumes shrink by
    n/(n+1) per dead point and the evidence integral is re-accumulated over the merged run;
  * stopping: n_mc simulated runs (strands resampled with replacement, volumes jittered
    as Beta(n,1)); stop when pfrac * (std/mean of the KL divergences) / post_thresh +
    (1-pfrac) * std(logz) / evid_thresh <= 1.
"""
import math

import numpy as np

from .nested import NestedSampler, Results, _logaddexp

__all__ = ["DynamicNestedSampler", "integrate_run", "live_counts"]

_NEG = -1e300


def live_counts(logl, birth):
    """Number of live particles just before each death of a run sorted by logl: particles
    born under a threshold below logl_i, minus those already dead."""
    logl = np.asarray(logl, dtype=np.float64)
    sb = np.sort(birth)
    born = np.searchsorted(sb, logl, side='left')                  # birth < logl_i
    from_prior = np.searchsorted(sb, -np.inf, side='right')        # born under no threshold: alive from the start,
    born = np.maximum(born, from_prior)                            # also for a dead point at logl = -inf
    dead = np.arange(len(logl))                                    # every earlier entry of the sorted run
    return np.maximum(born - dead, 1).astype(np.int64)


def _volumes(n, ln_t=None):
    """ln X_i for live counts n_i: expected shrinkage n/(n+1) per dead point, or the given draws."""
    n = np.asarray(n, dtype=np.float64)
    return np.cumsum(np.log(n / (n + 1.0)) if ln_t is None else ln_t)


def _weights(logl, logvol):
    """Trapezoid evidence weights and the running ln Z (vectorised part of the integral)."""
    prev_vol = np.concatenate([[0.0], logvol[:-1]])
    logdvol = math.log(0.5) + prev_vol + np.log1p(-np.exp(logvol - prev_vol))
    prev_l = np.concatenate([[_NEG], logl[:-1]])
    logwt = np.logaddexp(logl, prev_l) + logdvol
    return logdvol, logwt, np.logaddexp.accumulate(logwt)


def integrate_run(logl, n):
    """Evidence integral over a run with a varying number of live points: returns logvol, logwt,
    logz, logzvar, h per dead point -- the same recurrences as the static sampler's loop
    (thepayne_amd/csrc/ns_core.hpp) with ln((n_i+1)/n_i) as the per-point compression."""
    logl = np.asarray(logl, dtype=np.float64)
    n = np.asarray(n, dtype=np.float64)
    logvol = _volumes(n)
    logdvol, logwt, logz = _weights(logl, logvol)
    dlv = np.log((n + 1.0) / n)
    m = len(logl)
    h, logzvar = np.empty(m), np.empty(m)
    hp, zp, lp, var = 0.0, _NEG, _NEG, 0.0
    for i in range(m):
        li, zi, dv = logl[i], logz[i], logdvol[i]
        lz = (math.exp(lp - zi + dv) * lp if lp > -1e299 else 0.0) + (math.exp(li - zi + dv) * li if math.isfinite(li) else 0.0)
        hn = lz + (math.exp(zp - zi) * (hp + zp) if zp > -1e299 else 0.0) - zi
        var += (hn - hp) * dlv[i]
        h[i], logzvar[i] = hn, var
        hp, zp, lp = hn, zi, li
    return logvol, logwt, logz, logzvar, h


class DynamicNestedSampler(object):
    def __init__(self, loglikelihood, prior_transform, ndim, bound='multi', sample='unif', logl_args=None,
                 update_interval=None, bootstrap=0, walks=25, slices=5, enlarge=None, rstate=None, batched=False,
                 queue_size=None, proposer=None, native=True, **ignored):
        self.ndim = int(ndim)
        self.rng = rstate if rstate is not None else np.random.default_rng()
        self._kw = dict(bound=bound, sample=sample, logl_args=logl_args, bootstrap=bootstrap, walks=walks, slices=slices,
                        enlarge=enlarge, batched=batched, proposer=proposer, native=native)
        self._fn = (loglikelihood, prior_transform)
        self._update_interval = update_interval
        self._queue_size = queue_size
        self.batch = 0
        self.base = False
        self.ncall = 0
        self.eff = 100.0
        self.saved = None                     # merged run: dict of arrays sorted by logl
        self.new = None                       # the last batch, before combine_runs()
        self.batch_bounds, self.batch_nlive = [], []
        self.sampler = None

    # ---- helpers ----------------------------------------------------------------------------------
    def _static(self, nlive, **kw):
        ui = self._update_interval
        if ui is not None and ui < 1:        # dynesty: a fraction of nlive
            ui = max(1, int(round(ui * nlive)))
        args = dict(self._kw)
        args.update(kw)
        return NestedSampler(self._fn[0], self._fn[1], self.ndim, nlive=nlive, update_interval=ui,
                             queue_size=min(nlive, self._queue_size or nlive), rstate=self.rng, **args)

    @staticmethod
    def _run_arrays(S, birth0, id0):
        """Dead points of a finished static run (final live points included) as arrays with the
        threshold each particle was born under and a run-wide particle id."""
        r = S.results
        logl = np.asarray(r.logl, dtype=np.float64)
        it = np.asarray(r.samples_it, dtype=np.int64)                 # 0 = initial point, k = born at the k-th death
        birth = np.where(it > 0, logl[np.maximum(it, 1) - 1], birth0)
        return dict(u=np.asarray(r.samples_u), v=np.asarray(r.samples), logl=logl, birth=birth,
                    id=np.asarray(r.samples_id, dtype=np.int64) + id0, nc=np.asarray(r.ncall, dtype=np.int64),
                    scale=np.asarray(r.scale, dtype=np.float64))

    def _finish(self, run, batch_idx):
        """Sort by logl, count live particles, integrate; `run` holds u, v, logl, birth, id, nc, batch."""
        o = np.argsort(run["logl"], kind='stable')
        run = {k: v[o] for k, v in run.items()}
        run["n"] = live_counts(run["logl"], run["birth"])
        run["logvol"], run["logwt"], run["logz"], run["logzvar"], run["h"] = integrate_run(run["logl"], run["n"])
        return run

    # ---- baseline run -----------------------------------------------------------------------------
    def sample_initial(self, nlive=500, dlogz=0.01, maxiter=None, maxcall=None, logl_max=np.inf, live_points=None,
                       **ignored):
        """The baseline static run: yields dynesty's 15-tuples, the final live points included."""
        self.nlive0 = int(nlive)
        S = self.sampler = self._static(self.nlive0, live_points=live_points)
        for t in S.sample(dlogz=dlogz, maxiter=maxiter, maxcall=maxcall, logl_max=logl_max):
            yield t
        for t in S.add_live_points():
            yield t
        run = self._run_arrays(S, -np.inf, 0)
        run["batch"] = np.zeros(len(run["logl"]), dtype=np.int64)
        self.saved = self._finish(run, 0)
        self._next_id = self.nlive0
        self.ncall = int(S.ncall)
        self.eff = 100.0 * len(run["logl"]) / max(1, self.ncall)
        self.base = True
        self.batch_bounds = [(-np.inf, np.inf)]
        self.batch_nlive = [self.nlive0]

    # ---- where to put the next batch ------------------------------------------------------------------
    def weight_function(self, results, args=None, return_weights=False):
        """dynesty's default importance function: (logl_min, logl_max) of the next batch."""
        args = args or {}
        pfrac, maxfrac, pad = args.get('pfrac', 0.8), args.get('maxfrac', 0.8), args.get('pad', 1)
        if not 0.0 <= pfrac <= 1.0 or not 0.0 <= maxfrac <= 1.0 or pad < 0:
            raise ValueError("pfrac, maxfrac must be in [0, 1] and pad >= 0")
        logl, logz, logvol, logwt = results['logl'], results['logz'], results['logvol'], results['logwt']
        n = results['samples_n']
        logz_remain = logl[-1] + logvol[-1]
        logz_tot = np.logaddexp(logz[-1], logz_remain)
        with np.errstate(divide='ignore', invalid='ignore'):
            logzin = logz_tot + np.log1p(-np.exp(np.minimum(logz - logz_tot, 0.0)))       # ln(Z_tot - Z_i)
        logzw = logzin - np.log(n)
        logzw = np.where(np.isfinite(logzw), logzw, -np.inf)
        zweight = np.exp(logzw - np.max(logzw))
        zweight /= zweight.sum()
        pweight = np.exp(logwt - logz[-1])
        pweight /= pweight.sum()
        weight = (1.0 - pfrac) * zweight + pfrac * pweight
        idx = np.nonzero(weight > maxfrac * weight.max())[0]
        lo, hi = int(idx.min()) - pad, min(int(idx.max()) + pad, len(logl) - 1)
        logl_min = -np.inf if lo < 0 else float(logl[lo])
        logl_max = float(logl[hi])
        if return_weights:
            return (logl_min, logl_max), (pweight, zweight, weight)
        return (logl_min, logl_max)

    # ---- one batch ---------------------------------------------------------------------------------
    def _alive_at(self, logl_min):
        """Particles of the merged run that were alive under the threshold logl_min; each is a
        draw from the prior restricted to logl > logl_min."""
        s = self.saved
        alive = np.nonzero((s["birth"] <= logl_min) & (s["logl"] > logl_min))[0]
        need = 2 * self.ndim + 2
        if len(alive) < need:                 # too few to shape a proposal: take the next dead points as well
            above = np.nonzero(s["logl"] > logl_min)[0][:max(need, self.nlive0)]
            alive = np.union1d(alive, above)
        return alive

    def sample_batch(self, nlive_new=500, update_interval=None, logl_bounds=None, maxiter=None, maxcall=None,
                     save_bounds=True, **ignored):
        """One more batch of `nlive_new` particles between logl_bounds: yields dynesty's 9-tuples
        (worst, ustar, vstar, loglstar, nc, worst_it, boundidx, bounditer, eff)."""
        if not self.base:
            raise ValueError("sample_initial() has to run before a batch can be added")
        if logl_bounds is None:
            logl_bounds = (-np.inf, np.inf)
        logl_min, logl_max = float(logl_bounds[0]), float(logl_bounds[1])
        if not logl_max > logl_min:
            raise ValueError("logl_bounds must be increasing")
        nlive_new = int(nlive_new)
        ncall0 = 0
        if logl_min == -np.inf:
            S = self._static(nlive_new)
        else:
            alive = self._alive_at(logl_min)
            if len(alive) < 2:
                raise ValueError("no saved samples above logl_min = %r" % (logl_min,))
            s = self.saved
            seed = self._static(len(alive), live_points=(s["u"][alive], s["v"][alive], s["logl"][alive]),
                                loglstar=logl_min)
            seed.scale = float(s["scale"][alive].mean()) if "scale" in s else 1.0
            got_u, got_v, got_l, have = [], [], [], 0
            tries = 0
            while have < nlive_new:
                seed._fill_queue()
                ok = seed._ql > logl_min
                if ok.any():
                    got_u.append(seed._qU[ok]); got_v.append(seed._qV[ok]); got_l.append(seed._ql[ok])
                    have += int(ok.sum())
                tries += 1
                if tries > 10000:
                    raise RuntimeError("could not draw %d points above logl = %r" % (nlive_new, logl_min))
            ncall0 = int(seed.ncall)
            pts = [np.concatenate(a)[:nlive_new] for a in (got_u, got_v, got_l)]
            S = self._static(nlive_new, live_points=pts, loglstar=logl_min)
            S.scale = seed.scale
        self.sampler = S
        first = True
        for t in S.sample(dlogz=-1.0, maxiter=maxiter, maxcall=maxcall, logl_max=logl_max):
            nc = t[9] + (ncall0 if first else 0)
            first = False
            yield (t[0], t[1], t[2], t[3], nc, t[10], t[11], t[12], t[13])
        for t in S.add_live_points():
            yield (t[0], t[1], t[2], t[3], t[9], t[10], t[11], t[12], t[13])
        run = self._run_arrays(S, logl_min, self._next_id)
        if ncall0 and len(run["nc"]):
            run["nc"][0] += ncall0
        run["batch"] = np.full(len(run["logl"]), self.batch + 1, dtype=np.int64)
        self.new = run
        self.new_bounds = (logl_min, logl_max)
        self.new_nlive = nlive_new
        self.new_ncall = int(S.ncall) + ncall0

    def combine_runs(self):
        """Merge the last batch into the saved run and re-accumulate the evidence."""
        if self.new is None:
            raise ValueError("no new batch to combine")
        s, b = self.saved, self.new
        keys = ("u", "v", "logl", "birth", "id", "nc", "batch", "scale")
        run = {k: np.concatenate([s[k], b[k]]) for k in keys}
        self.saved = self._finish(run, self.batch + 1)
        self._next_id += self.new_nlive
        self.ncall += self.new_ncall
        self.eff = 100.0 * len(run["logl"]) / max(1, self.ncall)
        self.batch_bounds.append(self.new_bounds)
        self.batch_nlive.append(self.new_nlive)
        self.batch += 1
        self.new = None

    def add_batch(self, nlive=500, wt_function=None, wt_kwargs=None, maxiter=None, maxcall=None, logl_bounds=None,
                  **ignored):
        """dynesty's convenience wrapper: place, run and merge one batch."""
        if logl_bounds is None:
            logl_bounds = (wt_function or self.weight_function)(self.results, wt_kwargs)
        for _ in self.sample_batch(nlive_new=nlive, logl_bounds=logl_bounds, maxiter=maxiter, maxcall=maxcall):
            pass
        self.combine_runs()

    # ---- saved run ------------------------------------------------------------------------------------
    @property
    def results(self):
        s = self.saved
        if s is None:
            raise ValueError("no run has been saved yet")
        return Results(niter=len(s["logl"]), ncall=s["nc"], eff=self.eff, samples=s["v"], samples_u=s["u"],
                       samples_id=s["id"], samples_it=np.arange(len(s["logl"])), samples_n=s["n"],
                       samples_batch=s["batch"], samples_birth=s["birth"], logl=s["logl"], logvol=s["logvol"],
                       logwt=s["logwt"], logz=s["logz"], logzerr=np.sqrt(np.maximum(s["logzvar"], 0.0)),
                       information=s["h"], batch_nlive=np.array(self.batch_nlive),
                       batch_bounds=np.array(self.batch_bounds), scale=s["scale"])

    def posterior_weights(self):
        r = self.results
        w = np.exp(r.logwt - r.logz[-1])
        return w / w.sum()

    def summary(self):
        r = self.results
        w = self.posterior_weights()
        out = [r.logz[-1], r.logzerr[-1], float(r.niter), float(self.ncall), float(self.eff)]
        for j in range(self.ndim):
            x = r.samples[:, j]
            mean = float(np.sum(w * x))
            std = float(np.sqrt(max(0.0, np.sum(w * (x - mean) ** 2))))
            o = np.argsort(x)
            cw = np.cumsum(w[o])
            out += [mean, std] + [float(x[o][min(len(x) - 1, np.searchsorted(cw, p))]) for p in (0.16, 0.5, 0.84)]
        return np.array(out, dtype=np.float64)

    # ---- when to stop ---------------------------------------------------------------------------------
    @staticmethod
    def _strands(results):
        """Particle ids of the run, which samples belong to which, and the two groups that are resampled
        separately: strands started from the prior and strands added by batches."""
        ids, birth = results['samples_id'], results['samples_birth']
        uniq, inv = np.unique(ids, return_inverse=True)
        first_birth = np.full(len(uniq), np.inf)
        np.minimum.at(first_birth, inv, birth)
        from_prior = first_birth == -np.inf
        return inv, len(uniq), (np.nonzero(from_prior)[0], np.nonzero(~from_prior)[0])

    def _simulate(self, results, rng, strands=None):
        """One simulated realisation of the run: strands (particle ids) resampled with replacement --
        those started from the prior and those added by batches separately -- and the volume
        shrinkages drawn as Beta(n, 1).  Returns (indices into the run, logwt, final logz)."""
        birth, logl = results['samples_birth'], results['logl']
        inv, nstr, groups = strands if strands is not None else self._strands(results)
        count = np.zeros(nstr, dtype=np.int64)
        for grp in groups:
            if len(grp):
                count += np.bincount(rng.choice(grp, size=len(grp)), minlength=nstr)
        idx = np.repeat(np.arange(len(logl)), count[inv])             # stays sorted by logl
        if len(idx) < 2:
            idx = np.arange(len(logl))
        n = live_counts(logl[idx], birth[idx])
        ln_t = np.log(rng.uniform(size=len(idx))) / n
        logvol = np.cumsum(ln_t)
        prev_vol = np.concatenate([[0.0], logvol[:-1]])
        logdvol = math.log(0.5) + prev_vol + np.log1p(-np.exp(ln_t))
        ll = logl[idx]
        logwt = np.logaddexp(ll, np.concatenate([[_NEG], ll[:-1]])) + logdvol
        top = logwt.max()
        return idx, logwt, top + math.log(np.exp(logwt - top).sum())

    def stopping_function(self, results, args=None, rstate=None, M=None, return_vals=False):
        """dynesty's default stopping rule on simulated runs; True = stop."""
        args = args or {}
        pfrac = args.get('pfrac', 1.0)
        evid_thresh, post_thresh = args.get('evid_thresh', 0.1), args.get('post_thresh', 0.02)
        n_mc = int(args.get('n_mc', 128))
        if not 0.0 <= pfrac <= 1.0 or evid_thresh < 0 or post_thresh < 0 or n_mc <= 1:
            raise ValueError("bad stopping arguments")
        rng = rstate if rstate is not None else self.rng
        logp2_all = results['logwt'] - results['logz'][-1]
        kld, lnz = np.empty(n_mc), np.empty(n_mc)
        strands = self._strands(results)
        for k in range(n_mc):
            idx, logwt, logz = self._simulate(results, rng, strands)
            logp1 = logwt - logz
            kld[k] = np.sum(np.exp(logp1) * (logp1 - logp2_all[idx]))
            lnz[k] = logz
        stop_evid = float(np.std(lnz)) / evid_thresh if pfrac < 1.0 else 0.0
        stop_post = float(np.std(kld) / np.mean(kld)) / post_thresh if pfrac > 0.0 else 0.0
        stop_val = pfrac * stop_post + (1.0 - pfrac) * stop_evid
        stop = bool(stop_val <= 1.0)
        if return_vals:
            return stop, (stop_post, stop_evid, stop_val)
        return stop

    def run_nested(self, nlive_init=500, dlogz_init=0.01, nlive_batch=500, maxiter=None, maxcall=None, maxbatch=None,
                   wt_kwargs=None, stop_kwargs=None, **ignored):
        """Baseline + batches until the stopping rule fires (dynesty's run_nested, defaults only)."""
        for _ in self.sample_initial(nlive=nlive_init, dlogz=dlogz_init, maxiter=maxiter, maxcall=maxcall):
            pass
        maxbatch = np.inf if maxbatch is None else maxbatch
        while self.batch < maxbatch and (maxcall is None or self.ncall < maxcall):
            res = self.results
            if self.stopping_function(res, stop_kwargs):
                break
            self.add_batch(nlive=nlive_batch, wt_kwargs=wt_kwargs, maxiter=maxiter, maxcall=maxcall)
