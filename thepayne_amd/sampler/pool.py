"""A ``pool``-shaped adapter: real dynesty on top of the batched GPU likelihood.

The reference hands dynesty a scalar ``lnprobfn`` and, optionally, a ``pool`` whose ``map`` spreads
``queue_size`` proposals over worker processes (Payne/fitting/fitstar.py:309-321: ``pool=`` is commented
out there, ``dynesty.NestedSampler(lnprobfn, priortrans, ndim, logl_args=[likeobj, priorobj], ...)``).
``BatchPool`` is that pool for this build: ``map(fn, items)`` runs the items side by side and every
likelihood value they ask for at the same time is computed in ONE ``lnprob_batch`` call on the GPU:

    pool = BatchPool(likeobj, priorobj)
    dynesty.NestedSampler(pool.lnprob, pool.prior_transform, ndim, pool=pool, queue_size=pool.size,
                          use_pool={'prior_transform': True, 'loglikelihood': True, 'propose_point': False,
                                    'update_bound': False}, ...)

How: ``fn`` is whatever the sampler maps over the queue -- the likelihood itself (initial live points, 'unif'
proposals) or a point-evolving function that calls the likelihood several times ('rwalk', 'slice').  A direct map
of ``pool.lnprob`` / ``pool.prior_transform`` is one batched call.  Anything else runs one thread per item; inside
them ``pool.lnprob(v)`` parks the caller until every thread still running has asked for its next value, then the
last one to arrive evaluates the whole set as one batch and hands the values out.  The threads are plain Python
threads used as coroutines (the GIL serialises them; all arithmetic is on the GPU).
"""
import threading

import numpy as np

__all__ = ["BatchPool"]


class BatchPool(object):
    def __init__(self, likeobj=None, priorobj=None, size=None, lnprob_batch=None, priortrans_batch=None):
        """``likeobj`` / ``priorobj``: the fit's likelihood and prior objects (their ``*_batch`` methods are used);
        or pass the two batch callables directly (``lnprob_batch(theta[B, ndim]) -> [B]``)."""
        if lnprob_batch is None:
            from ..fitting.fitstar import lnprob_batch as _lpb
            lnprob_batch = lambda th: _lpb(th, likeobj, priorobj)      # noqa: E731
        if priortrans_batch is None and priorobj is not None:
            priortrans_batch = priorobj.priortrans_batch
        self._lnprob_batch = lnprob_batch
        self._priortrans_batch = priortrans_batch
        self.size = int(size or getattr(likeobj, "b_max", 0) or 512)
        self.ncall_batches = 0               # likelihood batches evaluated (diagnostic)
        self.ncall_points = 0
        self._cv = threading.Condition()
        self._tls = threading.local()
        self._active = 0
        self._pending = []                   # [(slot, v)]
        self._results = {}

    # -- the callables handed to the sampler ---------------------------------------
    def lnprob(self, v, *args, **kwargs):
        v = np.asarray(v, dtype=np.float64)
        if not getattr(self._tls, "worker", False):
            return float(self._eval(v[None, :])[0])
        return self._rendezvous(v)

    def prior_transform(self, u, *args, **kwargs):
        return np.asarray(self._priortrans_batch(np.asarray(u, dtype=np.float64)[None, :]))[0]

    # -- pool interface --------------------------------------------------------------
    def map(self, fn, iterable):
        items = list(iterable)
        if not items:
            return []
        target = getattr(fn, "func", fn)                     # dynesty wraps its callables (.func, .args, .kwargs)
        if target == self.lnprob:
            return [float(x) for x in self._eval(np.asarray(items, dtype=np.float64))]
        if target == self.prior_transform:
            return list(np.asarray(self._priortrans_batch(np.asarray(items, dtype=np.float64))))
        return self._run_side_by_side(fn, items)

    def close(self):
        pass

    def join(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    # -- internals -----------------------------------------------------------------------
    def _eval(self, theta):
        self.ncall_batches += 1
        self.ncall_points += len(theta)
        out = []
        for s in range(0, len(theta), self.size):            # the engine's workspaces hold `size` rows
            out.append(np.asarray(self._lnprob_batch(theta[s:s + self.size]), dtype=np.float64))
        return np.concatenate(out)

    def _flush_locked(self):
        """Evaluate everything parked (caller holds the lock) and wake the waiters."""
        slots = [s for s, _ in self._pending]
        theta = np.asarray([v for _, v in self._pending], dtype=np.float64)
        self._pending = []
        try:
            vals = self._eval(theta)
            for s, x in zip(slots, vals):
                self._results[s] = ("ok", float(x))
        except BaseException as e:                           # every waiter sees the failure
            for s in slots:
                self._results[s] = ("err", e)
        self._cv.notify_all()

    def _rendezvous(self, v):
        slot = self._tls.slot
        with self._cv:
            self._pending.append((slot, v))
            if len(self._pending) == self._active:           # everybody still running is parked: evaluate
                self._flush_locked()
            while slot not in self._results:
                self._cv.wait()
            kind, val = self._results.pop(slot)
        if kind == "err":
            raise val
        return val

    def _run_side_by_side(self, fn, items):
        n = len(items)
        out, errs = [None] * n, [None] * n
        self._active = n
        self._pending, self._results = [], {}

        def work(i):
            self._tls.worker, self._tls.slot = True, i
            try:
                out[i] = fn(items[i])
            except BaseException as e:
                errs[i] = e
            finally:
                self._tls.worker = False
                with self._cv:
                    self._active -= 1
                    if self._pending and len(self._pending) == self._active:   # the others were waiting for this one
                        self._flush_locked()

        threads = [threading.Thread(target=work, args=(i,), daemon=True) for i in range(n)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for e in errs:
            if e is not None:
                raise e
        return out
