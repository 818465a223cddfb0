"""Device-side sampler step: the prior transform, ln-priors and random-walk proposals of a fit
evaluated on the GPU (payne_sampler_* in include/payne_hip.h), so that a batched nested
sampler is not bound by 0.4 ms of scipy ppf calls and a host round trip per chain step.

``DeviceProposer(likeobj, priorobj)`` translates the reference-shaped ``prior`` object
(thepayne_amd.fitting.prior) into the C descriptor: the same per-name priority of prior kinds as
``priortrans_spec`` / ``priortrans_phot`` (Payne/fitting/prior.py:151-178, :236-270), the blaze
coefficient boxes (:180-191) and the additive 'gaussian' / 'uniform' priors (:379-465).
"""
import ctypes as C

import numpy as np

from .. import _lib

_SPEC = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Inst_R', 'CarbonScale']
_ATM = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]']
_ISO = ['log(A)', 'log(R)', 'Av', 'Rv', 'Dist']
_KIND = {'uniform': _lib.PRIOR_UNIFORM, 'gaussian': _lib.PRIOR_GAUSSIAN, 'tgaussian': _lib.PRIOR_TGAUSSIAN,
         'exp': _lib.PRIOR_EXP, 'texp': _lib.PRIOR_TEXP, 'loguniform': _lib.PRIOR_LOGUNIFORM}


class DeviceProposer(object):
    def __init__(self, likeobj, priorobj, k_max=None, engine=None):
        self.like = likeobj
        self.prior = priorobj
        self.eng = engine if engine is not None else likeobj.GM.engine
        self._held = False
        self.torch = self.eng.torch
        self.lib = self.eng.lib
        self.ndim = priorobj.ndim
        self.k_max = int(k_max or self.eng.b_max)
        if self.ndim > _lib.PAYNE_MAX_DIM:
            raise ValueError("more than %d sampled dimensions" % _lib.PAYNE_MAX_DIM)
        ncols, idx, fixed = likeobj._columns()
        col_of = {likeobj.fitpars_i[j]: c for c, j in idx}
        d = _lib.SamplerDesc()
        d.ndim = self.ndim
        for j, name in enumerate(priorobj.fitpars_i):
            kind, p = self._kind(name)
            dim = d.dims[j]
            dim.kind = kind
            dim.theta_col = col_of.get(name, -1)
            for q, val in enumerate(p):
                dim.p[q] = float(val)
            add = self._additional(name)
            if 'gaussian' in add:
                dim.has_gauss, dim.g_mu, dim.g_sigma = 1, float(add['gaussian'][0]), float(add['gaussian'][1])
            if 'uniform' in add:
                dim.has_box, dim.box_lo, dim.box_hi = 1, float(add['uniform'][0]), float(add['uniform'][1])
        self._adv_tables = None
        self._advanced(d, priorobj, likeobj)
        if len(fixed) > _lib.PAYNE_MAX_FIXED:
            raise ValueError("too many fixed parameters")
        d.n_fixed = len(fixed)
        for q, (c, val) in enumerate(fixed):
            d.fixed_col[q] = c
            d.fixed_val[q] = val
        self._handle = C.c_void_p()
        rc = self.lib.payne_sampler_create(self.eng._ctx, C.byref(d), self.k_max, C.byref(self._handle))
        if rc != 0:
            self.eng._err(rc, "payne_sampler_create")
        if engine is None and hasattr(self.eng, "hold"):     # (the fit's own context: kept alive, and out of the idle pool, while this walks on it)
            self.eng.hold()
            self._held = True
        dev = self.eng.device
        f64, i32 = self.torch.float64, self.torch.int32
        self._u = self.torch.empty((self.k_max, self.ndim), dtype=f64, device=dev)
        self._v = self.torch.empty((self.k_max, self.ndim), dtype=f64, device=dev)
        self._lp = self.torch.empty(self.k_max, dtype=f64, device=dev)
        self._nacc = self.torch.empty(self.k_max, dtype=i32, device=dev)
        self._ncall = self.torch.empty(self.k_max, dtype=i32, device=dev)
        self._pack_h = None                  # pinned staging for rwalk, made on first use
        self._qstats = np.zeros(4, dtype=np.int64)

    # -- priors on derived quantities (prior.py:286-336, :449-451) ----------------------
    def _advanced(self, d, P, L):
        """IMF / VROT terms, the GAL distance transform and the derived 'Parallax' prior as the library's
        payne_adv_priors; the cases the reference itself cannot evaluate stay on the host path (which raises as the
        reference does)."""
        a = d.adv
        a.dim_logg = a.dim_logr = a.dim_vrot = a.plx_dim = -1
        a.val_logg = a.val_logr = a.val_vrot = float("nan")
        names = list(P.fitpars_i)
        fixed = dict(P.fixedpars)

        def locate(name):
            if name in names:
                return names.index(name), float("nan")
            v = fixed.get(name, float("nan"))
            return -1, (float(v) if np.ndim(v) == 0 else float("nan"))

        if getattr(P, 'imf_bool', False) or getattr(P, 'vrot_bool', False):
            a.dim_logg, a.val_logg = locate('log(g)')
            a.dim_logr, a.val_logr = locate('log(R)')
            a.dim_vrot, a.val_vrot = locate('Vrot')
        if getattr(P, 'imf_bool', False):
            if 'log(R)' not in names and 'log(R)' not in fixed:
                raise NotImplementedError("IMF prior without log(R): the reference raises KeyError; host path")
            a.imf = 1
        if getattr(P, 'vrot_bool', False):
            if a.dim_vrot < 0 and not np.isfinite(a.val_vrot):
                raise NotImplementedError("VROT prior without Vrot: host path")
            a.vrot = 1
            a.vrot_mass_one = 1 if ('log(A)' in names or 'log(A)' in fixed or
                                    ('log(R)' not in names and 'log(R)' not in fixed)) else 0
        if getattr(P, 'gal_bool', False) and 'Dist' in names and P.phot_bool:          # prior.py:231-234
            j = names.index('Dist')
            d.dims[j].kind = _lib.PRIOR_TABLE
            d.dims[j].p[0] = 1000.0
            self._adv_tables = (np.ascontiguousarray(P.AP._cdf, dtype=np.float64),
                                np.ascontiguousarray(P.AP.distarr, dtype=np.float64))
            a.tab_cdf, a.tab_val = self._adv_tables[0].ctypes.data, self._adv_tables[1].ctypes.data
            a.tab_n = len(self._adv_tables[0])
        plx = P.additionalpriors.get('Parallax', {})
        plx = {k: v for k, v in plx.items() if k in ('gaussian', 'uniform')}
        if plx and P.phot_bool and 'Dist' in names:
            a.plx_dim = names.index('Dist')
            if 'gaussian' in plx:
                a.plx_has_gauss, a.plx_mu, a.plx_sigma = 1, float(plx['gaussian'][0]), float(plx['gaussian'][1])
            if 'uniform' in plx:
                a.plx_has_box, a.plx_lo, a.plx_hi = 1, float(plx['uniform'][0]), float(plx['uniform'][1])

    # -- prior description -----------------------------------------------------------
    def _kind(self, name):
        P = self.prior
        if 'pc' in name:                                            # prior.py:180-191
            if name == 'pc_0':
                return _lib.PRIOR_UNIFORM, (0.75, 1.25)
            mu, sig = P.polycoefarr[int(name.split('_')[-1])][:2]
            return _lib.PRIOR_UNIFORM, (mu - 5.0 * sig, mu + 5.0 * sig)
        if name in _ISO and P.phot_bool and not (P.spec_bool and name in _SPEC):
            order = ('uniform', 'gaussian', 'exp', 'tgaussian', 'texp', 'loguniform')
        else:
            order = ('uniform', 'gaussian', 'tgaussian', 'exp', 'texp')
        for kind in order:
            if name in P.priordict[kind]:
                return _KIND[kind], tuple(P.priordict[kind][name])
        lo, hi = P.defaultpars[name]
        return _lib.PRIOR_UNIFORM, (lo, hi)

    def _additional(self, name):
        P = self.prior
        add = P.additionalpriors.get(name, {})
        add = {k: v for k, v in add.items() if k in ('gaussian', 'uniform')}
        if not add:
            return {}
        in_spec = P.spec_bool and name in _SPEC
        in_phot = P.phot_bool and (name in ('log(R)', 'Dist', 'log(A)', 'Av') or (not P.spec_bool and name in _ATM))
        return add if (in_spec or in_phot) else {}

    # -- calls -------------------------------------------------------------------------
    def _stream(self):
        return self.eng._stream()

    def _up(self, a, dst):
        t = self.torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64))
        dst[:len(a)].copy_(t.reshape(dst[:len(a)].shape))
        return len(a)

    def prior_transform(self, U):
        """U[K, ndim] -> V[K, ndim] (numpy)."""
        out = []
        U = np.atleast_2d(U)
        for s in range(0, len(U), self.k_max):
            K = self._up(U[s:s + self.k_max], self._u)
            rc = self.lib.payne_prior_transform_batch(self._handle, self._u.data_ptr(), K, self._v.data_ptr(), self._stream())
            if rc != 0:
                self.eng._err(rc, "payne_prior_transform_batch")
            out.append(self._v[:K].cpu().numpy())
        return np.concatenate(out)

    def lnprob_u(self, U):
        """U[K, ndim] -> (V[K, ndim], lnprob[K]) with lnprob = lnprior + lnlike."""
        Vs, Ls = [], []
        U = np.atleast_2d(U)
        for s in range(0, len(U), self.k_max):
            K = self._up(U[s:s + self.k_max], self._u)
            rc = self.lib.payne_lnprob_u_batch(self._handle, self._u.data_ptr(), K, self._v.data_ptr(), self._lp.data_ptr(),
                                               self._stream())
            if rc != 0:
                self.eng._err(rc, "payne_lnprob_u_batch")
            Vs.append(self._v[:K].cpu().numpy())
            Ls.append(self._lp[:K].cpu().numpy())
        return np.concatenate(Vs), np.concatenate(Ls)

    def rwalk(self, U, V, lnprob, axes, scale, loglstar, walks, seed, ell=None):
        """K lock-step random-walk chains of `walks` steps under lnprob > loglstar.  `axes` is one
        [ndim, ndim] matrix, or [n_ell, ndim, ndim] with `ell[K]` naming each chain's ellipsoid.
        Returns (U, V, lnprob, nacc, ncall) as numpy arrays.  One packed pinned transfer each way."""
        self.rwalk_begin(U, V, lnprob, axes, scale, loglstar, walks, seed, ell=ell)
        for w in range(int(walks) + 1):
            self.rwalk_step(w)
        return self.rwalk_finish()

    def rwalk_queue(self, live_u, live_v, live_logl, K, axes_unit, ctr, ainv, scale, loglstar, walks, seed, qbuf, between=None):
        """One whole queue of random-walk proposals in ONE native call (payne_ns_rwalk_queue): start points, ellipsoid
        assignment, transfers, the walk and the selection of the chains that moved.  ``qbuf`` = (qU[K, nd], qV[K, nd],
        ql[K], qnc[K] int32) host arrays the queue is written to.  Returns (nq, accepted, calls, redrawn, idle_calls).
        ``between``: a callable run on the host while the GPU walks (payne_ns_rwalk_queue_begin .. _end)."""
        if between is None:
            if K > self.k_max:
                raise ValueError("K > k_max")
            axp, n_ell, cp, ap, keep = self._queue_bound(axes_unit, ctr, ainv)
            qU, qV, ql, qnc = qbuf
            nq = C.c_int(0)
            stats = self._qstats
            rc = self.lib.payne_ns_rwalk_queue(self._handle, live_u.ctypes.data, live_v.ctypes.data, live_logl.ctypes.data,
                                               len(live_logl), int(K), axp, n_ell, cp, ap, float(scale), float(loglstar),
                                               int(walks), int(seed) & 0xFFFFFFFFFFFFFFFF, qU.ctypes.data, qV.ctypes.data,
                                               ql.ctypes.data, qnc.ctypes.data, C.byref(nq), stats.ctypes.data, self._stream())
            if rc != 0:
                self.eng._err(rc, "payne_ns_rwalk_queue")
            return nq.value, int(stats[0]), int(stats[1]), int(stats[2]), int(stats[3])
        self.rwalk_queue_begin(live_u, live_v, live_logl, K, axes_unit, ctr, ainv, scale, loglstar, walks, seed)
        try:
            between()
        finally:                                  # (the queue in flight is always collected, whatever the host work did)
            out = self.rwalk_queue_end(qbuf)
        return out

    def _queue_bound(self, axes_unit, ctr, ainv):
        """The bound's arrays as the native call takes them; remembered while the caller hands the same objects (a bound lives for
        several queues, and three ascontiguousarray + six attribute lookups are 10 us between two queues)."""
        key = (id(axes_unit), id(ctr), id(ainv))
        if getattr(self, "_qb_key", None) == key:
            return self._qb
        ax = np.ascontiguousarray(axes_unit, dtype=np.float64)
        n_ell = 1 if ax.ndim == 2 else ax.shape[0]
        cp = ap = None
        keep = (ax, axes_unit, ctr, ainv)
        if n_ell > 1:
            c2 = np.ascontiguousarray(ctr, dtype=np.float64)
            a2 = np.ascontiguousarray(ainv, dtype=np.float64)
            cp, ap, keep = c2.ctypes.data, a2.ctypes.data, keep + (c2, a2)
        self._qb_key, self._qb = key, (ax.ctypes.data, n_ell, cp, ap, keep)     # (keep: the objects stay alive, so do their ids)
        return self._qb

    def rwalk_queue_begin(self, live_u, live_v, live_logl, K, axes_unit, ctr, ainv, scale, loglstar, walks, seed):
        """payne_ns_rwalk_queue_begin: everything of one queue enqueued on the stream (nothing of the arguments is read after the
        call returns); the host is free until rwalk_queue_end."""
        if K > self.k_max:
            raise ValueError("K > k_max")
        axp, n_ell, cp, ap, keep = self._queue_bound(axes_unit, ctr, ainv)
        rc = self.lib.payne_ns_rwalk_queue_begin(self._handle, live_u.ctypes.data, live_v.ctypes.data, live_logl.ctypes.data,
                                                 len(live_logl), int(K), axp, n_ell, cp, ap, float(scale),
                                                 float(loglstar), int(walks), int(seed) & 0xFFFFFFFFFFFFFFFF, self._stream())
        if rc != 0:
            self.eng._err(rc, "payne_ns_rwalk_queue_begin")

    def rwalk_queue_end(self, qbuf):
        """payne_ns_rwalk_queue_end: waits for the stream, writes the queue to ``qbuf``; returns (nq, accepted, calls, redrawn,
        idle_calls)."""
        qU, qV, ql, qnc = qbuf
        nq = C.c_int(0)
        stats = self._qstats
        rc = self.lib.payne_ns_rwalk_queue_end(self._handle, qU.ctypes.data, qV.ctypes.data, ql.ctypes.data, qnc.ctypes.data,
                                               C.byref(nq), stats.ctypes.data)
        if rc != 0:
            self.eng._err(rc, "payne_ns_rwalk_queue_end")
        return nq.value, int(stats[0]), int(stats[1]), int(stats[2]), int(stats[3])

    def rwalk_queue_turn(self, qbuf, live_u, live_v, live_logl, K, axes_unit, ctr, ainv, scale, loglstar, walks, seed):
        """payne_ns_rwalk_queue_turn: collect the queue in flight into ``qbuf``, adapt the scale, predict the state its consumption
        will leave and launch the next queue from there -- one native call between two queues.  Returns (nq, accepted, calls,
        redrawn, idle_calls, scale, loglstar_after, n_dead)."""
        if K > self.k_max:
            raise ValueError("K > k_max")
        axp, n_ell, cp, ap, keep = self._queue_bound(axes_unit, ctr, ainv)
        qU, qV, ql, qnc = qbuf
        nq, m = C.c_int(0), C.c_int(0)
        sc, ls = C.c_double(scale), C.c_double(loglstar)
        stats = self._qstats
        rc = self.lib.payne_ns_rwalk_queue_turn(self._handle, qU.ctypes.data, qV.ctypes.data, ql.ctypes.data, qnc.ctypes.data,
                                                C.byref(nq), stats.ctypes.data, live_u.ctypes.data, live_v.ctypes.data,
                                                live_logl.ctypes.data, len(live_logl), int(K), axp, n_ell, cp, ap, C.byref(sc),
                                                C.byref(ls), int(walks), int(seed) & 0xFFFFFFFFFFFFFFFF, C.byref(m))
        if rc != 0:
            self.eng._err(rc, "payne_ns_rwalk_queue_turn")
        return nq.value, int(stats[0]), int(stats[1]), int(stats[2]), int(stats[3]), sc.value, ls.value, m.value

    # ---- the queue's turn on the device (payne_ns_queue_dev_*): the live set lives there, queues follow each other without the host
    def queue_dev_init(self, live_u, live_v, live_logl, scale, loglstar):
        """Upload the live set and the scale / threshold the first queue starts from.  Queues still in flight (a sampling loop
        that was abandoned without being finalised: its generator still referenced somewhere) are collected and dropped first."""
        while getattr(self, "_dq_out", 0) > 0:
            K, nd = self.k_max, self.ndim
            self.queue_dev_collect((np.empty((K, nd)), np.empty((K, nd)), np.empty(K), np.empty(K, dtype=np.int32)))
        u = np.ascontiguousarray(live_u, dtype=np.float64)
        v = np.ascontiguousarray(live_v, dtype=np.float64)
        l = np.ascontiguousarray(np.where(np.isnan(live_logl), -np.inf, live_logl), dtype=np.float64)
        rc = self.lib.payne_ns_queue_dev_init(self._handle, u.ctypes.data, v.ctypes.data, l.ctypes.data, len(l), float(scale), float(loglstar))
        if rc != 0:
            self.eng._err(rc, "payne_ns_queue_dev_init")
        self._qb_dev = None
        # (whose queues are in flight from here on: a sampler that finds another epoch when it comes to drain "its" queues -- an
        #  abandoned loop finalised after another sampler started on this proposer -- has none left: they were dropped above)
        self._dq_epoch = getattr(self, "_dq_epoch", 0) + 1
        return self._dq_epoch

    def queue_dev_launch(self, K, axes_unit, ctr, ainv, walks, seed, merge=True):
        """Enqueue one queue behind whatever is in flight: [the bound, when it is not the one already on the device] + the turn
        (merge: the queue before this one into the live set) + the walk + its results' transfer."""
        if K > self.k_max:
            raise ValueError("K > k_max")
        key = (id(axes_unit), id(ctr), id(ainv))
        if getattr(self, "_qb_dev", None) == key:
            axp, n_ell, cp, ap = None, 0, None, None                     # (the bound on the device is this one)
        else:
            axp, n_ell, cp, ap, keep = self._queue_bound(axes_unit, ctr, ainv)
        rc = self.lib.payne_ns_queue_dev_launch(self._handle, int(K), axp, n_ell, cp, ap, int(walks), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                                1 if merge else 0, self._stream())
        if rc != 0:
            self.eng._err(rc, "payne_ns_queue_dev_launch")
        self._qb_dev = key
        self._dq_out = getattr(self, "_dq_out", 0) + 1

    def queue_dev_collect(self, qbuf):
        """The oldest queue in flight, as rwalk_queue_end returns it, + the scale and threshold it ran under."""
        qU, qV, ql, qnc = qbuf
        nq = C.c_int(0)
        stats = self._qstats
        if getattr(self, "_dyn_used", None) is None:
            self._dyn_used = np.zeros(2)
        rc = self.lib.payne_ns_queue_dev_collect(self._handle, qU.ctypes.data, qV.ctypes.data, ql.ctypes.data, qnc.ctypes.data,
                                                 C.byref(nq), stats.ctypes.data, self._dyn_used.ctypes.data)
        if rc != 0:
            self.eng._err(rc, "payne_ns_queue_dev_collect")
        self._dq_out = getattr(self, "_dq_out", 1) - 1
        return nq.value, int(stats[0]), int(stats[1]), int(stats[2]), int(stats[3]), float(self._dyn_used[0]), float(self._dyn_used[1])

    # the same in three parts (MultiPopProposer interleaves the steps of several populations)
    def rwalk_begin(self, U, V, lnprob, axes, scale, loglstar, walks, seed, stream=None, ell=None):
        K, nd = len(U), self.ndim
        if K > self.k_max:
            raise ValueError("K > k_max")
        t = self.torch
        if self._pack_h is None:
            n = self.k_max * (2 * nd + 1)
            self._pack_h = t.empty(n, dtype=t.float64).pin_memory()
            self._pack_d = t.empty(n, dtype=t.float64, device=self.eng.device)
            self._ipack_h = t.empty(2 * self.k_max, dtype=t.int32).pin_memory()
            self._ipack_d = t.empty(2 * self.k_max, dtype=t.int32, device=self.eng.device)
        self._run_stream = stream if stream is not None else t.cuda.current_stream(self.eng.device)
        n = K * (2 * nd + 1)
        h = self._pack_h.numpy()
        h[:K * nd] = np.asarray(U, dtype=np.float64).reshape(-1)
        h[K * nd:2 * K * nd] = np.asarray(V, dtype=np.float64).reshape(-1)
        h[2 * K * nd:n] = lnprob
        d = self._pack_d
        with t.cuda.stream(self._run_stream):
            d[:n].copy_(self._pack_h[:n], non_blocking=True)
        pu, pv, pl = d.data_ptr(), d.data_ptr() + 8 * K * nd, d.data_ptr() + 16 * K * nd
        ax = np.ascontiguousarray(axes, dtype=np.float64)
        n_ell, ell_p = 1, None
        if ax.ndim == 3:
            n_ell = ax.shape[0]
            if n_ell > 1:
                self._ell_h = np.ascontiguousarray(ell, dtype=np.int32)
                if self._ell_h.shape != (K,):
                    raise ValueError("ell must name one ellipsoid per chain")
                ell_p = self._ell_h.ctypes.data
        rc = self.lib.payne_rwalk_begin_ell(self._handle, pu, pv, pl, K, ax.ctypes.data, n_ell, ell_p,
                                            float(scale), float(loglstar), int(walks), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                            self._ipack_d.data_ptr(), self._ipack_d.data_ptr() + 4 * K,
                                            C.c_void_p(self._run_stream.cuda_stream))
        if rc != 0:
            self.eng._err(rc, "payne_rwalk_begin_ell")
        self._run_K = K

    def rwalk_step(self, w):
        rc = self.lib.payne_rwalk_step(self._handle, int(w))
        if rc != 0:
            self.eng._err(rc, "payne_rwalk_step")

    def rwalk_finish(self):
        K, nd, t = self._run_K, self.ndim, self.torch
        n = K * (2 * nd + 1)
        with t.cuda.stream(self._run_stream):
            self._pack_h[:n].copy_(self._pack_d[:n], non_blocking=True)
            self._ipack_h[:2 * K].copy_(self._ipack_d[:2 * K], non_blocking=True)
        self._run_stream.synchronize()
        h = self._pack_h.numpy()
        ih = self._ipack_h.numpy()
        return (h[:K * nd].reshape(K, nd).copy(), h[K * nd:2 * K * nd].reshape(K, nd).copy(), h[2 * K * nd:n].copy(),
                ih[:K].astype(np.int64), ih[K:2 * K].astype(np.int64))

    def step_counters(self):
        """(chain steps run at the likelihood-only post kernel's tail, chain steps launched on their own) so far."""
        out = (C.c_longlong * 2)()
        self.lib.payne_sampler_counters(self._handle, out)
        return int(out[0]), int(out[1])

    def close(self):
        if self._handle.value:
            if self.eng.is_open():
                self.torch.cuda.synchronize(self.eng.device)
            self.lib.payne_sampler_destroy(self._handle)
            self._handle = C.c_void_p()
        if getattr(self, "_held", False):
            self._held = False
            self.eng.drop()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiPopProposer(object):
    """Several chain populations in flight: ``n_pop`` DeviceProposers, each on its own context and HIP stream,
    their walk steps interleaved from this one host thread (payne_rwalk_begin / payne_rwalk_step).  One
    likelihood batch is three dependent launches with a fixed cost each; a second, independent batch fills the
    idle time (13.6 M against 10.6 M evaluations/s at the 4096-pixel / 512-candidate size).  ``rwalk`` takes up to
    ``n_pop * k_max`` chains and splits them into contiguous blocks; everything else goes to population 0."""

    def __init__(self, likeobj, priorobj, k_max=None, n_pop=2):
        if likeobj.fixedpars and any(np.ndim(v) > 0 for v in likeobj.fixedpars.values()):
            raise NotImplementedError("an LSF vector lives in one context; use a single population")
        first = DeviceProposer(likeobj, priorobj, k_max=k_max)
        self.pops = [first] + [DeviceProposer(likeobj, priorobj, k_max=k_max, engine=likeobj.GM.new_engine())
                               for _ in range(1, int(n_pop))]
        self.k_max = first.k_max * len(self.pops)
        self.ndim = first.ndim
        t = first.torch
        self._streams = [t.cuda.Stream(device=p.eng.device) for p in self.pops]

    def prior_transform(self, U):
        return self.pops[0].prior_transform(U)

    def lnprob_u(self, U):
        return self.pops[0].lnprob_u(U)

    def rwalk(self, U, V, lnprob, axes, scale, loglstar, walks, seed, ell=None):
        K, n = len(U), len(self.pops)
        if K > self.k_max:
            raise ValueError("K > n_pop * k_max")
        per = (K + n - 1) // n
        live = []
        for i, p in enumerate(self.pops):
            lo, hi = i * per, min(K, (i + 1) * per)
            if hi > lo:
                p.rwalk_begin(U[lo:hi], V[lo:hi], lnprob[lo:hi], axes, scale, loglstar, walks,
                              (int(seed) + 0x9E3779B9 * i) & 0xFFFFFFFFFFFFFFFF, stream=self._streams[i],
                              ell=None if ell is None else ell[lo:hi])
                live.append(p)
        for w in range(int(walks) + 1):                 # one step of every population, round robin
            for p in live:
                p.rwalk_step(w)
        parts = [p.rwalk_finish() for p in live]
        return tuple(np.concatenate([q[j] for q in parts]) for j in range(5))

    def close(self):
        for p in self.pops[1:]:
            p.close()
            p.eng.close()
        self.pops[0].close()
