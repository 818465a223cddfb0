// ns_core.hpp -- host-side bookkeeping of the static nested sampler (no GPU code).
//
// The reference hands its three callables to dynesty and consumes one 15-tuple per
// iteration (Payne/fitting/fitstar.py:309-338).  With proposals generated K at a time on the
// GPU (payne_rwalk_batch) the per-iteration work that is left -- find the worst live point,
// update ln Z / H by the trapezoid rule, take the next queued proposal that beats the
// threshold -- costs ~20 us per iteration in Python, several times what the GPU needs for the
// 25 likelihood calls behind it.  payne_ns_consume is that loop in C++: it walks a queue of
// proposals and emits one record per dead point until the queue is spent or a stop condition
// holds.  The caller owns every array (the live set is updated in place).
//
// Evidence arithmetic (Skilling 2006; the same as thepayne_amd/sampler/nested.py):
//   ln X_i = -i ln((n+1)/n);  ln w_i = ln(0.5 (X_{i-1} - X_i)) + logaddexp(L_i, L_{i-1});
//   Z += w_i;  H from the running sum of L w;  var(ln Z) += dH * dlnX.
#pragma once
#include <math.h>

#include "../../include/payne_hip.h"

namespace payne_ns {

inline double logaddexp(double a, double b) {
  if (a == b) return a + 0.6931471805599453;
  const double m = a > b ? a : b, d = a > b ? b - a : a - b;
  return (d == d) ? m + log1p(exp(d)) : (a + b);   // NaN only if an input is NaN (or inf - inf)
}

}  // namespace payne_ns

extern "C" int payne_ns_consume(payne_ns_state* s, double* live_u, double* live_v, double* live_logl, int* live_it,
                                const double* qu, const double* qv, const double* ql, const int* qnc, int nq,
                                double dlogz, long long max_emit, double logl_max, payne_ns_dead* out, int cap,
                                int* consumed, int* stop) {
  using payne_ns::logaddexp;
  if (!s || !live_u || !live_v || !live_logl || !live_it || !consumed || !stop || nq < 0 || cap < 0) return PAYNE_E_INVALID;
  if (nq > 0 && (!qu || !qv || !ql || !qnc)) return PAYNE_E_INVALID;
  if (cap > 0 && (!out || !out->worst || !out->u || !out->v || !out->logl || !out->logvol || !out->logwt || !out->logz ||
                  !out->logzvar || !out->h || !out->nc || !out->worst_it || !out->delta_logz))
    return PAYNE_E_INVALID;
  const int n = s->nlive, nd = s->ndim;
  if (n <= 0 || nd <= 0) return PAYNE_E_INVALID;
  const double dlv = log((n + 1.0) / n);
  const double logdfac = log(0.5 * expm1(dlv));          // ln(0.5 (X_{i-1} - X_i)) - ln X_i
  int qpos = 0, emitted = 0;
  *stop = PAYNE_NS_QUEUE_EMPTY;
  // the maximum only ever grows (the point that leaves is the minimum), so it is tracked; the
  // minimum is one pass over the live set per iteration
  double lmax = live_logl[0];
  for (int i = 1; i < n; ++i) lmax = live_logl[i] > lmax ? live_logl[i] : lmax;
  while (true) {
    int worst = 0;
    double lmin = live_logl[0];
    for (int i = 1; i < n; ++i) {
      const double l = live_logl[i];
      if (l < lmin) { lmin = l; worst = i; }
    }
    const double delta = (s->logz > -1e299) ? logaddexp(s->logz, lmax + s->logvol) - s->logz : INFINITY;
    if (delta < dlogz) { *stop = PAYNE_NS_CONVERGED; break; }
    if (emitted >= max_emit) { *stop = PAYNE_NS_LIMIT; break; }
    if (lmin >= logl_max) { *stop = PAYNE_NS_LOGL_MAX; break; }
    if (emitted >= cap) { *stop = PAYNE_NS_LIMIT; break; }
    // the next queued proposal that beats the threshold (a proposal made under an older, lower
    // threshold is kept iff it still beats the current one)
    long long nc = s->pending_nc;
    while (qpos < nq && !(ql[qpos] > lmin)) { nc += qnc[qpos]; ++qpos; }
    if (qpos >= nq) { s->pending_nc = nc; *stop = PAYNE_NS_QUEUE_EMPTY; break; }
    nc += qnc[qpos];
    s->pending_nc = 0;
    // evidence update for the point that dies
    const double logvol = s->logvol - dlv;
    const double logdvol = logdfac + logvol;
    const double logwt = logaddexp(lmin, s->loglstar) + logdvol;
    const double logz_new = logaddexp(s->logz, logwt);
    const double lz = ((s->loglstar > -1e299) ? exp(s->loglstar - logz_new + logdvol) * s->loglstar : 0.0) +
                      (isfinite(lmin) ? exp(lmin - logz_new + logdvol) * lmin : 0.0);
    const double h_new = lz + ((s->logz > -1e299) ? exp(s->logz - logz_new) * (s->h + s->logz) : 0.0) - logz_new;
    const double dh = h_new - s->h;
    s->h = h_new; s->logz = logz_new; s->logzvar += dh * dlv; s->logvol = logvol; s->loglstar = lmin;
    // record
    const int e = emitted++;
    out->worst[e] = worst;
    for (int d = 0; d < nd; ++d) { out->u[(size_t)e * nd + d] = live_u[(size_t)worst * nd + d]; out->v[(size_t)e * nd + d] = live_v[(size_t)worst * nd + d]; }
    out->logl[e] = lmin; out->logvol[e] = logvol; out->logwt[e] = logwt; out->logz[e] = s->logz;
    out->logzvar[e] = s->logzvar; out->h[e] = s->h; out->nc[e] = (int)nc; out->worst_it[e] = live_it[worst];
    // replace
    for (int d = 0; d < nd; ++d) { live_u[(size_t)worst * nd + d] = qu[(size_t)qpos * nd + d]; live_v[(size_t)worst * nd + d] = qv[(size_t)qpos * nd + d]; }
    live_logl[worst] = ql[qpos];
    live_it[worst] = (int)s->it;
    if (ql[qpos] > lmax) lmax = ql[qpos];
    ++qpos;
    out->delta_logz[e] = logaddexp(s->logz, lmax + s->logvol) - s->logz;
    s->it += 1;
  }
  *consumed = qpos;
  return emitted;
}
