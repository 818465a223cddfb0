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

#include <charconv>
#include <cstring>
#include <vector>

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
  // the maximum only ever grows (the point that leaves is the minimum), so it is tracked; the minimum is the top of a binary
  // heap of live-point indices ordered by (logl, index) -- the index breaks ties the way a first-minimum scan does (several -inf
  // points at the start of a run), so the records are those of the scan, at O(log n) per dead point instead of a pass over the
  // live set (0.5 us of the 0.76 us an iteration took at 512 live points)
  double lmax = live_logl[0];
  for (int i = 1; i < n; ++i) lmax = live_logl[i] > lmax ? live_logl[i] : lmax;
  std::vector<int> heap(n);
  for (int i = 0; i < n; ++i) heap[i] = i;
  auto less = [&](int a, int b) { const double la = live_logl[a], lb = live_logl[b]; return la < lb || (la == lb && a < b); };
  auto sift = [&](int pos) {
    const int v = heap[pos];
    while (true) {
      int c = 2 * pos + 1;
      if (c >= n) break;
      if (c + 1 < n && less(heap[c + 1], heap[c])) ++c;
      if (!less(heap[c], v)) break;
      heap[pos] = heap[c];
      pos = c;
    }
    heap[pos] = v;
  };
  for (int i = n / 2 - 1; i >= 0; --i) sift(i);
  // delta ln Z of the state at the head of an iteration IS the delta_logz recorded at the end of the one before: computed once
  double delta = (s->logz > -1e299) ? logaddexp(s->logz, lmax + s->logvol) - s->logz : INFINITY;
  while (true) {
    const int worst = heap[0];
    const double lmin = live_logl[worst];
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
    // logaddexp(lmin, loglstar) with its exponential kept: e1 = exp(-|lmin - loglstar|) is also the ratio of the two weights below
    double lae, e1;
    const bool min_big = lmin >= s->loglstar;
    if (lmin == s->loglstar) { lae = lmin + 0.6931471805599453; e1 = 1.0; }
    else {
      const double d1 = min_big ? s->loglstar - lmin : lmin - s->loglstar;
      if (d1 == d1) { e1 = exp(d1); lae = (min_big ? lmin : s->loglstar) + log1p(e1); }
      else { e1 = 0.0; lae = lmin + s->loglstar; }
    }
    const double logwt = lae + logdvol;
    // logz_new = logaddexp(logz, logwt) and, from the same exponential, exp(logz - logz_new) (H's weight of the old evidence):
    // with d = -|logz - logwt| and e = exp(d), logz_new = max + log1p(e) and the weight is 1 / (1 + e) or e / (1 + e)
    double logz_new, w_old;
    if (s->logz == logwt) { logz_new = logwt + 0.6931471805599453; w_old = 0.5; }
    else {
      const bool zbig = s->logz > logwt;
      const double d = zbig ? logwt - s->logz : s->logz - logwt;
      if (d == d) {
        const double e = exp(d), q = 1.0 / (1.0 + e);
        logz_new = (zbig ? s->logz : logwt) + log1p(e);
        w_old = zbig ? q : e * q;
      } else { logz_new = s->logz + logwt; w_old = exp(s->logz - logz_new); }     // (NaN / inf - inf: as logaddexp)
    }
    // exp(loglstar - logz_new + logdvol) loglstar + exp(lmin - logz_new + logdvol) lmin: one exponential (of the larger), the other is e1 times it
    double lz = 0.0;
    {
      const bool has_star = s->loglstar > -1e299, has_min = isfinite(lmin);
      if (has_star && has_min) {
        const double xb = exp((min_big ? lmin : s->loglstar) - logz_new + logdvol), xs = xb * e1;
        lz = (min_big ? xs : xb) * s->loglstar + (min_big ? xb : xs) * lmin;
      } else if (has_star) lz = exp(s->loglstar - logz_new + logdvol) * s->loglstar;
      else if (has_min) lz = exp(lmin - logz_new + logdvol) * lmin;
    }
    const double h_new = lz + ((s->logz > -1e299) ? w_old * (s->h + s->logz) : 0.0) - logz_new;
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
    sift(0);                                              // the new point (logl above the old minimum) sinks to its place
    if (ql[qpos] > lmax) lmax = ql[qpos];
    ++qpos;
    delta = (s->logz > -1e299) ? logaddexp(s->logz, lmax + s->logvol) - s->logz : INFINITY;
    out->delta_logz[e] = delta;
    s->it += 1;
  }
  *consumed = qpos;
  return emitted;
}

// What the live set and the threshold WILL be once payne_ns_consume has walked the whole queue (no stop condition but the
// queue's end): the replacements only -- same heap order, same test; no evidence arithmetic, no records.  The batched sampler
// launches its next queue of proposals from this state before it consumes the current one, so that the GPU walks while the host
// does the bookkeeping (thepayne_amd/sampler/nested.py: pipeline).
namespace payne_ns {
// By index: src[i] = the queue row that will sit in live slot i, or -1 (the live point stays); lg[i] its lnprob.
inline void peek_index(int n, const double* live_logl, const double* ql, int nq, std::vector<int>& src, std::vector<double>& lg,
                       std::vector<int>& heap, double* loglstar, int* n_dead) {
  src.assign(n, -1);
  lg.assign(live_logl, live_logl + n);
  heap.resize(n);
  for (int i = 0; i < n; ++i) heap[i] = i;
  auto less = [&](int a, int b) { const double la = lg[a], lb = lg[b]; return la < lb || (la == lb && a < b); };
  auto sift = [&](int pos) {
    const int v = heap[pos];
    while (true) {
      int c = 2 * pos + 1;
      if (c >= n) break;
      if (c + 1 < n && less(heap[c + 1], heap[c])) ++c;
      if (!less(heap[c], v)) break;
      heap[pos] = heap[c];
      pos = c;
    }
    heap[pos] = v;
  };
  for (int i = n / 2 - 1; i >= 0; --i) sift(i);
  int qpos = 0, m = 0;
  while (true) {
    const int worst = heap[0];
    const double lmin = lg[worst];
    while (qpos < nq && !(ql[qpos] > lmin)) ++qpos;
    if (qpos >= nq) break;
    *loglstar = lmin;
    src[worst] = qpos;
    lg[worst] = ql[qpos];
    sift(0);
    ++qpos;
    ++m;
  }
  *n_dead = m;
}
}  // namespace payne_ns

extern "C" int payne_ns_peek(int nlive, int ndim, const double* live_u, const double* live_v, const double* live_logl,
                             const double* qu, const double* qv, const double* ql, int nq, double* out_u, double* out_v,
                             double* out_logl, double* loglstar, int* n_dead) {
  if (nlive <= 0 || ndim <= 0 || nq < 0 || !live_u || !live_v || !live_logl || !out_u || !out_v || !out_logl || !loglstar || !n_dead)
    return PAYNE_E_INVALID;
  if (nq > 0 && (!qu || !qv || !ql)) return PAYNE_E_INVALID;
  const int n = nlive, nd = ndim;
  std::vector<int> src, heap;
  std::vector<double> lg;
  payne_ns::peek_index(n, live_logl, ql, nq, src, lg, heap, loglstar, n_dead);
  for (int i = 0; i < n; ++i) {
    const bool q = src[i] >= 0;
    std::memcpy(out_u + (size_t)i * nd, q ? qu + (size_t)src[i] * nd : live_u + (size_t)i * nd, (size_t)nd * 8);
    std::memcpy(out_v + (size_t)i * nd, q ? qv + (size_t)src[i] * nd : live_v + (size_t)i * nd, (size_t)nd * 8);
    out_logl[i] = lg[i];
  }
  return PAYNE_OK;
}

// ---- bounding ellipsoids of the live points ------------------------------------------------
// dynesty's bound='single' / 'multi' as the reference requests them (fitstar.py:314): the smallest
// scaled covariance ellipsoid holding every live point (unit-cube coordinates), enlarged in volume;
// for 'multi' the cloud is split recursively by 2-means while the children's ellipsoids hold less than
// half the parent's volume.  Same arithmetic as thepayne_amd/sampler/nested.py (_Ell, _split_ellipsoids);
// here because one fit costs ~0.1 ms in numpy and a decomposition needs a dozen of them per update.
namespace payne_ns {

struct Ell {
  std::vector<double> ctr, L, Linv, cov;   // L lower-triangular Cholesky factor of cov, row-major [nd][nd]
  double f = 1.0, logvol = 0.0;
};

inline bool cholesky(const std::vector<double>& a, int nd, std::vector<double>& L) {
  L.assign((size_t)nd * nd, 0.0);
  for (int i = 0; i < nd; ++i)
    for (int j = 0; j <= i; ++j) {
      double sum = a[(size_t)i * nd + j];
      for (int k = 0; k < j; ++k) sum -= L[(size_t)i * nd + k] * L[(size_t)j * nd + k];
      if (i == j) {
        if (!(sum > 0.0)) return false;
        L[(size_t)i * nd + i] = sqrt(sum);
      } else {
        L[(size_t)i * nd + j] = sum / L[(size_t)j * nd + j];
      }
    }
  return true;
}

inline bool fit_ell(const double* u, const int* idx, int n, int nd, double enlarge, Ell& e) {
  e.ctr.assign(nd, 0.0);
  for (int i = 0; i < n; ++i)
    for (int d = 0; d < nd; ++d) e.ctr[d] += u[(size_t)idx[i] * nd + d];
  for (int d = 0; d < nd; ++d) e.ctr[d] /= n;
  e.cov.assign((size_t)nd * nd, 0.0);
  std::vector<double> dev(nd);
  for (int i = 0; i < n; ++i) {
    for (int d = 0; d < nd; ++d) dev[d] = u[(size_t)idx[i] * nd + d] - e.ctr[d];
    for (int a = 0; a < nd; ++a)
      for (int b = 0; b <= a; ++b) e.cov[(size_t)a * nd + b] += dev[a] * dev[b];
  }
  double trace = 0.0;
  for (int a = 0; a < nd; ++a) {
    for (int b = 0; b <= a; ++b) { e.cov[(size_t)a * nd + b] /= (n > 1 ? n - 1 : 1); e.cov[(size_t)b * nd + a] = e.cov[(size_t)a * nd + b]; }
    trace += e.cov[(size_t)a * nd + a];
  }
  double ridge = 1e-14 * fmax(1e-300, trace / nd);
  bool ok = false;
  for (int attempt = 0; attempt < 6 && !ok; ++attempt, ridge *= 1e3) {   // a degenerate cloud gets a larger ridge
    std::vector<double> c = e.cov;
    for (int a = 0; a < nd; ++a) c[(size_t)a * nd + a] += ridge;
    ok = cholesky(c, nd, e.L);
    if (ok) e.cov = c;
  }
  if (!ok) return false;
  // inverse of the triangular factor by forward substitution
  e.Linv.assign((size_t)nd * nd, 0.0);
  for (int c = 0; c < nd; ++c)
    for (int r = c; r < nd; ++r) {
      double sum = (r == c) ? 1.0 : 0.0;
      for (int k = c; k < r; ++k) sum -= e.L[(size_t)r * nd + k] * e.Linv[(size_t)k * nd + c];
      e.Linv[(size_t)r * nd + c] = sum / e.L[(size_t)r * nd + r];
    }
  double r2max = 0.0, logdet = 0.0;
  for (int i = 0; i < n; ++i) {
    for (int d = 0; d < nd; ++d) dev[d] = u[(size_t)idx[i] * nd + d] - e.ctr[d];
    double r2 = 0.0;
    for (int r = 0; r < nd; ++r) {
      double z = 0.0;
      for (int k = 0; k <= r; ++k) z += e.Linv[(size_t)r * nd + k] * dev[k];
      r2 += z * z;
    }
    r2max = r2 > r2max ? r2 : r2max;
  }
  for (int d = 0; d < nd; ++d) logdet += log(e.L[(size_t)d * nd + d]);
  e.f = sqrt(r2max) * pow(enlarge, 1.0 / nd);
  e.logvol = logdet + nd * log(e.f);
  return true;
}

inline void split_ell(const double* u, const std::vector<int>& idx, int nd, double enlarge, const Ell& ell, int& budget,
                      double vol_dec, std::vector<Ell>& out) {
  const int n = (int)idx.size();
  if (n < 4 * nd + 2 || budget <= 1) { out.push_back(ell); return; }
  // direction of largest spread: power iteration on the covariance
  std::vector<double> dir(nd), tmp(nd);
  for (int d = 0; d < nd; ++d) dir[d] = 1.0 + 0.01 * d;
  for (int it = 0; it < 64; ++it) {
    double nrm = 0.0;
    for (int a = 0; a < nd; ++a) {
      double v = 0.0;
      for (int b = 0; b < nd; ++b) v += ell.cov[(size_t)a * nd + b] * dir[b];
      tmp[a] = v; nrm += v * v;
    }
    nrm = sqrt(nrm);
    if (!(nrm > 0.0)) { out.push_back(ell); return; }
    for (int a = 0; a < nd; ++a) dir[a] = tmp[a] / nrm;
  }
  std::vector<char> lab(n), nlab(n);
  std::vector<double> c0(nd), c1(nd), w(nd);
  auto proj = [&](int i, const std::vector<double>& v) {
    double p = 0.0;
    for (int d = 0; d < nd; ++d) p += (u[(size_t)idx[i] * nd + d] - ell.ctr[d]) * v[d];
    return p;
  };
  int n1 = 0;
  for (int i = 0; i < n; ++i) { lab[i] = proj(i, dir) > 0.0; n1 += lab[i]; }
  for (int it = 0; it < 8; ++it) {
    if (n1 == 0 || n1 == n) { out.push_back(ell); return; }
    std::fill(c0.begin(), c0.end(), 0.0); std::fill(c1.begin(), c1.end(), 0.0);
    for (int i = 0; i < n; ++i) {
      std::vector<double>& c = lab[i] ? c1 : c0;
      for (int d = 0; d < nd; ++d) c[d] += u[(size_t)idx[i] * nd + d] - ell.ctr[d];
    }
    double q0 = 0.0, q1 = 0.0;
    for (int d = 0; d < nd; ++d) { c0[d] /= (n - n1); c1[d] /= n1; q0 += c0[d] * c0[d]; q1 += c1[d] * c1[d]; w[d] = c1[d] - c0[d]; }
    const double thr = 0.5 * (q1 - q0);
    int m1 = 0; bool same = true;
    for (int i = 0; i < n; ++i) { nlab[i] = proj(i, w) > thr; m1 += nlab[i]; same = same && (nlab[i] == lab[i]); }
    lab.swap(nlab); n1 = m1;
    if (same) break;
  }
  if (n1 < 2 * nd + 1 || n - n1 < 2 * nd + 1) { out.push_back(ell); return; }
  std::vector<int> i0, i1;
  i0.reserve(n - n1); i1.reserve(n1);
  for (int i = 0; i < n; ++i) (lab[i] ? i1 : i0).push_back(idx[i]);
  Ell e0, e1;
  if (!fit_ell(u, i0.data(), (int)i0.size(), nd, enlarge, e0) || !fit_ell(u, i1.data(), (int)i1.size(), nd, enlarge, e1)) {
    out.push_back(ell); return;
  }
  if (logaddexp(e0.logvol, e1.logvol) >= ell.logvol + log(vol_dec)) { out.push_back(ell); return; }
  budget -= 1;
  split_ell(u, i0, nd, enlarge, e0, budget, vol_dec, out);
  split_ell(u, i1, nd, enlarge, e1, budget, vol_dec, out);
}

}  // namespace payne_ns

extern "C" int payne_ns_bound(const double* u, int n, int ndim, double enlarge, int multi, int max_ell, double* ctr,
                              double* axes, double* axes_unit, double* ainv, double* logvol, int* n_ell) {
  using namespace payne_ns;
  if (!u || !ctr || !axes || !axes_unit || !ainv || !logvol || !n_ell || n < 2 || ndim < 1 || max_ell < 1 || !(enlarge > 0.0))
    return PAYNE_E_INVALID;
  const int nd = ndim;
  std::vector<int> idx(n);
  for (int i = 0; i < n; ++i) idx[i] = i;
  Ell whole;
  if (!fit_ell(u, idx.data(), n, nd, enlarge, whole)) return PAYNE_E_INVALID;
  std::vector<Ell> ells;
  if (multi && max_ell > 1) {
    int budget = max_ell;
    split_ell(u, idx, nd, enlarge, whole, budget, 0.5, ells);
  } else {
    ells.push_back(whole);
  }
  const int E = (int)ells.size() > max_ell ? max_ell : (int)ells.size();
  const double su = sqrt(nd + 2.0);
  for (int e = 0; e < E; ++e) {
    const Ell& el = ells[e];
    for (int d = 0; d < nd; ++d) ctr[(size_t)e * nd + d] = el.ctr[d];
    for (int i = 0; i < nd * nd; ++i) {
      axes[(size_t)e * nd * nd + i] = el.L[i] * el.f;
      axes_unit[(size_t)e * nd * nd + i] = el.L[i] * su;
      ainv[(size_t)e * nd * nd + i] = el.Linv[i] / el.f;
    }
    logvol[e] = el.logvol;
  }
  *n_ell = E;
  return PAYNE_OK;
}

// ---- text rows of the output table -----------------------------------------------------------
// The reference writes one row per dead point with Python's str() of every value
// (Payne/fitting/fitstar.py:345-371): shortest round-trip digits, fixed notation with a trailing ".0" for
// integers while -4 < decimal-point position <= 16, otherwise d.ddde+XX.  Formatting ~140 k values per fit in
// Python costs as much as the sampling itself on the GPU; this is the same text from std::to_chars.
namespace payne_ns {

inline size_t py_float_repr(double x, char* out) {      // out: >= 32 bytes
  if (x != x) { memcpy(out, "nan", 3); return 3; }
  size_t n = 0;
  if (signbit(x)) { out[n++] = '-'; x = -x; }
  if (isinf(x)) { memcpy(out + n, "inf", 3); return n + 3; }
  char buf[40];
  const auto r = std::to_chars(buf, buf + sizeof(buf), x, std::chars_format::scientific);   // d[.ddd]e[+-]XX, shortest
  char digits[24];
  int nd = 0, e10 = 0;
  const char* p = buf;
  for (; p < r.ptr && *p != 'e'; ++p)
    if (*p != '.') digits[nd++] = *p;
  if (p < r.ptr) {
    ++p;
    const bool neg = (*p == '-');
    if (*p == '+' || *p == '-') ++p;
    for (; p < r.ptr; ++p) e10 = 10 * e10 + (*p - '0');
    if (neg) e10 = -e10;
  }
  while (nd > 1 && digits[nd - 1] == '0') --nd;          // (to_chars never pads, but keep the invariant)
  const int decpt = e10 + 1;
  if (decpt > -4 && decpt <= 16) {
    if (decpt <= 0) {
      out[n++] = '0'; out[n++] = '.';
      for (int i = 0; i < -decpt; ++i) out[n++] = '0';
      memcpy(out + n, digits, nd); n += nd;
    } else if (decpt >= nd) {
      memcpy(out + n, digits, nd); n += nd;
      for (int i = nd; i < decpt; ++i) out[n++] = '0';
      out[n++] = '.'; out[n++] = '0';
    } else {
      memcpy(out + n, digits, decpt); n += decpt;
      out[n++] = '.';
      memcpy(out + n, digits + decpt, nd - decpt); n += nd - decpt;
    }
    return n;
  }
  out[n++] = digits[0];
  if (nd > 1) { out[n++] = '.'; memcpy(out + n, digits + 1, nd - 1); n += nd - 1; }
  out[n++] = 'e';
  out[n++] = e10 < 0 ? '-' : '+';
  const int ae = e10 < 0 ? -e10 : e10;
  if (ae >= 100) out[n++] = (char)('0' + ae / 100);
  out[n++] = (char)('0' + (ae / 10) % 10);
  out[n++] = (char)('0' + ae % 10);
  return n;
}

}  // namespace payne_ns

// vals [m][ncol] row-major; is_int[c] != 0: the column is written as an integer.  Every value is followed by
// one blank, every row by a newline (the reference's row text).  Returns the number of bytes written, or
// PAYNE_E_INVALID if `cap` is too small (48 bytes per value are always enough).
extern "C" long long payne_format_rows(const double* vals, int m, int ncol, const int* is_int, char* out, long long cap) {
  if (!vals || !is_int || !out || m < 0 || ncol <= 0) return PAYNE_E_INVALID;
  long long n = 0;
  for (int i = 0; i < m; ++i) {
    for (int c = 0; c < ncol; ++c) {
      if (cap - n < 48) return PAYNE_E_INVALID;
      const double v = vals[(size_t)i * ncol + c];
      if (is_int[c]) {
        const auto r = std::to_chars(out + n, out + n + 32, (long long)v);
        n = r.ptr - out;
      } else {
        n += (long long)payne_ns::py_float_repr(v, out + n);
      }
      out[n++] = ' ';
    }
    out[n++] = '\n';
  }
  return n;
}
