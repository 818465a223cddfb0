// sampler_kernels.hpp -- the sampler step on the device: prior transform and ln-prior
// (Payne/fitting/prior.py:126-465), theta rows, random-walk proposals (one wave per chain).
// Device functions only (no kernels): included by sampler_kernels.hpp (payne_hip.hip) and by post_kernels.hpp, whose
// likelihood-only kernel runs a chain's next step at its tail.
#pragma once
#include "../../include/payne_hip.h"
#ifdef __HIPCC__

// ============================================================================
// device-side sampler step: prior transform, ln-prior, theta rows, random-walk proposals
// ============================================================================
constexpr int kMaxThetaCols = 8 + PAYNE_MAX_POLY + 4;
struct SamplerDev {
  int ndim, ncols, nfixed;
  payne_prior_dim dims[PAYNE_MAX_DIM];
  int fixed_col[PAYNE_MAX_FIXED];
  double fixed_val[PAYNE_MAX_FIXED];
  payne_adv_priors adv;             // (tab_cdf / tab_val: DEVICE copies here)
  // the theta row by column (what write_theta_row does, resolved once): col_src >= 0 the sampled dimension that fills the
  // column, < 0 the constant col_val (a fixed value or NaN = absent)
  int col_src[kMaxThetaCols];
  double col_val[kMaxThetaCols];
  // what a transform computes from its parameters alone, once (prior_cache): the truncated normal's two (survival) CDF values --
  // two normcdf and two divisions a call otherwise, half of a chain step's instructions --, expm1(-b), the logarithms of log-uniform
  double q0[PAYNE_MAX_DIM], q1[PAYNE_MAX_DIM];
};

// np.interp(u, xp, fp), xp non-decreasing: the last j with xp[j] <= u, slope form (advancedpriors.gal_ppf)
__device__ inline double table_interp(const payne_adv_priors& a, double u) {
  const double* xp = a.tab_cdf;
  const double* fp = a.tab_val;
  const int n = a.tab_n;
  if (!(u == u)) return u;
  if (u > xp[n - 1]) return fp[n - 1];
  if (u < xp[0]) return fp[0];
  int lo = 0, hi = n;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (xp[mid] <= u) lo = mid; else hi = mid; }
  if (lo == n - 1 || xp[lo] == u) return fp[lo];
  return (fp[lo + 1] - fp[lo]) / (xp[lo + 1] - xp[lo]) * (u - xp[lo]) + fp[lo];
}
// imf_lnprior (advancedpriors.py:93-137) and vrot_lnprior (:691-733) of the values gathered by the caller
__device__ inline double adv_lnprior(const payne_adv_priors& a, double logg, double logr, double vrot, double dist) {
  double lp = 0.0;
  if (a.imf) {
    const double m = pow(10.0, logg + 2.0 * logr - 4.437);                     // prior.py:291-294
    const double al = 1.3, ah = 2.3, mb = 0.5;
    double v;
    if (m > mb) v = -ah * log(m) + (ah - al) * log(mb);
    else if (m > 0.08) v = -al * log(m);
    else v = (m == m) ? -INFINITY : m;
    const double norm = pow(mb, 1.0 - al) / (ah - 1.0) + pow(0.08, 1.0 - al) / (al - 1.0) - pow(mb, 1.0 - al) / (al - 1.0);
    lp += v - log(norm);
  }
  if (a.vrot) {
    const bool have = !a.vrot_mass_one && (logg - logg == 0.0) && (logr - logr == 0.0);   // both finite
    const double mass = have ? pow(10.0, logg + 2.0 * logr) : 1.0;              // prior.py:320-331 (no zero point here)
    const bool hot = mass > 1.25, gi = !hot && (logg < 3.5);                     // eep = 350 < 450
    const double aa = hot ? -1.0 : -10.0, cc = hot ? 100.0 : (gi ? 7.0 : 10.0), nn = hot ? 1.0 : (gi ? 1.0 : 0.4);
    lp += aa / (1.0 + nn * exp(-(vrot - cc)));
  }
  if (a.plx_dim >= 0) {                                                        // 'Parallax' = 1000 / Dist, prior.py:449-451
    const double plx = 1000.0 / dist;
    if (a.plx_has_gauss) { const double z = plx - a.plx_mu; lp += -0.5 * ((z * z) / (a.plx_sigma * a.plx_sigma)); }
    if (a.plx_has_box && ((plx < a.plx_lo) || (plx > a.plx_hi))) lp = -INFINITY;
  }
  return lp;
}
__device__ __forceinline__ bool adv_any(const payne_adv_priors& a) { return a.imf || a.vrot || a.plx_dim >= 0; }

// unit cube -> parameter (Payne/fitting/prior.py:151-178, scipy.stats ppf's restated)
__device__ inline void prior_cache(const payne_prior_dim& d, double& q0, double& q1) {
  q0 = 0.0; q1 = 0.0;
  switch (d.kind) {
    case PAYNE_PRIOR_TGAUSSIAN: {
      const double a = (d.p[0] - d.p[2]) / d.p[3], b = (d.p[1] - d.p[2]) / d.p[3];
      if (a > 0.0) { const double sa = normcdf(-a), sb = normcdf(-b); q0 = sa; q1 = sa - sb; }   // upper tail: survival functions
      else { const double ca = normcdf(a), cb = normcdf(b); q0 = ca; q1 = cb - ca; }
      break;
    }
    case PAYNE_PRIOR_TEXP: q0 = expm1(-((d.p[1] - d.p[0]) / d.p[2])); break;
    case PAYNE_PRIOR_LOGUNIFORM: q0 = log(d.p[0]); q1 = log(d.p[1]) - log(d.p[0]); break;
    default: break;
  }
}
__device__ inline double prior_ppf(const payne_prior_dim& d, double q0, double q1, double u, const payne_adv_priors& adv) {
  switch (d.kind) {
    case PAYNE_PRIOR_TABLE: return d.p[0] * table_interp(adv, u);
    case PAYNE_PRIOR_UNIFORM: {
      const double lo = fmin(d.p[0], d.p[1]), hi = fmax(d.p[0], d.p[1]);
      return (hi - lo) * u + lo;
    }
    case PAYNE_PRIOR_GAUSSIAN: return d.p[0] + d.p[1] * normcdfinv(u);
    case PAYNE_PRIOR_TGAUSSIAN: {
      // both limits in the upper tail (lo > mu): survival functions, q0 = S(a), q1 = S(a) - S(b); else q0 = C(a), q1 = C(b) - C(a)
      const double x = (d.p[0] > d.p[2]) ? -normcdfinv(q0 - u * q1) : normcdfinv(q0 + u * q1);
      double v = d.p[2] + d.p[3] * x;
      if (!(v <= d.p[1])) v = (v != v) ? v : d.p[1];           // +inf (u = 1) -> hi, prior.py:165-166
      return v;
    }
    case PAYNE_PRIOR_EXP: return d.p[0] - d.p[1] * log1p(-u);
    case PAYNE_PRIOR_TEXP: {
      double v = d.p[0] - d.p[2] * log1p(u * q0);               // truncexpon.ppf, q0 = expm1(-b)
      if (!(v <= d.p[1])) v = (v != v) ? v : d.p[1];
      return v;
    }
    case PAYNE_PRIOR_LOGUNIFORM: return exp(q0 + u * q1);
    default: return u;
  }
}
__device__ inline double prior_ln(const payne_prior_dim& d, double v) {
  double lp = 0.0;
  if (d.has_gauss) { const double z = v - d.g_mu; lp += -0.5 * ((z * z) / (d.g_sigma * d.g_sigma)); }
  if (d.has_box && ((v < d.box_lo) || (v > d.box_hi))) lp = -INFINITY;
  return lp;
}
// one theta row: NaN = absent, fixed values, then the sampled dimensions
__device__ inline void write_theta_row(const SamplerDev& sd, const double* v, double* row) {
  for (int c = 0; c < sd.ncols; ++c) row[c] = __builtin_nan("");
  for (int i = 0; i < sd.nfixed; ++i) row[sd.fixed_col[i]] = sd.fixed_val[i];
  for (int d = 0; d < sd.ndim; ++d) if (sd.dims[d].theta_col >= 0) row[sd.dims[d].theta_col] = v[d];
}

// counter-based generator: splitmix64 of (seed, chain, step, draw)
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ double u01(unsigned long long seed, unsigned chain, unsigned step, unsigned draw) {
  const unsigned long long x = mix64(mix64(seed ^ ((unsigned long long)chain << 32 | step)) + draw);
  return ((double)(x >> 11) + 0.5) * (1.0 / 9007199254740992.0);      // (0,1)
}

__device__ __forceinline__ float u01f(unsigned long long seed, unsigned chain, unsigned step, unsigned draw) {
  const unsigned long long x = mix64(mix64(seed ^ ((unsigned long long)chain << 32 | step)) + draw);
  return ((float)(unsigned)(x >> 40) + 0.5f) * (1.0f / 16777216.0f);   // (0,1), 24 bits
}


// Everything one walk touches (device pointers) and its constants.
struct WalkState {
  double *u, *v, *lnprob;
  int *nacc, *ncall;
  double *u_prop, *v_prop, *lnprior_prop;
  int* inside;
  double* rows;
  const double* axes;
  const int* ell;
  int* nredraw;
  double scale, loglstar;
  unsigned long long seed;
  int K;
  int nd, ncols, adv_on;       // of the sampler (SamplerDev::ndim / ncols / adv_any): the step's first loads need no table read first
  double* spec;                // [K][2][kSpecStride] the next proposal made ahead for both outcomes of the pending one (rwalk_spec_wave); null: none
  // the queue call (payne_ns_rwalk_queue_begin): the walk's first step also finds the ellipsoid each chain steps in -- one that holds
  // its start point (a random one of those), else the nearest -- and writes it to ell_out; null: `ell` came from the host
  const double* as_ctr; const double* as_ainv; int* ell_out; int n_ell;
  // the queue launched WITHOUT the host in between (payne_ns_queue_dev_launch): scale and loglstar are not known when the
  // launch is made -- payne_ns_turn_kernel writes them here, {scale, loglstar}; null: the two fields above hold them
  const double* dyn;
};
__device__ __forceinline__ double walk_scale(const WalkState& W) { return W.dyn ? W.dyn[0] : W.scale; }
__device__ __forceinline__ double walk_lstar(const WalkState& W) { return W.dyn ? W.dyn[1] : W.loglstar; }
// device-resident copy for the post kernel's tail (payne_post_kernel<.., LEAN>: PostArgs::tail)
struct WalkTail { SamplerDev sd; WalkState w; };

// One random-walk step for every chain: first settle the previous proposal (accept iff inside the
// cube and lnprob > loglstar), then draw the next one.  `propose` = 0 on the closing call.
// ONE WAVE PER CHAIN, lane d = sampled dimension d: the inverse CDFs (the expensive part: normcdf /
// normcdfinv chains in fp64) of the dimensions run side by side, the ellipsoid step is a shuffle
// matvec, sums are wave reductions.  (One thread per chain spent 14 us per step in a ~3000-instruction
// dependent fp64 chain; the step sits between two likelihood batches, nothing overlaps it.)
constexpr int kRedrawPasses = 2;
#ifndef PAYNE_AX_BATCH
#define PAYNE_AX_BATCH 4
#endif
constexpr int kAxBatch = PAYNE_AX_BATCH;
// Lane l ^ J's value without the LDS crossbar: data-parallel-primitive moves (J <= 8: quad permutes, the mirrors of a half row and of a
// quad composed, a row rotated by 8) and gfx950's row / half swaps (J = 16, 32: both copies go in, the partner's value comes back in one
// of them, which one by the lane's own bit `upper` = lane & J).  A `ds_bpermute` round trip is ~130 cycles, and the sums below are chains
// of four to six of them on the critical path of a chain step; these are the same values, so the same sums to the bit.
template <int J>
__device__ __forceinline__ int lane_xor_i32(int v, bool upper) {
  if constexpr (J == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);          // quad_perm [1,0,3,2]
  else if constexpr (J == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
  else if constexpr (J == 4) {
    const int t = __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);                      // row_half_mirror: l ^ 7
    return __builtin_amdgcn_update_dpp(0, t, 0x1B, 0xf, 0xf, true);                              // quad_perm [3,2,1,0]: ^ 3
  } else if constexpr (J == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, true);  // row_ror:8
  else if constexpr (J == 16) { const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false); return (int)(upper ? r[0] : r[1]); }
  else { static_assert(J == 32, "in-wave distance"); const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false); return (int)(upper ? r[0] : r[1]); }
}
template <int J>
__device__ __forceinline__ double lane_xor_f64(double x, int lane) {
  union { double d; int w[2]; } a, b;
  a.d = x;
  const bool upper = (lane & J) != 0;
  b.w[0] = lane_xor_i32<J>(a.w[0], upper); b.w[1] = lane_xor_i32<J>(a.w[1], upper);
  return b.d;
}
template <int J>
__device__ __forceinline__ float lane_xor_f32(float x, int lane) {
  return __int_as_float(lane_xor_i32<J>(__float_as_int(x), (lane & J) != 0));
}
// x summed over the lanes of every aligned group of NP (a power of two, 8 .. 64), in every lane: partner distances NP/2 .. 1 in that order
__device__ __forceinline__ float group_sum_f32(float x, int NP, int lane) {
  if (NP >= 64) x += lane_xor_f32<32>(x, lane);
  if (NP >= 32) x += lane_xor_f32<16>(x, lane);
  if (NP >= 16) x += lane_xor_f32<8>(x, lane);
  x += lane_xor_f32<4>(x, lane); x += lane_xor_f32<2>(x, lane); x += lane_xor_f32<1>(x, lane);
  return x;
}
__device__ __forceinline__ double wave_sum(double x) {          // (partner distances 32, 16, .. 1 in that order)
  const int lane = (int)(threadIdx.x & 63);
  x += lane_xor_f64<32>(x, lane); x += lane_xor_f64<16>(x, lane); x += lane_xor_f64<8>(x, lane);
  x += lane_xor_f64<4>(x, lane); x += lane_xor_f64<2>(x, lane); x += lane_xor_f64<1>(x, lane);
  return x;
}
// What a step reads from global memory before it can do anything: requested in one go -- first what hangs off the
// sampler's tables, then what hangs off the walk's pointers (the step is a chain of dependent L2 / HBM round trips
// otherwise, 0.3-1 us each: seven of them as first written).
struct WalkLoads {
  double uc, vc, u_p, v_p, lpr;
  int was_in, my_ell, ncall0, nacc0, nredraw0;
  int col_src; double col_val;                                  // this lane's theta column
  int nd, ncols, adv_on;
};
__device__ __forceinline__ WalkLoads walk_loads(const SamplerDev& sd, const WalkState& W, int c, int lane) {
  WalkLoads L;                                                  // (valid addresses whatever the flags say)
  L.nd = W.nd; L.ncols = W.ncols;
  L.adv_on = W.adv_on;
  const int nd = L.nd;
  const int dl = lane < nd ? lane : 0;
  const int colc = lane < L.ncols ? lane : 0;
  L.col_src = sd.col_src[colc]; L.col_val = sd.col_val[colc];
  const size_t off = (size_t)c * nd + dl;
  L.uc = W.u[off]; L.vc = W.v[off];
  L.was_in = W.inside[c]; L.lpr = W.lnprior_prop[c];
  L.u_p = W.u_prop[off]; L.v_p = W.v_prop[off];
  L.ncall0 = W.ncall[c]; L.nacc0 = W.nacc[c];
  L.nredraw0 = W.nredraw ? W.nredraw[c] : 0;
  L.my_ell = W.ell ? W.ell[c] : 0;
  return L;
}
// A wave-uniform record behind a pointer, as scalars: every lane loads it (one round trip for the whole record, where
// field-by-field reads interleaved with stores make one each), the first lane's copy goes to SGPRs.
template <class T>
__device__ __forceinline__ T uniform_copy(const T* p) {
  static_assert(sizeof(T) % 4 == 0, "dword-sized records");
  constexpr int N = (int)(sizeof(T) / 4);
  union U { T t; unsigned w[N]; __device__ U() {} } x;
  const unsigned* q = reinterpret_cast<const unsigned*>(p);
#pragma unroll
  for (int i = 0; i < N; ++i) x.w[i] = q[i];
#pragma unroll
  for (int i = 0; i < N; ++i) x.w[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)x.w[i]);
  return x.t;
}

// The step of chain c on one wave (lane = dimension), its loads done.  lnl_p: the likelihood of the pending proposal.
__device__ __forceinline__ void rwalk_step_core(const SamplerDev& sd, const WalkState& W, const WalkLoads& L, int c, int lane,
                                                double lnl_p, int step, int settle, int propose) {
  double* const u = W.u; double* const v = W.v; double* const lnprob = W.lnprob;
  int* const nacc = W.nacc; int* const ncall = W.ncall;
  double* const u_prop = W.u_prop; double* const v_prop = W.v_prop; double* const lnprior_prop = W.lnprior_prop;
  int* const inside = W.inside; double* const rows = W.rows;
  const double* const axes = W.axes; int* const nredraw = W.nredraw;
  const double scale = walk_scale(W), loglstar = walk_lstar(W);
  const unsigned long long seed = W.seed;
  const int nd = L.nd;
  const bool act = lane < nd;
  const int dl = act ? lane : 0;
  const size_t off = (size_t)c * nd + dl;
  double uc = L.uc;
  double vc = L.vc;
  const int was_in = L.was_in;
  const double lpr = L.lpr;
  const double u_p = L.u_p, v_p = L.v_p;
  const int my_ell = L.my_ell;
  // the ellipsoid coefficients of this lane's coordinate (up to eight: the usual fits) are requested before the settle and
  // the random numbers are worked out: one L2 round trip under ~800 cycles of arithmetic instead of two in the draw
  int NP = 8;
  while (NP < nd) NP <<= 1;
  const int G = 64 / NP, g = lane / NP, dg = lane - g * NP;
  const bool actg = dg < nd;
  const int dgl = actg ? dg : 0;
  const double* ax = axes + (size_t)my_ell * nd * nd;                // this chain's ellipsoid (bound='multi')
  double ax8[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) ax8[k] = propose ? ax[dgl * nd + (k < nd ? k : nd - 1)] : 0.0;
  if (!settle && lane == 0) { ncall[c] = 0; nacc[c] = 0; }     // a walk's first step: its counters start here (no memsets before it)
  if (settle && was_in) {
    const double lp = (lpr == -INFINITY) ? -INFINITY : lpr + lnl_p;
    const bool accept = lp > loglstar;                          // false for NaN
    if (accept && act) { uc = u_p; vc = v_p; u[off] = uc; v[off] = vc; }
    if (lane == 0) {
      ncall[c] = L.ncall0 + 1;
      if (accept) { lnprob[c] = lp; nacc[c] = L.nacc0 + 1; }
    }
  }
  if (!propose) return;
  // z uniform in the unit ball: normal direction (one Box-Muller cosine per lane), radius U^(1/n).  A proposal that
  // leaves the unit cube is redrawn at once, without a likelihood call, as dynesty's rwalk does (it counts such a
  // draw as a rejection for the scale adaptation: `nredraw`).  The wave draws 64 / NP candidates SIDE BY SIDE (NP =
  // dimensions rounded up to a power of two: lanes g NP .. g NP + NP - 1 hold candidate g) and takes the first one
  // inside the cube: one pass costs what one draw costs, and the step is given up only after kRedrawPasses passes.
  const double ucg = __shfl(uc, dgl);                               // the chain's position, seen by every candidate group
  double up = uc;
  bool in = false;
  int skipped = 0;
  for (int pass = 0; pass < kRedrawPasses && !in; ++pass) {
    const unsigned d0 = (unsigned)(pass * G + g) * 192u;
    // the random direction and radius in fp32 (v_log_f32 / v_cos_f32 / v_exp_f32 / v_rsq_f32: a draw carries 24 random
    // bits per coordinate anyway); the chain's position and the step added to it stay fp64
    float z = 0.f;
    if (actg) {
      const float a = u01f(seed, c, step, d0 + 2 * dg), b = u01f(seed, c, step, d0 + 2 * dg + 1);
      z = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(a)) * __builtin_amdgcn_cosf(b);   // -2 ln a = -2 ln2 log2 a; cos(2 pi b)
    }
    float n2 = z * z;
    n2 = group_sum_f32(n2, NP, lane);                              // sum over the candidate's own lanes
    const float rad = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(u01f(seed, c, step, d0 + 128)) / (float)nd) * __builtin_amdgcn_rsqf(n2);
    double sdot = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {                                   // the first eight coordinates: coefficients already here
      const float ze = __shfl(z, g * NP + (k & (NP - 1)));
      sdot = (k < nd) ? fma(ax8[k], (double)ze, sdot) : sdot;
    }
    // (the rest, kAxBatch coefficients requested at a time: a load -> wait -> fma loop pays one L2 round trip per dimension)
    for (int e0 = 8; e0 < nd; e0 += kAxBatch) {
      double ab[kAxBatch];
#pragma unroll
      for (int k = 0; k < kAxBatch; ++k) ab[k] = ax[dgl * nd + (e0 + k < nd ? e0 + k : nd - 1)];
#pragma unroll
      for (int k = 0; k < kAxBatch; ++k) {
        const float ze = __shfl(z, g * NP + ((e0 + k) & (NP - 1)));
        sdot = (e0 + k < nd) ? fma(ab[k], (double)ze, sdot) : sdot;
      }
    }
    const double upg = ucg + (scale * (double)rad) * sdot;
    const unsigned long long bad = __ballot(actg && !((upg > 0.0) && (upg < 1.0)));
    int first = -1;
    for (int q = G - 1; q >= 0; --q) {
      const unsigned long long m = (NP == 64 ? ~0ull : ((1ull << NP) - 1ull)) << (q * NP);
      if ((bad & m) == 0ull) first = q;
    }
    in = first >= 0;
    skipped += in ? first : G;
    up = __shfl(upg, (in ? first : 0) * NP + dl);                    // lanes d < nd take candidate `first`
  }
  if (nredraw && lane == 0) nredraw[c] = (settle ? L.nredraw0 : 0) + skipped;
  const payne_prior_dim dim = sd.dims[dl];                      // (an L2 hit; twenty registers the loop above could not spare)
  const double vp = in ? prior_ppf(dim, sd.q0[dl], sd.q1[dl], up, sd.adv) : vc;       // outside: a harmless valid row
  double lp = wave_sum(act ? prior_ln(dim, vp) : 0.0);
  if (L.adv_on) {                                       // priors on derived quantities: the values they need by shuffle
    const payne_adv_priors& a = sd.adv;
    const double g_ = __shfl(vp, a.dim_logg >= 0 ? a.dim_logg : 0), r_ = __shfl(vp, a.dim_logr >= 0 ? a.dim_logr : 0);
    const double v_ = __shfl(vp, a.dim_vrot >= 0 ? a.dim_vrot : 0), d_ = __shfl(vp, a.plx_dim >= 0 ? a.plx_dim : 0);
    const double add = adv_lnprior(a, a.dim_logg >= 0 ? g_ : a.val_logg, a.dim_logr >= 0 ? r_ : a.val_logr,
                                   a.dim_vrot >= 0 ? v_ : a.val_vrot, a.plx_dim >= 0 ? d_ : 1.0);
    lp = (lp == -INFINITY || add == -INFINITY) ? -INFINITY : lp + add;
  }
  if (act) { u_prop[off] = up; v_prop[off] = vp; }
  if (lane == 0) { inside[c] = in ? 1 : 0; lnprior_prop[c] = lp; }
  // theta row, lane = column: NaN = absent, fixed values, then the sampled dimensions
  const double vs = __shfl(vp, L.col_src >= 0 ? L.col_src : 0);
  const double val = L.col_src >= 0 ? vs : L.col_val;
  if (lane < L.ncols) rows[(size_t)c * L.ncols + lane] = val;
}
// The ellipsoid chain c steps in (the queue call's first step; on the host this loop was 8 us per ellipsoid for 512 chains, on the
// queue's critical path): d^2 = |Ainv (u - ctr)|^2 per ellipsoid, lane a = row a of the product, the sums in index order and without
// contraction (the host form of this loop, payne_ns_rwalk_queue_begin before it moved here, rounds the same way); among the
// ellipsoids that hold the point a reservoir choice on the chain's own hash stream, else the nearest.
__device__ __forceinline__ int walk_assign_ell(const WalkState& W, int c, int lane) {
#pragma clang fp contract(off)
  const int nd = W.nd;
  const bool act = lane < nd;
  const int a = act ? lane : 0;
  const double uc = W.u[(size_t)c * nd + a];
  unsigned long long r1 = mix64(mix64(W.seed ^ (0xA5A5A5A5ull + (unsigned long long)c * 0x100000001B3ull)));
  int pick = 0, best = 0, nin = 0;
  double dbest = INFINITY;
  for (int e = 0; e < W.n_ell; ++e) {
    const double* ce = W.as_ctr + (size_t)e * nd;
    const double* ai = W.as_ainv + (size_t)e * nd * nd;
    // (the row's elements requested together, sixteen at a time, the index clamped; the lanes' values fetched together; THEN the
    // sums, in index order as before -- one element, its wait, one product at a time this loop was nd memory latencies per ellipsoid,
    // 2.5 us each at twelve dimensions, in front of every queue's first likelihood batch)
    const double ce_a = ce[a];
    double y = 0.0;
    for (int b0 = 0; b0 < nd; b0 += 16) {
      double r[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) { const int b = b0 + q < nd ? b0 + q : nd - 1; r[q] = ai[a * nd + b]; }
      const double diff = uc - ce_a;
      double dq[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) dq[q] = __shfl(diff, (b0 + q) & 63);
#pragma unroll
      for (int q = 0; q < 16; ++q) if (b0 + q < nd) y += r[q] * dq[q];
    }
    const double y2 = act ? y * y : 0.0;
    double d2 = 0.0;
    for (int b0 = 0; b0 < nd; b0 += 16) {
      double t[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) t[q] = __shfl(y2, (b0 + q) & 63);
#pragma unroll
      for (int q = 0; q < 16; ++q) if (b0 + q < nd) d2 += t[q];
    }
    if (d2 < dbest) { dbest = d2; best = e; }
    if (d2 <= 1.0) {
      ++nin;
      r1 = mix64(r1);
      // (r1 % nin == 0: the 64-bit remainder is a routine of 200 instructions; the first two cases are the usual ones)
      const bool take = nin == 1 ? true : nin == 2 ? (r1 & 1ull) == 0 : r1 % (unsigned long long)nin == 0;
      if (take) pick = e;
    }
  }
  if (nin == 0) pick = best;
  if (lane == 0) W.ell_out[c] = pick;
  return pick;
}
__device__ __forceinline__ void rwalk_step_wave(const SamplerDev& sd, const WalkState& W, int c, int lane, double lnl_p, int step,
                                                int settle, int propose) {
  WalkLoads L = walk_loads(sd, W, c, lane);
  if (!settle && W.as_ctr) L.my_ell = walk_assign_ell(W, c, lane);
  rwalk_step_core(sd, W, L, c, lane, lnl_p, step, settle, propose);
}

// ----------------------------------------------------------------------------------------------------------------------
// The next proposal made AHEAD.  While the likelihood of proposal w is being computed, both things step w + 1 can start from are
// already known: the chain's position (proposal w rejected) and proposal w itself (accepted).  The draws are counter-based
// (seed, chain, step, draw), so the step's random direction and radius are the same either way; only the point they are added to,
// the redraws that follow from it, and the prior transform differ.  rwalk_spec_wave makes BOTH next proposals -- in workgroups of
// the hidden-layer launch that would otherwise idle (payne_dense_hidden_kernel) -- and the post kernel's tail only settles the
// pending proposal and copies the outcome's record (rwalk_settle_spec): a memory round trip instead of ~1 200 dependent fp64
// instructions on one wave at the end of every likelihood batch.  Same draws, same arithmetic, same order of every sum as
// rwalk_step_core: the chains are the same TO THE BIT (tests/test_sampler_gpu.py).
// Sixteen lanes per (chain, outcome): a wave makes two chains' records, a 256-thread workgroup eight chains'.  The walk's pointers and
// constants come by value (kernel arguments): behind the WalkTail pointer they were one more dependent memory round trip.
// ----------------------------------------------------------------------------------------------------------------------
constexpr int kSpecLanes = 16;                                   // lanes per (chain, outcome): sampled dimensions <= 16
constexpr int kSpecCols = 32;                                    // theta columns a record holds
constexpr int kSpecStride = 2 * kSpecLanes + kSpecCols + 2;      // doubles per record: u' | v' | theta row | ln-prior | (inside, skipped)
constexpr int kSpecChainsPerWg = 8;
__host__ __device__ inline bool spec_fits(int nd, int ncols) { return nd <= kSpecLanes && ncols <= kSpecCols; }

__device__ __forceinline__ void rwalk_spec_wave(const SamplerDev& sd, const WalkState& W, int pair, int lane, int step) {
  const int nd = W.nd, ncols = W.ncols;
  // lane = [chain of the pair][outcome][16]: o = 0 from the chain's position, 1 from the pending proposal
  const int l = lane & (kSpecLanes - 1), gb = lane & ~(kSpecLanes - 1), cb = lane & ~31, l32 = lane & 31, o = (lane >> 4) & 1;
  const int c_ = pair * 2 + (lane >> 5);
  const bool live = c_ < W.K;
  const int c = live ? c_ : W.K - 1;                             // (a dead half computes the last chain again and stores nothing)
  const bool act = l < nd;
  const int dl = act ? l : 0;
  const size_t off = (size_t)c * nd + dl;
  const double uc = (o ? W.u_prop : W.u)[off], vc = (o ? W.v_prop : W.v)[off];
  const double wscale = walk_scale(W);
  const int my_ell = W.ell ? W.ell[c] : 0;
  const payne_prior_dim dim = sd.dims[dl];
  const double q0 = sd.q0[dl], q1 = sd.q1[dl];
  int col_src[2]; double col_val[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = l + kSpecLanes * j, colc = col < ncols ? col : 0;
    col_src[j] = sd.col_src[colc]; col_val[j] = sd.col_val[colc];
  }
  // The candidates' displacements do not depend on the outcome: the chain's 32 lanes draw 32 / NP of them per pass (NP: the
  // dimensions rounded up as rwalk_step_core does), each is added to both starting points.  rwalk_step_core tries candidates
  // 0 .. kRedrawPasses * 64 / NP - 1 in order and keeps the first one inside the cube: the same ones in the same order here.
  const int NP = nd <= 8 ? 8 : 16;
  const int G = 32 / NP, g = l32 / NP, dg = l32 - g * NP;
  const bool actg = dg < nd;
  const int dgl = actg ? dg : 0;
  const double* ax = W.axes + (size_t)my_ell * nd * nd;
  double ax16[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) ax16[k] = ax[dgl * nd + (k < nd ? k : nd - 1)];
  const double ucgA = __shfl(uc, cb + dgl), ucgB = __shfl(uc, cb + kSpecLanes + dgl);
  const int gold = 64 / NP, maxcand = kRedrawPasses * gold;
  bool in = false;
  int skipped = 0;
  double up = uc, up_fail = uc;
  for (int c0 = 0; c0 < maxcand && __any(!in); c0 += G) {
    const unsigned d0 = (unsigned)(c0 + g) * 192u;
    float z = 0.f;
    if (actg) {
      const float a = u01f(W.seed, c, step, d0 + 2 * dg), b = u01f(W.seed, c, step, d0 + 2 * dg + 1);
      z = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(a)) * __builtin_amdgcn_cosf(b);
    }
    float n2 = z * z;
    n2 = group_sum_f32(n2, NP, lane);
    const float rad = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(u01f(W.seed, c, step, d0 + 128)) / (float)nd) * __builtin_amdgcn_rsqf(n2);
    double sdot = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float ze = __shfl(z, cb + g * NP + (k & (NP - 1)));
      sdot = (k < nd) ? fma(ax16[k], (double)ze, sdot) : sdot;
    }
    const double upgA = ucgA + (wscale * (double)rad) * sdot;
    const double upgB = ucgB + (wscale * (double)rad) * sdot;
    const unsigned gmA = (unsigned)(__ballot(actg && !((upgA > 0.0) && (upgA < 1.0))) >> cb);
    const unsigned gmB = (unsigned)(__ballot(actg && !((upgB > 0.0) && (upgB < 1.0))) >> cb);
    int fA = -1, fB = -1;
    for (int q = G - 1; q >= 0; --q) {
      const unsigned m = (NP == 16 ? 0xFFFFu : 0xFFu) << (q * NP);
      if ((gmA & m) == 0u) fA = q;
      if ((gmB & m) == 0u) fB = q;
    }
    const double candA = __shfl(upgA, cb + (fA >= 0 ? fA : 0) * NP + dl), candB = __shfl(upgB, cb + (fB >= 0 ? fB : 0) * NP + dl);
    const double cand0A = __shfl(upgA, cb + dl), cand0B = __shfl(upgB, cb + dl);
    const int first = o ? fB : fA;
    const bool found = first >= 0;
    if (c0 == gold * (kRedrawPasses - 1)) up_fail = o ? cand0B : cand0A;   // what rwalk_step_core leaves in u_prop when every candidate is outside
    if (!in) { up = o ? candB : candA; skipped += found ? first : G; in = found; }
  }
  if (!in) up = up_fail;
  const double vp = in ? prior_ppf(dim, q0, q1, up, sd.adv) : vc;
  double lp = act ? prior_ln(dim, vp) : 0.0;
  lp += lane_xor_f64<8>(lp, lane); lp += lane_xor_f64<4>(lp, lane); lp += lane_xor_f64<2>(lp, lane); lp += lane_xor_f64<1>(lp, lane);   // (wave_sum's last four levels: the others add zeros)
  if (W.adv_on) {
    const payne_adv_priors& a = sd.adv;
    const double g_ = __shfl(vp, gb + (a.dim_logg >= 0 ? a.dim_logg : 0)), r_ = __shfl(vp, gb + (a.dim_logr >= 0 ? a.dim_logr : 0));
    const double v_ = __shfl(vp, gb + (a.dim_vrot >= 0 ? a.dim_vrot : 0)), d_ = __shfl(vp, gb + (a.plx_dim >= 0 ? a.plx_dim : 0));
    const double add = adv_lnprior(a, a.dim_logg >= 0 ? g_ : a.val_logg, a.dim_logr >= 0 ? r_ : a.val_logr,
                                   a.dim_vrot >= 0 ? v_ : a.val_vrot, a.plx_dim >= 0 ? d_ : 1.0);
    lp = (lp == -INFINITY || add == -INFINITY) ? -INFINITY : lp + add;
  }
  double* rec = W.spec + ((size_t)c * 2 + o) * kSpecStride;
  if (live && act) { rec[l] = up; rec[kSpecLanes + l] = vp; }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = l + kSpecLanes * j;
    const double vs = __shfl(vp, gb + (col_src[j] >= 0 ? col_src[j] : 0));
    if (live && col < ncols) rec[2 * kSpecLanes + col] = col_src[j] >= 0 ? vs : col_val[j];
  }
  if (live && l == 0) {
    rec[2 * kSpecLanes + kSpecCols] = lp;
    int* m = reinterpret_cast<int*>(rec + 2 * kSpecLanes + kSpecCols + 1);
    m[0] = in ? 1 : 0; m[1] = skipped;
  }
}

// The tail's half: settle the pending proposal of chain c with its likelihood, take the next proposal from the record of the outcome.
__device__ __forceinline__ void rwalk_settle_spec(const WalkState& W, int c, int lane, double lnl_p) {
  const int nd = W.nd, ncols = W.ncols;
  const bool act = lane < nd;
  const int dl = act ? lane : 0, colc = lane < ncols ? lane : 0;
  const size_t off = (size_t)c * nd + dl;
  const double* rA = W.spec + (size_t)c * 2 * kSpecStride;
  const double* rB = rA + kSpecStride;
  // every load before the first store
  const int was_in = W.inside[c];
  const double lpr = W.lnprior_prop[c];
  const double u_p = W.u_prop[off], v_p = W.v_prop[off];
  const int ncall0 = W.ncall[c], nacc0 = W.nacc[c], nredraw0 = W.nredraw ? W.nredraw[c] : 0;
  const double upA = rA[dl], vpA = rA[kSpecLanes + dl], rowA = rA[2 * kSpecLanes + colc], lpA = rA[2 * kSpecLanes + kSpecCols];
  const double upB = rB[dl], vpB = rB[kSpecLanes + dl], rowB = rB[2 * kSpecLanes + colc], lpB = rB[2 * kSpecLanes + kSpecCols];
  const int* mA = reinterpret_cast<const int*>(rA + 2 * kSpecLanes + kSpecCols + 1);
  const int* mB = reinterpret_cast<const int*>(rB + 2 * kSpecLanes + kSpecCols + 1);
  const int inA = mA[0], skA = mA[1], inB = mB[0], skB = mB[1];
  const double lp = (lpr == -INFINITY) ? -INFINITY : lpr + lnl_p;
  const bool accept = was_in && (lp > W.loglstar);                // false for NaN
  if (was_in) {
    if (accept && act) { W.u[off] = u_p; W.v[off] = v_p; }
    if (lane == 0) {
      W.ncall[c] = ncall0 + 1;
      if (accept) { W.lnprob[c] = lp; W.nacc[c] = nacc0 + 1; }
    }
  }
  if (act) { W.u_prop[off] = accept ? upB : upA; W.v_prop[off] = accept ? vpB : vpA; }
  if (lane == 0) {
    W.inside[c] = accept ? inB : inA;
    W.lnprior_prop[c] = accept ? lpB : lpA;
    if (W.nredraw) W.nredraw[c] = nredraw0 + (accept ? skB : skA);
  }
  if (lane < ncols) W.rows[(size_t)c * ncols + lane] = accept ? rowB : rowA;
}
#endif
