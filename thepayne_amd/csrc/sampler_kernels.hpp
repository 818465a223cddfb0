// sampler_kernels.hpp -- the sampler step on the device: prior transform and ln-prior
// (Payne/fitting/prior.py:126-465), theta rows, random-walk proposals (one wave per chain).
// Included once by payne_hip.hip.
#pragma once

// ============================================================================
// device-side sampler step: prior transform, ln-prior, theta rows, random-walk proposals
// ============================================================================
struct SamplerDev {
  int ndim, ncols, nfixed;
  payne_prior_dim dims[PAYNE_MAX_DIM];
  int fixed_col[PAYNE_MAX_FIXED];
  double fixed_val[PAYNE_MAX_FIXED];
};

// unit cube -> parameter (Payne/fitting/prior.py:151-178, scipy.stats ppf's restated)
__device__ double prior_ppf(const payne_prior_dim& d, double u) {
  switch (d.kind) {
    case PAYNE_PRIOR_UNIFORM: {
      const double lo = fmin(d.p[0], d.p[1]), hi = fmax(d.p[0], d.p[1]);
      return (hi - lo) * u + lo;
    }
    case PAYNE_PRIOR_GAUSSIAN: return d.p[0] + d.p[1] * normcdfinv(u);
    case PAYNE_PRIOR_TGAUSSIAN: {
      const double a = (d.p[0] - d.p[2]) / d.p[3], b = (d.p[1] - d.p[2]) / d.p[3];
      double x;
      if (a > 0.0) {                    // both limits in the upper tail: work with survival functions
        const double sa = normcdf(-a), sb = normcdf(-b);
        x = -normcdfinv(sa - u * (sa - sb));
      } else {
        const double ca = normcdf(a), cb = normcdf(b);
        x = normcdfinv(ca + u * (cb - ca));
      }
      double v = d.p[2] + d.p[3] * x;
      if (!(v <= d.p[1])) v = (v != v) ? v : d.p[1];           // +inf (u = 1) -> hi, prior.py:165-166
      return v;
    }
    case PAYNE_PRIOR_EXP: return d.p[0] - d.p[1] * log1p(-u);
    case PAYNE_PRIOR_TEXP: {
      const double b = (d.p[1] - d.p[0]) / d.p[2];
      double v = d.p[0] - d.p[2] * log1p(u * expm1(-b));        // truncexpon.ppf
      if (!(v <= d.p[1])) v = (v != v) ? v : d.p[1];
      return v;
    }
    case PAYNE_PRIOR_LOGUNIFORM: return exp(log(d.p[0]) + u * (log(d.p[1]) - log(d.p[0])));
    default: return u;
  }
}
__device__ double prior_ln(const payne_prior_dim& d, double v) {
  double lp = 0.0;
  if (d.has_gauss) { const double z = v - d.g_mu; lp += -0.5 * ((z * z) / (d.g_sigma * d.g_sigma)); }
  if (d.has_box && ((v < d.box_lo) || (v > d.box_hi))) lp = -INFINITY;
  return lp;
}
// one theta row: NaN = absent, fixed values, then the sampled dimensions
__device__ void write_theta_row(const SamplerDev& sd, const double* v, double* row) {
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

// transform only (mode 0) or transform + ln-prior + theta row (mode 1)
__global__ void payne_prior_kernel(SamplerDev sd, const double* u, int K, double* v, double* lnprior, double* rows, int mode) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= K) return;
  double vv[PAYNE_MAX_DIM];
  double lp = 0.0;
  for (int d = 0; d < sd.ndim; ++d) {
    vv[d] = prior_ppf(sd.dims[d], u[(size_t)c * sd.ndim + d]);
    v[(size_t)c * sd.ndim + d] = vv[d];
    lp += prior_ln(sd.dims[d], vv[d]);
  }
  if (mode) { lnprior[c] = lp; write_theta_row(sd, vv, rows + (size_t)c * sd.ncols); }
}
// lnprob = lnprior + lnlike (-inf prior wins; NaN likelihood stays NaN)
__global__ void payne_lnprob_kernel(const double* lnprior, const double* lnl, int K, double* out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < K) out[c] = (lnprior[c] == -INFINITY) ? -INFINITY : lnprior[c] + lnl[c];
}

// One random-walk step for every chain: first settle the previous proposal (accept iff inside the
// cube and lnprob > loglstar), then draw the next one.  `propose` = 0 on the closing call.
// ONE WAVE PER CHAIN, lane d = sampled dimension d: the inverse CDFs (the expensive part: normcdf /
// normcdfinv chains in fp64) of the dimensions run side by side, the ellipsoid step is a shuffle
// matvec, sums are wave reductions.  (One thread per chain spent 14 us per step in a ~3000-instruction
// dependent fp64 chain; the step sits between two likelihood batches, nothing overlaps it.)
__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}
__global__ void __launch_bounds__(256) payne_rwalk_kernel(SamplerDev sd, int K, double* u, double* v, double* lnprob, int* nacc, int* ncall,
                                   double* u_prop, double* v_prop, double* lnprior_prop, int* inside,
                                   const double* lnl_prop, double* rows, const double* axes, const int* ell, double scale,
                                   double loglstar, unsigned long long seed, int step, int settle, int propose) {
  const int lane = threadIdx.x & 63;
  const int c = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (c >= K) return;                                           // the whole wave leaves together
  const int nd = sd.ndim;
  const bool act = lane < nd;
  const int dl = act ? lane : 0;
  const size_t off = (size_t)c * nd + dl;
  double uc = u[off];
  if (settle && inside[c]) {
    const double lpr = lnprior_prop[c];
    const double lp = (lpr == -INFINITY) ? -INFINITY : lpr + lnl_prop[c];
    const bool accept = lp > loglstar;                          // false for NaN
    if (accept && act) { uc = u_prop[off]; u[off] = uc; v[off] = v_prop[off]; }
    if (lane == 0) {
      ncall[c] += 1;
      if (accept) { lnprob[c] = lp; nacc[c] += 1; }
    }
  }
  if (!propose) return;
  // z uniform in the unit ball: normal direction (one Box-Muller cosine per lane), radius U^(1/n)
  double z = 0.0;
  if (act) {
    const double a = u01(seed, c, step, 2 * lane), b = u01(seed, c, step, 2 * lane + 1);
    z = sqrt(-2.0 * log(a)) * cos(6.283185307179586 * b);
  }
  const double n2 = wave_sum(z * z);
  const double rad = pow(u01(seed, c, step, 128), 1.0 / (double)nd) / sqrt(n2);
  const double* ax = axes + (ell ? (size_t)ell[c] * nd * nd : 0);   // this chain's ellipsoid (bound='multi')
  double sdot = 0.0;
  for (int e = 0; e < nd; ++e) {
    const double ze = __shfl(z, e);
    sdot = fma(ax[dl * nd + e], ze, sdot);
  }
  const double up = uc + scale * rad * sdot;
  const bool in = __ballot(act && !((up > 0.0) && (up < 1.0))) == 0ull;
  const payne_prior_dim dim = sd.dims[dl];
  const double vp = in ? prior_ppf(dim, up) : v[off];           // outside: a harmless valid row
  const double lp = wave_sum(act ? prior_ln(dim, vp) : 0.0);
  if (act) { u_prop[off] = up; v_prop[off] = vp; }
  if (lane == 0) { inside[c] = in ? 1 : 0; lnprior_prop[c] = lp; }
  // theta row, lane = column: NaN = absent, fixed values, then the sampled dimensions
  double val = __builtin_nan("");
  for (int i = 0; i < sd.nfixed; ++i) val = (sd.fixed_col[i] == lane) ? sd.fixed_val[i] : val;
  for (int d = 0; d < nd; ++d) {
    const double vd = __shfl(vp, d);
    val = (sd.dims[d].theta_col == lane) ? vd : val;
  }
  if (lane < sd.ncols) rows[(size_t)c * sd.ncols + lane] = val;
}
