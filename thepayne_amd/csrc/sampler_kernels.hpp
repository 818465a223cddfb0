// sampler_kernels.hpp -- the sampler step's kernels (device functions: sampler_core.hpp).  Included once by payne_hip.hip.
#pragma once
#include "sampler_core.hpp"

// transform only (mode 0) or transform + ln-prior + theta row (mode 1).
// (No private arrays and no reference to the by-value argument handed to an out-of-line function: the first version kept the
//  transformed point in a `double vv[PAYNE_MAX_DIM]` indexed by the loop counter and passed `sd.dims[d]` / `sd.adv` by reference to
//  helpers the compiler did not inline -- the whole 4 KB argument struct was copied to scratch memory, 4 400 bytes and 319 spilled
//  registers per thread.  The point is read back from `v`, which this thread has just written; the helpers are inlined.)
__global__ void __launch_bounds__(128) __attribute__((flatten)) payne_prior_kernel(SamplerDev sd, const double* u, int K, double* v, double* lnprior, double* rows, int mode) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= K) return;
  double* __restrict__ vc = v + (size_t)c * sd.ndim;
  double lp = 0.0;
  for (int d = 0; d < sd.ndim; ++d) {
    const payne_prior_dim dim = sd.dims[d];
    const double vd = prior_ppf(dim, sd.q0[d], sd.q1[d], u[(size_t)c * sd.ndim + d], sd.adv);
    vc[d] = vd;
    lp += prior_ln(dim, vd);
  }
  if (adv_any(sd.adv)) {
    const payne_adv_priors& a = sd.adv;
    const double add = adv_lnprior(a, a.dim_logg >= 0 ? vc[a.dim_logg] : a.val_logg, a.dim_logr >= 0 ? vc[a.dim_logr] : a.val_logr,
                                   a.dim_vrot >= 0 ? vc[a.dim_vrot] : a.val_vrot, a.plx_dim >= 0 ? vc[a.plx_dim] : 1.0);
    lp = (lp == -INFINITY || add == -INFINITY) ? -INFINITY : lp + add;
  }
  if (mode) {
    lnprior[c] = lp;
    double* row = rows + (size_t)c * sd.ncols;                    // the theta row by column (col_src / col_val: resolved at sampler creation)
    for (int q = 0; q < sd.ncols; ++q) { const int src = sd.col_src[q]; row[q] = src >= 0 ? vc[src] : sd.col_val[q]; }
  }
}
// the per-dimension constants of the transforms, by the device's own normcdf / expm1 / log (sampler creation)
__global__ void payne_prior_cache_kernel(SamplerDev sd, double* q) {
  const int d = threadIdx.x;
  if (d >= sd.ndim) return;
  prior_cache(sd.dims[d], q[d], q[PAYNE_MAX_DIM + d]);
}
// lnprob = lnprior + lnlike (-inf prior wins; NaN likelihood stays NaN)
__global__ void payne_lnprob_kernel(const double* lnprior, const double* lnl, int K, double* out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < K) out[c] = (lnprior[c] == -INFINITY) ? -INFINITY : lnprior[c] + lnl[c];
}

// One random-walk step for every chain, one wave per chain (rwalk_step_wave).  `propose` = 0 on the closing call.
__global__ void __launch_bounds__(256) payne_rwalk_kernel(SamplerDev sd, WalkState W, const double* lnl_prop, int step, int settle, int propose,
                                                             WalkTail* publish) {
  const int lane = threadIdx.x & 63;
  const int c = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  // the walk for the post kernel's tail (the steps that follow run there): written in stream order, by the launch that
  // opens the walk
  // (a queue launched without the host in between: the record carries the scale and threshold the turn kernel left)
  // (publish->sd: uploaded when the sampler was created)
  // (the two values read FIRST, together: behind the record's stores -- which may alias them -- each was a round trip of its own on the
  // wave that then walks chain 0, and the launch lasts as long as its slowest wave)
  if (publish && blockIdx.x == 0 && threadIdx.x == 0) {
    const double sc = walk_scale(W), ls = walk_lstar(W);
    publish->w = W; publish->w.scale = sc; publish->w.loglstar = ls; publish->w.dyn = nullptr;
  }
  if (c >= W.K) return;                                         // the whole wave leaves together
  rwalk_step_wave(sd, W, c, lane, lnl_prop[c], step, settle, propose);
}
