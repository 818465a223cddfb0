// sampler_kernels.hpp -- the sampler step's kernels (device functions: sampler_core.hpp).  Included once by payne_hip.hip.
#pragma once
#include "sampler_core.hpp"

// transform only (mode 0) or transform + ln-prior + theta row (mode 1)
__global__ void payne_prior_kernel(SamplerDev sd, const double* u, int K, double* v, double* lnprior, double* rows, int mode) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= K) return;
  double vv[PAYNE_MAX_DIM];
  double lp = 0.0;
  for (int d = 0; d < sd.ndim; ++d) {
    vv[d] = prior_ppf(sd.dims[d], sd.q0[d], sd.q1[d], u[(size_t)c * sd.ndim + d], sd.adv);
    v[(size_t)c * sd.ndim + d] = vv[d];
    lp += prior_ln(sd.dims[d], vv[d]);
  }
  if (adv_any(sd.adv)) {
    const payne_adv_priors& a = sd.adv;
    const double add = adv_lnprior(a, a.dim_logg >= 0 ? vv[a.dim_logg] : a.val_logg, a.dim_logr >= 0 ? vv[a.dim_logr] : a.val_logr,
                                   a.dim_vrot >= 0 ? vv[a.dim_vrot] : a.val_vrot, a.plx_dim >= 0 ? vv[a.plx_dim] : 1.0);
    lp = (lp == -INFINITY || add == -INFINITY) ? -INFINITY : lp + add;
  }
  if (mode) { lnprior[c] = lp; write_theta_row(sd, vv, rows + (size_t)c * sd.ncols); }
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
  if (publish && blockIdx.x == 0 && threadIdx.x == 0) { publish->sd = sd; publish->w = W; }
  if (c >= W.K) return;                                         // the whole wave leaves together
  rwalk_step_wave(sd, W, c, lane, lnl_prop[c], step, settle, propose);
}
