// post_seq.hpp -- the order of the phases of post_core.hpp for one candidate.
// `Ex` supplies "run this phase on every thread of the workgroup, then barrier"
// (`par`) and the merge of the mask bounds (`imin`/`imax`): DevExec in payne_hip.hip
// (threadIdx + __syncthreads + LDS atomics), HostExec in tests/emul/cpu_emul.cpp.
#pragma once
#include "post_core.hpp"

#ifdef __HIPCC__
#define PAYNE_SEQ __device__ __forceinline__
#else
#define PAYNE_SEQ inline
#endif

namespace payne {

// M-point complex FFT by ping-pong between a and b; returns where the result is.
template <class Ex>
PAYNE_SEQ c32* fft_run(Ex& ex, c32* a, c32* b, int M, const c32* tw, int tw_n, bool conj_last) {
  c32 *src = a, *dst = b;
  int p = 1;
  while (p < M) {
    const int R = pass_radix(M, p);
    const bool cj = conj_last && (p * R == M);
    if (R == 8) ex.par([&](int t, int n) { fft_pass<8>(t, n, src, dst, M, p, tw, tw_n, cj); });
    else if (R == 4) ex.par([&](int t, int n) { fft_pass<4>(t, n, src, dst, M, p, tw, tw_n, cj); });
    else ex.par([&](int t, int n) { fft_pass<2>(t, n, src, dst, M, p, tw, tw_n, cj); });
    c32* t_ = src; src = dst; dst = t_;
    p *= R;
  }
  return src;
}

// out_stage: -1 chi^2 only | 0 raw ANN | 1 after vsini | 2 getspec on obs grid | 3 genspec (x blaze)
template <class Ex>
PAYNE_SEQ void run_candidate(Ex& ex, const PostTables& T, const c32* tw, const double* th, double instr_factor,
                             const float* raw, float* bufA, float* bufB, CandState& S, double* red,
                             float* out, int out_stage, double* chi2_out) {
  ex.par([&](int t, int n) {
    phase_setup(t, n, T, th, instr_factor, S);
    phase_load(t, n, T, raw, bufA);
  });
  float* spec = bufA;
  float* work = bufB;
  if (out_stage == 0) {
    ex.par([&](int t, int n) { for (int i = t; i < T.npix; i += n) out[i] = spec[i] + kBase; });
    return;
  }
  if (S.do_rot) {
    ex.par([&](int t, int n) { phase_rot_resample(t, n, T, spec, work); });
    const int M = T.n1 / 2;
    c32* z = fft_run(ex, (c32*)work, (c32*)spec, M, tw, T.nmax, false);
    ex.par([&](int t, int n) { rfft_taper_phase<true>(t, n, z, M, tw, T.nmax, S.vs_a, T.vs_val, T.vs_tab); });
    c32* zo = ((float*)z == bufA) ? (c32*)bufB : (c32*)bufA;
    c32* y = fft_run(ex, z, zo, M, tw, T.nmax, true);
    float* conv = (float*)y;
    float* dst = (conv == bufA) ? bufB : bufA;
    if (T.rot_identity) { float* t_ = dst; dst = conv; conv = t_; }          // conv IS on the ANN grid
    else ex.par([&](int t, int n) { phase_rot_back(t, n, T, conv, dst); });
    ex.par([&](int t, int n) { phase_rot_edges(t, T, dst); });
    spec = dst;
    work = conv;
  }
  if (out_stage == 1) {
    ex.par([&](int t, int n) { for (int i = t; i < T.npix; i += n) out[i] = spec[i] + kBase; });
    return;
  }
  const float* on_grid = spec;
  if (S.do_smooth) {
    ex.par([&](int t, int n) { phase_mask_scan(t, n, T, th, instr_factor, S); });
    ex.par([&](int t, int n) { phase_window(t, T, S); });
    if (!S.bad) {
      ex.par([&](int t, int n) { phase_R_resample(t, n, T, S, spec, work); });
      const int M = S.n2 / 2;
      c32* z = fft_run(ex, (c32*)work, (c32*)spec, M, tw, T.nmax, false);
      ex.par([&](int t, int n) { rfft_taper_phase<false>(t, n, z, M, tw, T.nmax, S.g_a, S.g_val, nullptr); });
      c32* zo = ((float*)z == bufA) ? (c32*)bufB : (c32*)bufA;
      on_grid = (const float*)fft_run(ex, z, zo, M, tw, T.nmax, true);
    }
  }
  ex.par([&](int t, int n) { red[t] = phase_obs(t, n, T, S, on_grid, out, out_stage); });
  ex.par([&](int t, int n) {
    const int chunk = (n + 15) / 16;
    if (t < 16) {
      double s = 0.0;
      for (int i = t * chunk; i < (t + 1) * chunk && i < n; ++i) s += red[i];
      red[n + t] = s;
    }
  });
  ex.par([&](int t, int n) {
    if (t == 0) {
      double s = 0.0;
      for (int i = 0; i < 16; ++i) s += red[n + i];
      *chi2_out = s;
    }
  });
}

}  // namespace payne
