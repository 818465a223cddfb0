// sed_core.hpp -- photometry: FastPayneSEDPredict.sed (Payne/predict/predictsed.py:75-103, photANN.py:95-131,
// highred.py:4-25) as device functions shared by the two forms that ship:
//   * payne_sed_kernel (sed_kernel.hpp): one wave per (candidate, filter) -- the stand-alone entry points
//     (payne_sed_batch, payne_bc_batch) and likelihoods without a hidden-layer launch;
//   * sed_tile (below): one 256-thread workgroup per (filter, block of 64 candidates), run by workgroups appended to the
//     hidden-layer launch on compute units its 160 GEMM tiles leave idle -- the joint likelihood's photometry then costs
//     no launch of its own (C3: 9.1 us as a launch between the dense kernels and the post kernel).
// Arithmetic: fp64 on fp32 weights, as numpy promotes them (photANN.py:125-131).
#pragma once

struct PhotTables {
  int F, H;
  const float *w1, *b1, *w2t, *b2, *w3, *b3;   // w2t: [F][k][h] (transposed for coalesced lanes)
  double xmin[6], xden[6];
  const double* hiav;                           // device [F][5] or null
};

// What one (candidate) row asks of every filter: the six encoded labels and the magnitude formula's scalars.
struct SedRow {
  double xs[6];                 // (x - xmin)/(xmax - xmin), photANN.py:118-120 (no -0.5)
  double logt, av, rv, logl, dist, logA;
  bool hi;                      // av >= 5: the nets are evaluated at av = 0, rv = 3.1 and highAv corrects (predictsed.py:86-90)
};
// mode 0: in = [logt,logg,feh,afe,av,rv,logl,dist,logA] (sed kwargs, NaN = absent)
// mode 1: in = theta row; phot block at column `off` = [logA | logR, Dist, Av, Rv]
// mode 2: in = [Teff,logg,feh,afe,av,rv]; output = the bolometric corrections themselves
//         (fastANN.eval, photANN.py:125-131: no high-Av branch, no magnitude formula)
__device__ __forceinline__ SedRow sed_row(const PhotTables& P, const double* r, int mode, int off, int photscale) {
  const double nan = __builtin_nan("");
  SedRow s;
  double logg, feh, afe;
  s.logl = nan; s.dist = nan; s.logA = nan;
  if (mode == 0) {
    s.logt = r[0]; logg = r[1]; feh = r[2]; afe = r[3]; s.av = r[4]; s.rv = r[5]; s.logl = r[6]; s.dist = r[7]; s.logA = r[8];
  } else if (mode == 2) {
    s.logt = nan; logg = r[1]; feh = r[2]; afe = r[3]; s.av = r[4]; s.rv = r[5];
  } else {
    s.logt = log10(r[0]); logg = r[1]; feh = r[2]; afe = r[3];    // genmod.py:124,172
    s.av = r[off + 2]; s.rv = 3.1;                                // Rv never honoured: likelihood.py:104-106
    if (photscale) s.logA = r[off];                               // genphot_scaled, genmod.py:157-187
    else { s.logl = 2.0 * r[off] + 4.0 * (s.logt - log10(5770.0)); s.dist = r[off + 1]; }   // genphot, genmod.py:126
  }
  double x[6] = {mode == 2 ? r[0] : pow(10.0, s.logt), logg, feh, afe, s.av, s.rv};   // predictsed.py:84
  s.hi = (mode != 2) && !(s.av < 5.0);                            // predictsed.py:86-90
  if (s.hi) { x[4] = 0.0; x[5] = 3.1; }
#pragma unroll
  for (int d = 0; d < 6; ++d) s.xs[d] = (x[d] - P.xmin[d]) / P.xden[d];
  return s;
}
// bolometric correction of filter f -> magnitude (predictsed.py:92-103, highred.py:19-25)
__device__ __forceinline__ double sed_mag(const PhotTables& P, const SedRow& s, int f, int mode, double BC) {
  const double nan = __builtin_nan("");
  if (s.hi) {
    const double* c = P.hiav ? P.hiav + 5 * f : nullptr;
    const double offv = c ? (c[0] + c[1] * s.av * (c[2] + c[3] * s.rv + c[4] * (s.rv * s.rv))) : nan;
    BC = BC - offv;
  }
  if (mode == 2) return BC;
  if (!(s.logl != s.logl) && !(s.dist != s.dist)) return -2.5 * s.logl + 4.74 - BC + (5.0 * log10(s.dist) - 5.0);
  if (!(s.logA != s.logA)) return 5.0 * s.logA - 10.0 * (s.logt - log10(5770.0)) - 0.26 - BC;
  return nan;
}

// ---- tile form -------------------------------------------------------------------------------------------
// One workgroup of 256 threads: filter f, candidates c0 .. c0 + 63 of a theta batch (mode 1).  The second layer
// ([64 x H] . [H x H], 86 % of the work) is register-tiled 4 outputs x 4 candidates per thread with both operands in LDS:
//   xs[64][6] fp64 | act[H][64] fp64 (layer-1 outputs, then layer-2 outputs) | w2[H][H] fp32 ([k][h])
constexpr int kSedCands = 64;
constexpr int kSedMaxH = 64;
__host__ __device__ constexpr size_t sed_tile_lds_bytes(int H) { return (size_t)kSedCands * 6 * 8 + (size_t)H * kSedCands * 8 + (size_t)H * H * 4; }
__host__ __device__ inline bool sed_tile_ok(int H) { return H >= 4 && H <= kSedMaxH && (H & 3) == 0; }

__device__ __forceinline__ double sed_sigmoid(double z) { return 1.0 / (1.0 + exp(-z)); }

__device__ inline void sed_tile(const PhotTables& P, const double* __restrict__ theta, int ld, int off, int photscale, int B,
                                int f, int c0, double* __restrict__ mags, unsigned char* smem) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int H = P.H, tid = threadIdx.x;
  double* xs = reinterpret_cast<double*>(smem);                 // [64][6]
  double* act = xs + kSedCands * 6;                             // [H][64]
  float* w2 = reinterpret_cast<float*>(act + (size_t)H * kSedCands);   // [k][h]
  // ---- everything this workgroup reads from memory, requested up front: the w2 tile (H*H/4 16-byte pieces), theta rows
  const f4* w2g = reinterpret_cast<const f4*>(P.w2t + (size_t)f * H * H);
  const int n4 = (H * H) >> 2;
  constexpr int WPT = (kSedMaxH * kSedMaxH / 4 + 255) / 256;    // pieces per thread: 4
  f4 wreg[WPT];
#pragma unroll
  for (int q = 0; q < WPT; ++q) { const int i = tid + 256 * q; wreg[q] = w2g[i < n4 ? i : n4 - 1]; }
  SedRow s{};
  const int c = c0 + tid;
  const bool live = tid < kSedCands && c < B;
  if (tid < kSedCands) {
    s = sed_row(P, theta + (size_t)(live ? c : (B - 1)) * ld, 1, off, photscale);
#pragma unroll
    for (int d = 0; d < 6; ++d) xs[tid * 6 + d] = s.xs[d];
  }
#pragma unroll
  for (int q = 0; q < WPT; ++q) { const int i = tid + 256 * q; if (i < n4) reinterpret_cast<f4*>(w2)[i] = wreg[q]; }
  __syncthreads();
  // ---- layer 1: act[k][c] = sigmoid(b1[k] + sum_d w1[k][d] xs[c][d]); thread -> candidate tid & 63, k = tid >> 6, + 4, ..
  {
    const int cc = tid & 63;
    double x6[6];
#pragma unroll
    for (int d = 0; d < 6; ++d) x6[d] = xs[cc * 6 + d];
    for (int k = tid >> 6; k < H; k += 4) {
      const float* w = P.w1 + (size_t)(f * H + k) * 6;
      double z = (double)P.b1[f * H + k];
#pragma unroll
      for (int d = 0; d < 6; ++d) z += (double)w[d] * x6[d];
      act[k * kSedCands + cc] = sed_sigmoid(z);
    }
  }
  __syncthreads();
  // ---- layer 2: thread -> outputs h = 4 hq .. + 3 (hq = tid % (H/4)), candidates 4 cq .. + 3: H/4 x 16 <= 256 thread tiles
  const int nhq = H >> 2, ntile = nhq * (kSedCands / 4);
  const bool on = tid < ntile;
  const int hq = on ? tid % nhq : 0, cq = on ? tid / nhq : 0;
  double z2[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double bz = (double)P.b2[f * H + 4 * hq + i];
#pragma unroll
    for (int j = 0; j < 4; ++j) z2[i][j] = bz;
  }
#pragma unroll 4
  for (int k = 0; k < H; ++k) {
    const f4 w = *reinterpret_cast<const f4*>(w2 + (size_t)k * H + 4 * hq);
    const d2 a01 = *reinterpret_cast<const d2*>(act + (size_t)k * kSedCands + 4 * cq);
    const d2 a23 = *reinterpret_cast<const d2*>(act + (size_t)k * kSedCands + 4 * cq + 2);
    const double wd[4] = {(double)w.x, (double)w.y, (double)w.z, (double)w.w};
    const double ad[4] = {a01.x, a01.y, a23.x, a23.y};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) z2[i][j] = fma(wd[i], ad[j], z2[i][j]);
  }
  __syncthreads();                                              // every read of layer 1's outputs is done: reuse `act`
  if (on) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) act[(size_t)(4 * hq + i) * kSedCands + 4 * cq + j] = sed_sigmoid(z2[i][j]);
  }
  __syncthreads();
  // ---- layer 3 + magnitude: thread c (< 64) sums over h
  if (tid < kSedCands) {
    double bc = (double)P.b3[f];
    for (int h = 0; h < H; ++h) bc += (double)P.w3[f * H + h] * act[(size_t)h * kSedCands + tid];
    if (live) mags[(size_t)c * P.F + f] = sed_mag(P, s, f, 1, bc);
  }
}
