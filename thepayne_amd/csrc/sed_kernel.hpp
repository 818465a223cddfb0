// sed_kernel.hpp -- photometry: FastPayneSEDPredict.sed (Payne/predict/predictsed.py:75-103,
// photANN.py:95-131, highred.py:4-25), one wave per (candidate, filter).  Included once by payne_hip.hip.
#pragma once

#include "sed_core.hpp"

__global__ void __launch_bounds__(64) payne_sed_kernel(PhotTables P, const double* in, int ld, int mode, int off,
                                                       int photscale, double* mags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* a1 = reinterpret_cast<double*>(smem);
  double* a2 = a1 + P.H;
  const int f = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const SedRow sr = sed_row(P, in + (size_t)b * ld, mode, off, photscale);
  const double* xs = sr.xs;
  const int H = P.H;
  for (int h = lane; h < H; h += 64) {
    double z = (double)P.b1[f * H + h];
    const float* w = P.w1 + (size_t)(f * H + h) * 6;
#pragma unroll
    for (int d = 0; d < 6; ++d) z += (double)w[d] * xs[d];
    a1[h] = 1.0 / (1.0 + exp(-z));
  }
  __syncthreads();
  for (int h = lane; h < H; h += 64) {
    double z = (double)P.b2[f * H + h];
    const float* w = P.w2t + (size_t)f * H * H + h;
    // sixteen weights requested at a time (a load -> fma loop pays one L2 round trip per k: 64 of them; all 64 at once, with
    // every other weight of the lane, measured slower: 14.4 us against 9.1 -- the registers cost the waves that hide the rest)
    int k = 0;
    for (; k + 16 <= H; k += 16) {
      float wv[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) wv[q] = w[(size_t)(k + q) * H];
#pragma unroll
      for (int q = 0; q < 16; ++q) z += (double)wv[q] * a1[k + q];
    }
    for (; k < H; ++k) z += (double)w[(size_t)k * H] * a1[k];
    a2[h] = 1.0 / (1.0 + exp(-z));
  }
  __syncthreads();
  double part = 0.0;
  for (int h = lane; h < H; h += 64) part += (double)P.w3[f * H + h] * a2[h];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o);
  if (lane == 0) mags[(size_t)b * P.F + f] = sed_mag(P, sr, f, mode, part + (double)P.b3[f]);
}
