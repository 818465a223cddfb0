// sed_kernel.hpp -- photometry: FastPayneSEDPredict.sed (Payne/predict/predictsed.py:75-103,
// photANN.py:95-131, highred.py:4-25), one wave per (candidate, filter).  Included once by payne_hip.hip.
#pragma once

// ============================================================================
// photometric SED
// ============================================================================
struct PhotTables {
  int F, H;
  const float *w1, *b1, *w2t, *b2, *w3, *b3;   // w2t: [F][k][h] (transposed for coalesced lanes)
  double xmin[6], xden[6];
  const double* hiav;                           // device [F][5] or null
};

// mode 0: in = [logt,logg,feh,afe,av,rv,logl,dist,logA] (sed kwargs, NaN = absent)
// mode 1: in = theta row; phot block at column `off` = [logA | logR, Dist, Av, Rv]
// mode 2: in = [Teff,logg,feh,afe,av,rv]; output = the bolometric corrections themselves
//         (fastANN.eval, photANN.py:125-131: no high-Av branch, no magnitude formula)
__global__ void __launch_bounds__(64) payne_sed_kernel(PhotTables P, const double* in, int ld, int mode, int off,
                                                       int photscale, double* mags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* a1 = reinterpret_cast<double*>(smem);
  double* a2 = a1 + P.H;
  const int f = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const double* r = in + (size_t)b * ld;
  const double nan = __builtin_nan("");
  double logt, logg, feh, afe, av, rv, logl = nan, dist = nan, logA = nan;
  if (mode == 0) {
    logt = r[0]; logg = r[1]; feh = r[2]; afe = r[3]; av = r[4]; rv = r[5]; logl = r[6]; dist = r[7]; logA = r[8];
  } else if (mode == 2) {
    logt = nan; logg = r[1]; feh = r[2]; afe = r[3]; av = r[4]; rv = r[5];
  } else {
    logt = log10(r[0]); logg = r[1]; feh = r[2]; afe = r[3];      // genmod.py:124,172
    av = r[off + 2]; rv = 3.1;                                    // Rv never honoured: likelihood.py:104-106
    if (photscale) logA = r[off];                                 // genphot_scaled, genmod.py:157-187
    else { logl = 2.0 * r[off] + 4.0 * (logt - log10(5770.0)); dist = r[off + 1]; }   // genphot, genmod.py:126
  }
  double x[6] = {mode == 2 ? r[0] : pow(10.0, logt), logg, feh, afe, av, rv};   // predictsed.py:84
  const bool hi = (mode != 2) && !(av < 5.0);                     // predictsed.py:86-90
  if (hi) { x[4] = 0.0; x[5] = 3.1; }
  double xs[6];
#pragma unroll
  for (int d = 0; d < 6; ++d) xs[d] = (x[d] - P.xmin[d]) / P.xden[d];   // photANN.py:118-120 (no -0.5)
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
  if (lane == 0) {
    double BC = part + (double)P.b3[f];
    if (hi) {                                                     // highred.py:19-25
      const double* c = P.hiav ? P.hiav + 5 * f : nullptr;
      const double offv = c ? (c[0] + c[1] * av * (c[2] + c[3] * rv + c[4] * (rv * rv))) : nan;
      BC = BC - offv;
    }
    double m;
    if (mode == 2) m = BC;
    else if (!(logl != logl) && !(dist != dist)) m = -2.5 * logl + 4.74 - BC + (5.0 * log10(dist) - 5.0);
    else if (!(logA != logA)) m = 5.0 * logA - 10.0 * (logt - log10(5770.0)) - 0.26 - BC;
    else m = nan;
    mags[(size_t)b * P.F + f] = m;
  }
}
