// sed_core.hpp -- photometry: FastPayneSEDPredict.sed (Payne/predict/predictsed.py:75-103, photANN.py:95-131,
// highred.py:4-25) as device functions shared by the two forms that ship:
//   * payne_sed_kernel (sed_kernel.hpp): one wave per (candidate, filter) -- the stand-alone entry points
//     (payne_sed_batch, payne_bc_batch) and likelihoods without a hidden-layer launch;
//   * sed_tile (below): one 256-thread workgroup per (filter, block of 64 candidates), run by workgroups appended to the
//     hidden-layer launch on compute units its 160 GEMM tiles leave idle -- the joint likelihood's photometry then costs
//     no launch of its own (C3: 9.1 us as a launch between the dense kernels and the post kernel).
// Arithmetic: fp64 on fp32 weights, as numpy promotes them (photANN.py:125-131).
#pragma once
#include <type_traits>

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
  // predictsed.py:84 evaluates 10**logt; for a theta row logt IS log10(Teff) (genmod.py:124,172), and Teff differs from
  // 10**log10(Teff) by an ulp or two (1e-15 of the encoded label): the row's own value is used, not a pow() per candidate
  double x[6] = {mode == 0 ? pow(10.0, s.logt) : r[0], logg, feh, afe, s.av, s.rv};
  s.hi = (mode != 2) && !(s.av < 5.0);                            // predictsed.py:86-90
  if (s.hi) { x[4] = 0.0; x[5] = 3.1; }
#pragma unroll
  for (int d = 0; d < 6; ++d) s.xs[d] = (x[d] - P.xmin[d]) / P.xden[d];
  return s;
}
// bolometric correction of filter f -> magnitude (predictsed.py:92-103, highred.py:19-25).  The magnitude formulae are
// (a1 - BC) + a2 with a1, a2 functions of the row alone (sed_mag_terms: same association as the expressions they come from).
__device__ __forceinline__ void sed_mag_terms(const SedRow& s, double& a1, double& a2) {
  const double nan = __builtin_nan("");
  if (!(s.logl != s.logl) && !(s.dist != s.dist)) { a1 = -2.5 * s.logl + 4.74; a2 = 5.0 * log10(s.dist) - 5.0; }
  else if (!(s.logA != s.logA)) { a1 = 5.0 * s.logA - 10.0 * (s.logt - log10(5770.0)) - 0.26; a2 = 0.0; }
  else { a1 = nan; a2 = nan; }
}
__device__ __forceinline__ double sed_bc_hi(const PhotTables& P, const SedRow& s, int f, double BC) {
  if (s.hi) {
    const double* c = P.hiav ? P.hiav + 5 * f : nullptr;
    const double offv = c ? (c[0] + c[1] * s.av * (c[2] + c[3] * s.rv + c[4] * (s.rv * s.rv))) : __builtin_nan("");
    BC = BC - offv;
  }
  return BC;
}
__device__ __forceinline__ double sed_mag(const PhotTables& P, const SedRow& s, int f, int mode, double BC) {
  BC = sed_bc_hi(P, s, f, BC);
  if (mode == 2) return BC;
  double a1, a2;
  sed_mag_terms(s, a1, a2);
  return (a1 - BC) + a2;
}

// ---- tile form -------------------------------------------------------------------------------------------
// One workgroup of 256 threads: filter f, candidates c0 .. c0 + cb - 1 of a theta batch (mode 1), cb <= 48.  Both hidden
// layers run on the fp64 matrix instruction (v_mfma_f64_16x16x4_f64: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// D[i = (lane >> 4) + 4 v][j = lane & 15], v = 0..3): candidates are the rows (up to three 16-row tiles), the layer's outputs
// the columns (wave w owns the 16-column tiles w, w + 4, ..), fp32 weights promoted on the way into the operand as numpy
// promotes them.  Every global value the tile needs is requested in one batch at its start; a single wave issues an
// instruction every ~5 cycles whatever it is, so sixteen fp64 fma per lane cost what ONE matrix instruction costs.
//   LDS: X [48][kSedPitchA] f64 (encoded labels, then layer-1 outputs, then layer-2 outputs) | W2 [H][kSedPitchW] f32 ([k][h])
//        | W1 [8][kSedPitchW] f32 ([d][h], rows 6, 7 zero) | b1, b2, w3 [H] f32
constexpr int kSedCandsMax = 48;
constexpr int kSedMaxH = 64;
constexpr int kSedPitchA = kSedMaxH + 2;      // doubles: row stride 132 dwords = 4 banks (mod 64): the 16 rows of an A fragment hit 16 bank quads
constexpr int kSedPitchW = kSedMaxH + 16;     // floats: k rows 80 dwords apart = 16 banks (mod 32): the k = 0..3 rows of a B fragment do not collide
__host__ __device__ constexpr size_t sed_tile_lds_bytes() {
  return (size_t)kSedCandsMax * kSedPitchA * 8 + (size_t)(kSedMaxH + 8) * kSedPitchW * 4 + 3 * kSedMaxH * 4 + (size_t)6 * kSedCandsMax * 8;
}
__host__ __device__ inline bool sed_tile_ok(int H) { return H >= 16 && H <= kSedMaxH && (H & 15) == 0; }

// 1 / (1 + exp(-z)) in fp64 (photANN.py:125-131 through numpy) for N values at once: exp by 2^n * 2^f (|f| <= 1/2, degree-13
// Taylor polynomial of exp(f ln 2): truncation < 5e-18, n ln 2 subtracted from -z in two parts), the quotient by v_rcp_f64 +
// two Newton steps; relative error ~3e-16.  NaN stays NaN, z -> -inf gives 0, +inf gives 1.  Written step by step ACROSS
// the N values: one value's chain is ~35 dependent fp64 instructions (and libm's exp + an IEEE division ~3x that); a wave
// alone on its SIMD runs such a chain at its latency, N chains side by side at its issue rate (measured in the tile below:
// twelve sigmoids one after the other 5 000 cycles).
template <int N>
__device__ __forceinline__ void sed_sigmoid_n(double* z) {
  double n[N], g[N], p[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double t = z[i] * -1.4426950408889634;                        // -z log2(e)
    const double tc = (t > 1000.0) ? 1000.0 : ((t < -1000.0) ? -1000.0 : t);   // (comparisons are false for NaN: it passes)
    n[i] = __builtin_rint(tc);
    const double zz = (tc == t) ? z[i] : tc * -0.6931471805599453;  // z as clamped (t only picks n: its rounding must not reach g)
    g[i] = fma(n[i], -0.6931471803691238, -zz);                   // -z - n ln2, ln2 in two parts (Cody-Waite): |g| <= 0.3466 + 1e-13
  }
#pragma unroll
  for (int i = 0; i < N; ++i) g[i] = fma(n[i], -1.9082149292705877e-10, g[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) p[i] = fma(1.6059043836821613e-10, g[i], 2.08767569878681e-09);   // 1/13!, 1/12!
  constexpr double C[11] = {2.505210838544172e-08, 2.755731922398589e-07, 2.7557319223985893e-06, 2.48015873015873e-05,
                            0.0001984126984126984, 0.001388888888888889, 0.008333333333333333, 0.041666666666666664,
                            0.16666666666666666, 0.5, 1.0};       // 1/11! .. 1/1!
#pragma unroll
  for (int k = 0; k < 11; ++k)
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = fma(p[i], g[i], C[k]);
#pragma unroll
  for (int i = 0; i < N; ++i) p[i] = fma(p[i], g[i], 1.0);
  double d[N], r[N];
#pragma unroll
  for (int i = 0; i < N; ++i) d[i] = 1.0 + __builtin_ldexp(p[i], (int)n[i]);   // 1 + exp(-z), exp in [2^-1000, 2^1000] (NaN stays NaN)
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = __builtin_amdgcn_rcp(d[i]);
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = fma(fma(-d[i], r[i], 1.0), r[i], r[i]);
#pragma unroll
  for (int i = 0; i < N; ++i) z[i] = r[i];
}
__device__ __forceinline__ double sed_sigmoid(double z) { sed_sigmoid_n<1>(&z); return z; }

typedef double sed_d4 __attribute__((ext_vector_type(4)));

// Row scalars of the tile form without the logarithms: the six encoded labels need none (mode 1: Teff itself).
__device__ __forceinline__ void sed_row_labels(const PhotTables& P, const double* r, int off, double* xs, double& av, bool& hi) {
  av = r[off + 2];
  hi = !(av < 5.0);                                             // predictsed.py:86-90
  const double x[6] = {r[0], r[1], r[2], r[3], hi ? 0.0 : av, 3.1};
#pragma unroll
  for (int d = 0; d < 6; ++d) xs[d] = (x[d] - P.xmin[d]) / P.xden[d];
}

__device__ inline void sed_tile(const PhotTables& P, const double* __restrict__ theta, int ld, int off, int photscale, int B,
                                int f, int c0, int cb, double* __restrict__ mags, unsigned char* smem,
                                unsigned long long* st = nullptr /* diagnostic build: cycle stamps of this tile */) {
#define SED_STAMP(k) do { if (st && threadIdx.x == 0) st[k] = __builtin_amdgcn_s_memtime(); } while (0)
  SED_STAMP(0);
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int H = P.H, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* X = reinterpret_cast<double*>(smem);                              // [48][kSedPitchA]
  float* W2 = reinterpret_cast<float*>(X + kSedCandsMax * kSedPitchA);      // [H][kSedPitchW]
  float* W1 = W2 + kSedMaxH * kSedPitchW;                                   // [8][kSedPitchW]
  float* Bv = W1 + 8 * kSedPitchW;                                          // b1 | b2 | w3, kSedMaxH each
  double* Mt = reinterpret_cast<double*>(Bv + 3 * kSedMaxH);                // [48][2] magnitude terms | [4][48] layer-3 partials
  double* Pt = Mt + 2 * kSedCandsMax;
  // ---- requests: the W2 tile (H*H/4 16-byte pieces), W1, the three vectors, this thread's theta row
  const f4* w2g = reinterpret_cast<const f4*>(P.w2t + (size_t)f * H * H);
  const int n4 = (H * H) >> 2, hq4 = H >> 2;
  constexpr int WPT = kSedMaxH * kSedMaxH / 4 / 256;                        // 4 pieces per thread at most
  f4 wreg[WPT];
#pragma unroll
  for (int q = 0; q < WPT; ++q) { const int i = tid + 256 * q; wreg[q] = w2g[i < n4 ? i : n4 - 1]; }
  float w1reg[2], vreg = 0.f;                                               // W1 is [h][6] in memory: 6 H <= 384 values
#pragma unroll
  for (int q = 0; q < 2; ++q) { const int i = tid + 256 * q; w1reg[q] = P.w1[(size_t)f * H * 6 + (i < 6 * H ? i : 0)]; }
  if (tid < 3 * kSedMaxH) {
    const int which = tid / kSedMaxH, h = tid - which * kSedMaxH;
    const float* src = which == 0 ? P.b1 : (which == 1 ? P.b2 : P.w3);
    vreg = src[f * H + (h < H ? h : 0)];
  }
  const float b3 = P.b3[f];
  // rows: wave 0 encodes the labels (what the first layer waits for), wave 1 works out the magnitude formula's terms
  // (two fp64 logarithms) for the same rows meanwhile; both read the row themselves
  const int rt = tid & 63;
  const int c = c0 + rt;
  const bool rowt = rt < kSedCandsMax && wave < 2, live = rt < cb && c < B;
  const double* rowp = theta + (size_t)(live ? c : (c0 < B ? c0 : B - 1)) * ld;
  // (wave 1's terms -- a log10 or two a row -- are worked out HERE, while the weight tile is still on its way: behind the
  //  operands' barrier they sat in front of the first layer's own barrier, 2 700 of that layer's 5 500 cycles)
  double xs[6], av = 0.0; bool hi = false;
  double mt1 = 0.0, mt2 = 0.0;
  if (rowt && wave == 0) sed_row_labels(P, rowp, off, xs, av, hi);
  if (rowt && wave == 1) {
    const SedRow s = sed_row(P, rowp, 1, off, photscale);
    sed_mag_terms(s, mt1, mt2);
  }
  SED_STAMP(1);
  // ---- commit to LDS
  if (rowt && wave == 0) {
#pragma unroll
    for (int d = 0; d < 6; ++d) X[rt * kSedPitchA + d] = xs[d];
    X[rt * kSedPitchA + 6] = 0.0; X[rt * kSedPitchA + 7] = 0.0;             // k padding of the first layer (K = 6 -> 8)
  }
#pragma unroll
  for (int q = 0; q < WPT; ++q) {
    const int i = tid + 256 * q;
    if (i < n4) { const int k = i / hq4, h4 = i - k * hq4; *reinterpret_cast<f4*>(W2 + k * kSedPitchW + 4 * h4) = wreg[q]; }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) { const int i = tid + 256 * q; if (i < 6 * H) { const int h = i / 6, d = i - 6 * h; W1[d * kSedPitchW + h] = w1reg[q]; } }
  if (tid < 2 * kSedMaxH) W1[(6 + tid / kSedMaxH) * kSedPitchW + (tid % kSedMaxH)] = 0.f;
  if (tid < 3 * kSedMaxH) Bv[tid] = vreg;
  if (rowt && wave == 1) { Mt[2 * rt] = mt1; Mt[2 * rt + 1] = mt2; }
  __syncthreads();
  SED_STAMP(3);
  const int mt_n = (cb + 15) >> 4, nt_n = H >> 4;                           // row tiles (<= 3), column tiles (<= 4)
  const int li = lane & 15, lk = lane >> 4;
  // one layer: X[.][0..K) -> sigmoid(X W + b) back into X[.][0..H); every wave reads all of X before anyone writes.
  // H <= 64: at most four 16-column tiles, wave w owns tile w (or none).  Every operand fragment of the layer is requested
  // from LDS before the first matrix instruction (K/4 <= 16 steps: 64 doubles a lane).
  const bool on = wave < nt_n;
  const int nt = on ? wave : 0;
  // (MT = the tile's row tiles, a compile-time copy per count: a 16-candidate tile reads a third of the fragments and runs four
  //  sigmoid chains a lane instead of twelve -- as one body sized for 48 candidates every tile paid for three)
  auto layer = [&](auto mtc, const float* Wl, int K, const float* bias) {
    constexpr int MT = decltype(mtc)::value;
    sed_d4 acc[MT];
    const double bz = (double)bias[16 * nt + li];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (sed_d4){bz, bz, bz, bz};
    if (on) {
      constexpr int KS = kSedMaxH / 4;
      float bfr[KS]; double afr[MT][KS];
      const int ks_n = K >> 2;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int k0 = 4 * (ks < ks_n ? ks : 0);
        bfr[ks] = Wl[(k0 + lk) * kSedPitchW + 16 * nt + li];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) afr[mt][ks] = X[(16 * mt + li) * kSedPitchA + k0 + lk];
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks < ks_n) {
          const double bfrag = (double)bfr[ks];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[mt][ks], bfrag, acc[mt], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (on) {
      double y[4 * MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) y[4 * mt + v] = acc[mt][v];
      sed_sigmoid_n<4 * MT>(y);                                             // 4 MT chains side by side
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int v = 0; v < 4; ++v) X[(16 * mt + lk + 4 * v) * kSedPitchA + 16 * nt + li] = y[4 * mt + v];
      }
    }
    __syncthreads();
  };
  auto layers = [&](auto mtc) {
    layer(mtc, W1, 8, Bv);                                                  // photANN.py:127
    SED_STAMP(4);
    layer(mtc, W2, H, Bv + kSedMaxH);                                       // :128
  };
  if (mt_n == 1) layers(std::integral_constant<int, 1>{});
  else if (mt_n == 2) layers(std::integral_constant<int, 2>{});
  else layers(std::integral_constant<int, 3>{});
  SED_STAMP(6);
  // ---- layer 3 (:129-131): wave w sums the outputs h = 16 w .. 16 w + 15 of row `rt`, wave 0 adds the four partials
  {
    double part = 0.0;
    if (rt < kSedCandsMax && on) {
#pragma unroll
      for (int q = 0; q < 16; ++q) part = fma((double)Bv[2 * kSedMaxH + 16 * wave + q], X[rt * kSedPitchA + 16 * wave + q], part);
    }
    if (rt < kSedCandsMax) Pt[wave * kSedCandsMax + rt] = part;
  }
  __syncthreads();
  if (wave == 0 && live) {
    double bc = (double)b3;
    for (int w = 0; w < 4; ++w) bc += Pt[w * kSedCandsMax + rt];
    if (hi) {                                                               // highred.py:19-25
      const double* cf = P.hiav ? P.hiav + 5 * f : nullptr;
      const double rv = 3.1;
      const double offv = cf ? (cf[0] + cf[1] * av * (cf[2] + cf[3] * rv + cf[4] * (rv * rv))) : __builtin_nan("");
      bc = bc - offv;
    }
    mags[(size_t)c * P.F + f] = (Mt[2 * rt] - bc) + Mt[2 * rt + 1];
  }
  SED_STAMP(5);
#undef SED_STAMP
}
