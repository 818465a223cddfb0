// post_core.hpp -- per-candidate spectrum pipeline of the likelihood hot path,
// written as barrier-separated PHASES so that the same source runs
//   * on gfx950 inside post_kernel (one 256-thread workgroup per candidate, all
//     spectra resident in LDS), and
//   * on the host (csrc/cpu_emul.cpp: phases executed for tid = 0..NT-1 in turn)
//     as a bit-faithful emulation used by the CPU tests and sanitizer builds.
//
// What it computes (reference: /root/reference, restated in oracle/payne_oracle.py):
//   raw ANN spectrum  ->  [vsini FFT broadening, Payne/predict/ystpred.py:211-224,
//   Payne/utils/smoothing.py:293-312,610-629]  ->  Doppler shift (ystpred.py:226-232)
//   ->  data-dependent mask + log-lambda pow-2 resample + Gaussian FFT smoothing to the
//   instrument R + interpolation onto the observed grid (smoothing.py:103-115,131-169,
//   252-291,588-608,631-668)  ->  [Chebyshev blaze, Payne/fitting/fitutils.py:11-20]
//   ->  chi^2 (Payne/fitting/likelihood.py:95-97).
//
// Numerics: flux arithmetic is fp32 on a spectrum shifted by -1 (normalised spectra
// live near 1; every linear stage has unit DC gain, so f -> f-1 commutes with the
// pipeline and buys ~20x smaller rounding error).  Wavelength arithmetic is fp64 in
// ln(lambda): the interpolation weight (x-x_k)/(x_{k+1}-x_k) is evaluated as
// expm1(u)/expm1(v) ~ (u/v)(1+(u-v)/2) from fp64 differences of logs, so there is no
// per-pixel exp() and no fp32 wavelength anywhere.  Fourier tapers
// are evaluated in fp64 (vsini taper cancels catastrophically in fp32, 7.3-2).
#pragma once
#include <math.h>
#include <stdint.h>

#ifdef __HIPCC__
#define PAYNE_HD __host__ __device__ __forceinline__
#else
#define PAYNE_HD inline
#endif

namespace payne {

constexpr double kCkms = 2.998e5;            // smoothing.py:16
constexpr double kCDoppler = 299792.458;     // ystpred.py:11-12
constexpr double kPi = 3.141592653589793;
constexpr float kBase = 1.0f;                // the flux shift

struct c32 { float x, y; };
PAYNE_HD c32 cmul(c32 a, c32 b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
PAYNE_HD c32 cadd(c32 a, c32 b) { return {a.x + b.x, a.y + b.y}; }
PAYNE_HD c32 csub(c32 a, c32 b) { return {a.x - b.x, a.y - b.y}; }
PAYNE_HD c32 cconj(c32 a) { return {a.x, -a.y}; }
PAYNE_HD c32 cscale(c32 a, float s) { return {a.x * s, a.y * s}; }
PAYNE_HD c32 cmul_negi(c32 a) { return {a.y, -a.x}; }   // a * (-i)
PAYNE_HD c32 cmul_posi(c32 a) { return {-a.y, a.x}; }   // a * (+i)

// ---------------------------------------------------------------------------
// Static (per-context) tables, all device-resident; built once on the host.
// ---------------------------------------------------------------------------
struct PostTables {
  int npix;            // ANN pixels
  int nobs;            // observed pixels (0: no obs grid bound)
  int n1;              // pow2ceil(npix): vsini FFT length
  int nmax;            // twiddle table length (= largest real FFT length)
  const double* lnlam;     // [npix]  ln(lambda_ANN)
  const double* lam;       // [npix]  lambda_ANN (exact mask test)
  const c32* tw;           // [nmax]  exp(-2 pi i j / nmax)
  // vsini stage maps (theta-independent, smoothing.py:649-668 + :311)
  const int* rs1_idx;      // [n1]   source pixel k of resampled point j
  const float* rs1_frac;   // [n1]   weight of pixel k+1
  const int* bk1_idx;      // [npix] source point j of ANN pixel i (-1: NaN)
  const float* bk1_frac;   // [npix]
  double vs_val;           // 1/(n1*dv1): rfftfreq spacing of the vsini grid
  // observed grid
  const double* lnobs;     // [nobs] ln(obs wave)
  const double* xcheb;     // [nobs] polycalc abscissa in [-1,1]
  const float* obs_f1;     // [nobs] obs flux - 1
  const float* obs_ivar;   // [nobs] 1/eflux^2
  double obs_min, obs_max; // min/max of obs wave (mask_wave limits)
  double r_ann;            // sigma-based R of the ANN
  double geo_inv_dln;      // 1/mean(d lnlam): index guess for the resamplers
  int npoly;               // blaze coefficients (0: off)
  // geometric ANN grid (readc3k construction): ln lam_k = ln0 + k*dln to < 1e-12, so
  // pixel positions come from arithmetic instead of dependent loads of lnlam[]
  int geo;
  double ln0, dln, ln_last;   // ln0 = lnlam[0], ln_last = lnlam[npix-1] (always set)
  // vsini taper sb(u) tabulated at u = i*kVsTabStep (host, fp64); cubic interpolation
  const double* vs_tab;
  int vs_tab_n;
  int rot_identity;    // the vsini resampling maps are the identity (to fp32): skip them
};

constexpr double kVsTabStep = 1.0 / 64.0;
constexpr double kVsTabMax = 256.0;

// Per-candidate scalars, shared by the workgroup (lives in LDS).
struct CandState {
  double one_plus;   // 1 + rv/c
  double dop;        // ln(one_plus)
  double lnmin, lnmax, step, inv_step;
  double vs_a;       // 2 pi sigma        (vsini)
  double g_a;        // -2 pi^2 sigma^2   (gauss)
  double g_val;      // 1/(n2*dv2)
  double poly[12];
  int do_rot, do_smooth;
  int i0, i1, n2;
  int bad;           // window too small for an FFT -> NaN result
};

// ---------------------------------------------------------------------------
// Radix-2/4/8 Stockham passes on M complex points held in LDS.
// Thread i owns butterfly i of M/R: reads src[i + r*M/R], twiddles by
// exp(-2 pi i k r/(pR)) (k = i mod p), writes dst[(i-k)R + k + r p].
// ---------------------------------------------------------------------------
PAYNE_HD void dft2(c32* u) { c32 a = u[0], b = u[1]; u[0] = cadd(a, b); u[1] = csub(a, b); }
PAYNE_HD void dft4(c32& a0, c32& a1, c32& a2, c32& a3) {
  c32 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmul_negi(csub(a1, a3));
  a0 = cadd(t0, t2); a1 = cadd(t1, t3); a2 = csub(t0, t2); a3 = csub(t1, t3);
}
PAYNE_HD void dft8(c32* u) {
  c32 e0 = u[0], e1 = u[2], e2 = u[4], e3 = u[6], o0 = u[1], o1 = u[3], o2 = u[5], o3 = u[7];
  dft4(e0, e1, e2, e3); dft4(o0, o1, o2, o3);
  const float h = 0.70710678118654752f;
  c32 w1o = {h * (o1.x + o1.y), h * (o1.y - o1.x)};       // o1 * (1-i)/sqrt2
  c32 w2o = cmul_negi(o2);                                // o2 * (-i)
  c32 w3o = {h * (o3.y - o3.x), -h * (o3.x + o3.y)};      // o3 * (-1-i)/sqrt2
  u[0] = cadd(e0, o0); u[4] = csub(e0, o0);
  u[1] = cadd(e1, w1o); u[5] = csub(e1, w1o);
  u[2] = cadd(e2, w2o); u[6] = csub(e2, w2o);
  u[3] = cadd(e3, w3o); u[7] = csub(e3, w3o);
}

// Twiddle exp(-2 pi i j / tw_n) from a table holding the first half circle only.
PAYNE_HD c32 tw_get(const c32* tw, int half, int j) {
  const c32 w = tw[j & (half - 1)];
  return (j & half) ? c32{-w.x, -w.y} : w;
}

template <int R>
PAYNE_HD void fft_pass(int tid, int nthr, const c32* __restrict__ src, c32* __restrict__ dst, int M, int p,
                       const c32* __restrict__ tw, int tw_n, bool conj_out) {
  const int nb = M / R;
  const int half = tw_n >> 1;
  for (int i = tid; i < nb; i += nthr) {
    const int k = i & (p - 1);
    c32 u[R];
#pragma unroll
    for (int r = 0; r < R; ++r) u[r] = src[i + r * nb];
    if (p > 1) {
      const int ts = tw_n / (p * R);
      c32 w[R];
#pragma unroll
      for (int r = 1; r < R; ++r) w[r] = tw_get(tw, half, (k * r) * ts);
#pragma unroll
      for (int r = 1; r < R; ++r) u[r] = cmul(u[r], w[r]);
    }
    if (R == 8) dft8(u);
    else if (R == 4) dft4(u[0], u[1], u[2], u[3]);
    else dft2(u);
    const int j = (i - k) * R + k;
#pragma unroll
    for (int r = 0; r < R; ++r) dst[j + r * p] = conj_out ? cconj(u[r]) : u[r];
  }
}

// Radix of the pass that starts at sub-length p for an M-point transform.
PAYNE_HD int pass_radix(int M, int p) { int rem = M / p; return rem >= 8 ? 8 : rem; }

// ---------------------------------------------------------------------------
// Tapers.
// ---------------------------------------------------------------------------
// sb(ub) of smoothing.py:616-617 evaluated directly in fp64
PAYNE_HD double vsini_sb_exact(double ub) {
  double s, c;
#ifdef __HIP_DEVICE_COMPILE__
  sincos(ub, &s, &c);
#else
  s = sin(ub); c = cos(ub);
#endif
  double u2 = ub * ub;
  return j1(ub) / ub - 3.0 * c / (2.0 * u2) + 3.0 * s / (2.0 * (u2 * ub));
}
// 4-point Lagrange interpolation in the host-built table (|error| < 1e-9); sb is even in u.
PAYNE_HD double vsini_sb_table(const double* __restrict__ tab, double ub) {
  const double t = ub * (1.0 / kVsTabStep);
  const int i = (int)t;
  const double f = t - (double)i;
  const double ym = tab[i > 0 ? i - 1 : 1], y0 = tab[i], y1 = tab[i + 1], y2 = tab[i + 2];
  const double fm1 = f - 1.0, fm2 = f - 2.0, fp1 = f + 1.0;
  return (-f * fm1 * fm2 * (1.0 / 6.0)) * ym + (fp1 * fm1 * fm2 * 0.5) * y0 + (-fp1 * f * fm2 * 0.5) * y1 +
         (fp1 * f * fm1 * (1.0 / 6.0)) * y2;
}
PAYNE_HD double vsini_taper(const double* tab, double vs_a, double vs_val, int k) {
  // smoothing.py:612-620 (ss[0] hack irrelevant: sb[0] is overwritten with 1)
  if (k == 0) return 1.0;
  const double ub = vs_a * ((double)k * vs_val);
  if (tab && ub < kVsTabMax) return vsini_sb_table(tab, ub);
  return vsini_sb_exact(ub);          // NaN/huge arguments and the far tail
}
PAYNE_HD float gauss_taper(double g_a, double g_val, int k) {
  // smoothing.py:598-599: exp(-2 pi^2 sigma^2 ss^2), ss = k/(n dv)
  double ss = (double)k * g_val;
  return expf((float)(g_a * (ss * ss)));
}

// Middle step of a real convolution done with a half-length complex FFT.
// In: Z = FFT_M(z), z[n] = s[2n] + i s[2n+1].  Out (in place): Y with
// FFT_M(Y) = conj(z'), z'[n] = s'[2n] + i s'[2n+1], s' = irfft(rfft(s) * taper).
// A thread owns conjugate pairs (k, M-k), k = 1..M/2-1, PU at a time (all loads of the PU
// pairs are issued before any store: the pairs are disjoint, so this is safe in place);
// the two self-conjugate bins k = 0 and k = M/2 go to the last two threads.
template <bool VSINI>
PAYNE_HD void rfft_taper_phase(int tid, int nthr, c32* Z, int M, const c32* __restrict__ tw, int tw_n,
                               double ta, double tval, const double* vs_tab) {
  constexpr int PU = 4;
  const int ts = tw_n / (2 * M);
  const float g = 0.25f / (float)M, invM = 1.0f / (float)M;
  const int npair = M / 2 - 1;                        // k = 1 .. M/2-1
  for (int base = tid; base < npair; base += PU * nthr) {
    c32 zk[PU], zm[PU], w[PU];
    float tk[PU], tm[PU];
#pragma unroll
    for (int q = 0; q < PU; ++q) {
      const int k = 1 + base + q * nthr;
      if (k <= npair) {
        zk[q] = Z[k]; zm[q] = cconj(Z[M - k]); w[q] = tw[k * ts];
        tk[q] = VSINI ? (float)vsini_taper(vs_tab, ta, tval, k) : gauss_taper(ta, tval, k);
        tm[q] = VSINI ? (float)vsini_taper(vs_tab, ta, tval, M - k) : gauss_taper(ta, tval, M - k);
      }
    }
#pragma unroll
    for (int q = 0; q < PU; ++q) {
      const int k = 1 + base + q * nthr;
      if (k <= npair) {
        const c32 A = cadd(zk[q], zm[q]);
        const c32 C = cmul(w[q], cmul_negi(csub(zk[q], zm[q])));
        const c32 S1 = cscale(cadd(A, C), tk[q] * g), S2 = cscale(csub(A, C), tm[q] * g);
        const c32 E = cadd(S1, S2);
        const c32 iO = cmul_posi(cmul(cconj(w[q]), csub(S1, S2)));
        Z[k] = cconj(cadd(E, iO));
        Z[M - k] = csub(E, iO);
      }
    }
  }
  if (tid == nthr - 1) {                               // k = 0 with k = M (real bins X[0], X[M])
    const float t0 = VSINI ? (float)vsini_taper(vs_tab, ta, tval, 0) : gauss_taper(ta, tval, 0);
    const float tM = VSINI ? (float)vsini_taper(vs_tab, ta, tval, M) : gauss_taper(ta, tval, M);
    const c32 z0 = Z[0];
    const float x0 = t0 * (z0.x + z0.y), xm = tM * (z0.x - z0.y);
    Z[0] = {0.5f * (x0 + xm) * invM, -0.5f * (x0 - xm) * invM};
  }
  if (tid == (nthr > 1 ? nthr - 2 : 0) && M >= 2) {    // k = M/2 (self-conjugate)
    const int k = M / 2;
    const float th = VSINI ? (float)vsini_taper(vs_tab, ta, tval, k) : gauss_taper(ta, tval, k);
    Z[k] = cscale(cconj(Z[k]), th * invM);
  }
}

// ---------------------------------------------------------------------------
// Search helpers (monotone fp64 arrays).
// ---------------------------------------------------------------------------
// k in [lo, hi-2] with x[k] <= v < x[k+1] (clamped at the ends), from a guess.
PAYNE_HD int locate(const double* x, int lo, int hi, double v, int guess) {
  int k = guess < lo ? lo : (guess > hi - 2 ? hi - 2 : guess);
  while (k > lo && v < x[k]) --k;
  while (k < hi - 2 && v >= x[k + 1]) ++k;
  return k;
}
// Position of ln-wavelength v on the ANN grid restricted to pixels [lo, hi): pixel k with
// lnlam[k] <= v < lnlam[k+1] (clamped), u = v - lnlam[k], dv = lnlam[k+1] - lnlam[k].
PAYNE_HD void grid_locate(const PostTables& T, int lo, int hi, double v, int& k, double& u, double& dv) {
  if (T.geo) {
    k = (int)((v - T.ln0) * T.geo_inv_dln);
    k = k < lo ? lo : (k > hi - 2 ? hi - 2 : k);
    u = v - (T.ln0 + (double)k * T.dln);
    dv = T.dln;
    if (u < 0.0 && k > lo) { --k; u += dv; }
    else if (u >= dv && k < hi - 2) { ++k; u -= dv; }
  } else {
    const int guess = lo + (int)((v - T.lnlam[lo]) * T.geo_inv_dln);
    k = locate(T.lnlam, lo, hi, v, guess);
    u = v - T.lnlam[k];
    dv = T.lnlam[k + 1] - T.lnlam[k];
  }
}

// interpolation weight from log-space offsets: expm1(u)/expm1(v)
PAYNE_HD float lerp_weight(double u, double v) {
  float uf = (float)u, vf = (float)v;
  return (uf / vf) * (1.0f + 0.5f * (uf - vf));
}
PAYNE_HD float nanf_() { return __builtin_nanf(""); }
PAYNE_HD float nan_to_zero(float v) { return (v != v) ? 0.0f : v; }
PAYNE_HD int pow2ceil(int n) { int p = 1; while (p < n) p <<= 1; return p; }

// ---------------------------------------------------------------------------
// Phases.  `spec`/`work` are the two LDS spectra buffers (n1 floats each).  Loops that
// read one LDS buffer and write the other are written "load U items, then store U
// items" so that the U gathers are in flight together (the kernel runs 2 waves per
// SIMD: latency is hidden by ILP, not by occupancy).
// ---------------------------------------------------------------------------
constexpr int kU = 4;

// P0: per-candidate scalars from theta.  The independent fp64 chains (log / sqrt / the
// instrument width) are given to the first thread of different waves so they overlap.
// theta columns: 0 Teff 1 logg 2 FeH 3 aFe 4 Vrad 5 Vrot 6 Vmic 7 Inst_R 8.. pc_*
PAYNE_HD void phase_setup(int tid, int nthr, const PostTables& T, const double* th, double instr_factor,
                          CandState& S) {
  const int lanes = nthr >= 256 ? 64 : 0;            // 0: everything on thread 0 (small / emulated groups)
  if (tid == 0) {
    const double rv = th[4];
    S.one_plus = (rv != 0.0) ? (1.0 + (rv / kCDoppler)) : 1.0;   // ystpred.py:228-232
    S.dop = log(S.one_plus);
  }
  if (tid == lanes) {
    const double vrot = th[5];
    S.do_rot = (vrot != 0.0);                       // ystpred.py:214 (NaN passes)
    S.vs_a = 2.0 * kPi * sqrt(vrot * vrot - 0.0);   // smoothing.py:297,614
    S.i0 = T.npix; S.i1 = -1; S.n2 = 0; S.bad = 0;
  }
  if (tid == 2 * lanes) {
    const double Rs = th[7] * instr_factor;         // genmod.py:82-85
    S.do_smooth = (Rs > 0.0);                       // ystpred.py:238-240 (false for NaN)
    S.g_a = 0.0;
    if (Rs > 0.0) {
      const double sig_out = kCkms / Rs, inres = kCkms / T.r_ann;   // smoothing.py:107,113
      const double sig = sqrt(sig_out * sig_out - inres * inres);   // :271 (NaN if negative)
      S.g_a = -2.0 * (kPi * kPi) * (sig * sig);
    }
  }
  if (tid == 3 * lanes)
    for (int i = 0; i < T.npoly && i < 12; ++i) S.poly[i] = th[8 + i];
}

// P1: load the raw ANN spectrum (already shifted by -1) into LDS.
PAYNE_HD void phase_load(int tid, int nthr, const PostTables& T, const float* __restrict__ raw,
                         float* __restrict__ spec) {
  if (((T.npix & 3) == 0) && ((((uintptr_t)raw) & 15) == 0)) {
    const int n4 = T.npix >> 2;
    for (int base = tid; base < n4; base += kU * nthr) {
      float v[kU][4];
#pragma unroll
      for (int q = 0; q < kU; ++q) {
        const int i = base + q * nthr;
        if (i < n4) { v[q][0] = raw[4 * i]; v[q][1] = raw[4 * i + 1]; v[q][2] = raw[4 * i + 2]; v[q][3] = raw[4 * i + 3]; }
      }
#pragma unroll
      for (int q = 0; q < kU; ++q) {
        const int i = base + q * nthr;
        if (i < n4) { spec[4 * i] = v[q][0]; spec[4 * i + 1] = v[q][1]; spec[4 * i + 2] = v[q][2]; spec[4 * i + 3] = v[q][3]; }
      }
    }
  } else {
    for (int i = tid; i < T.npix; i += nthr) spec[i] = raw[i];
  }
}

// vsini a: resample onto the pow-2 log grid (static map) into `work`; an identity map
// (geometric grid with npix a power of two) degenerates to a NaN-scrubbing copy.
PAYNE_HD void phase_rot_resample(int tid, int nthr, const PostTables& T, const float* __restrict__ spec,
                                 float* __restrict__ work) {
  for (int base = tid; base < T.n1; base += kU * nthr) {
    float a[kU], b[kU], f[kU];
#pragma unroll
    for (int q = 0; q < kU; ++q) {
      const int j = base + q * nthr;
      if (j < T.n1) {
        if (T.rot_identity) { a[q] = spec[j]; b[q] = a[q]; f[q] = 0.f; }
        else { const int k = T.rs1_idx[j]; f[q] = T.rs1_frac[j]; a[q] = spec[k]; b[q] = spec[k + 1]; }
      }
    }
#pragma unroll
    for (int q = 0; q < kU; ++q) {
      const int j = base + q * nthr;
      if (j < T.n1) {
        const float aa = nan_to_zero(a[q]), bb = nan_to_zero(b[q]);   // nan_to_num(nan=1.0), smoothing.py:138
        work[j] = aa + (bb - aa) * f[q];
      }
    }
  }
}
// vsini c: back onto the ANN grid (left/right = NaN), into `spec` (skipped for identity maps:
// the convolved buffer then IS the spectrum on the ANN grid).
PAYNE_HD void phase_rot_back(int tid, int nthr, const PostTables& T, const float* __restrict__ work,
                             float* __restrict__ spec) {
  for (int base = tid; base < T.npix; base += kU * nthr) {
    float a[kU], b[kU], f[kU];
    int jj[kU];
#pragma unroll
    for (int q = 0; q < kU; ++q) {
      const int i = base + q * nthr;
      if (i < T.npix) {
        jj[q] = T.bk1_idx[i]; f[q] = T.bk1_frac[i];
        const int j = jj[q] < 0 ? 0 : jj[q];
        a[q] = work[j]; b[q] = work[j + 1];
      }
    }
#pragma unroll
    for (int q = 0; q < kU; ++q) {
      const int i = base + q * nthr;
      if (i < T.npix) spec[i] = (jj[q] < 0) ? nanf_() : a[q] + (b[q] - a[q]) * f[q];
    }
  }
}
// vsini d: spec[0]=spec[1]; spec[-1]=spec[-2]  (ystpred.py:223-224)
PAYNE_HD void phase_rot_edges(int tid, const PostTables& T, float* spec) {
  if (tid == 0) spec[0] = spec[1];
  if (tid == 1) spec[T.npix - 1] = spec[T.npix - 2];
}

// R a: data-dependent mask (smoothing.py:631-647), exact fp64 products as numpy.  The
// ANN grid is increasing, so the mask is one run of pixels: the thread that sees the run
// start (end) writes S.i0 (S.i1, inclusive) -- exactly one writer each, no atomics.
PAYNE_HD void phase_mask_scan(int tid, int nthr, const PostTables& T, const double* th,
                              double instr_factor, CandState& S) {
  const double Rs = th[7] * instr_factor;
  const double pad = 20.0 / Rs;
  const double wl = T.obs_min * (1.0 + pad * -1.0), wh = T.obs_max * (1.0 + pad * 1.0);
  const double op = S.one_plus;
  for (int base = tid; base < T.npix; base += kU * nthr) {
    double wc[kU], wp[kU];
#pragma unroll
    for (int q = 0; q < kU; ++q) {
      const int i = base + q * nthr;
      if (i < T.npix) { wc[q] = T.lam[i]; wp[q] = T.lam[i > 0 ? i - 1 : 0]; }
    }
#pragma unroll
    for (int q = 0; q < kU; ++q) {
      const int i = base + q * nthr;
      if (i < T.npix) {
        const double c = wc[q] * op, p = wp[q] * op;
        const bool in_c = (c > wl) && (c < wh);
        const bool in_p = (i > 0) && (p > wl) && (p < wh);
        if (in_c && !in_p) S.i0 = i;
        if (!in_c && in_p) S.i1 = i - 1;
        if (in_c && i == T.npix - 1) S.i1 = i;
      }
    }
  }
}
// R b: window scalars (one thread): resample_wave's grid (smoothing.py:654-661).
// ln(lam_k (1+rv/c)) is taken as lnlam[k] + dop (differs from log of the rounded product
// by < 2e-16, i.e. < 1e-10 pixel: the grid and the interpolation depend on it continuously).
PAYNE_HD void phase_window(int tid, const PostTables& T, CandState& S) {
  if (tid != 0) return;
  int n = S.i1 - S.i0 + 1;
  if (S.i1 < 0) n = 0;
  S.i1 = S.i0 + n;                     // exclusive from here on
  if (n < 8) { S.bad = 1; S.n2 = 8; return; }
  S.n2 = pow2ceil(n);
  S.lnmin = T.lnlam[S.i0] + S.dop;
  S.lnmax = T.lnlam[S.i1 - 1] + S.dop;
  S.step = (S.lnmax - S.lnmin) / (double)(S.n2 - 1);     // np.linspace
  S.inv_step = 1.0 / S.step;
  S.g_val = 1.0 / ((double)S.n2 * (kCkms * S.step));     // rfftfreq(n, d=dv), dv = ckms*median(diff(ln w))
}
// R c: resample the masked, Doppler-shifted spectrum onto its pow-2 log grid.
PAYNE_HD void phase_R_resample(int tid, int nthr, const PostTables& T, const CandState& S,
                               const float* __restrict__ spec, float* __restrict__ work) {
  for (int base = tid; base < S.n2; base += kU * nthr) {
    float a[kU], b[kU], w[kU];
#pragma unroll
    for (int q = 0; q < kU; ++q) {
      const int j = base + q * nthr;
      if (j < S.n2) {
        const double lw = (j == S.n2 - 1) ? S.lnmax : ((double)j * S.step + S.lnmin);
        const double v = lw - S.dop;                       // position on the unshifted ANN grid
        int k; double u, dv;
        grid_locate(T, S.i0, S.i1, v, k, u, dv);
        a[q] = spec[k]; b[q] = spec[k + 1];
        w[q] = (u <= 0.0) ? 0.f : ((u >= dv) ? 1.f : lerp_weight(u, dv));   // np.interp clamps (default left/right)
      }
    }
#pragma unroll
    for (int q = 0; q < kU; ++q) {
      const int j = base + q * nthr;
      if (j < S.n2) {
        const float aa = nan_to_zero(a[q]), bb = nan_to_zero(b[q]);
        work[j] = (w[q] == 0.f) ? aa : ((w[q] == 1.f) ? bb : aa + (bb - aa) * w[q]);
      }
    }
  }
}

// Final: interpolate onto the observed grid, blaze, chi^2 partial per thread.
// `conv` = smoothed spectrum on the candidate's log grid (do_smooth) or the
// (rotated) spectrum on the ANN grid (plain np.interp branch, ystpred.py:271-272).
PAYNE_HD double phase_obs(int tid, int nthr, const PostTables& T, const CandState& S,
                          const float* __restrict__ conv, float* __restrict__ out, int out_stage) {
  double acc = 0.0;
  const bool cheb = T.npoly > 0;
  for (int base = tid; base < T.nobs; base += kU * nthr) {
    float a[kU], b[kU], w[kU], of1[kU], iv[kU];
    double xc[kU];
    bool nanv[kU];
#pragma unroll
    for (int q = 0; q < kU; ++q) {
      const int i = base + q * nthr;
      if (i < T.nobs) {
        const double lo = T.lnobs[i];
        if (T.obs_f1) { of1[q] = T.obs_f1[i]; iv[q] = T.obs_ivar[i]; }
        if (cheb) xc[q] = T.xcheb[i];
        nanv[q] = false; a[q] = 0.f; b[q] = 0.f; w[q] = 0.f;
        if (S.bad) nanv[q] = true;
        else if (S.do_smooth) {
          if (lo < S.lnmin || lo > S.lnmax) nanv[q] = true;   // np.interp(left=nan, right=nan)
          else {
            int j = (int)((lo - S.lnmin) * S.inv_step);
            if (j > S.n2 - 2) j = S.n2 - 2;
            double u = lo - ((double)j * S.step + S.lnmin);
            if (u < 0.0 && j > 0) { --j; u += S.step; }
            else if (u >= S.step && j < S.n2 - 2) { ++j; u -= S.step; }
            a[q] = conv[j]; b[q] = conv[j + 1];
            const float ww = lerp_weight(u, S.step);
            w[q] = ww < 0.f ? 0.f : (ww > 1.f ? 1.f : ww);
          }
        } else {
          const double v = lo - S.dop;
          if (v < T.ln0 || v > T.ln_last) nanv[q] = true;
          else {
            int k; double u, dv;
            grid_locate(T, 0, T.npix, v, k, u, dv);
            a[q] = conv[k]; b[q] = conv[k + 1];
            const float ww = lerp_weight(u, dv);
            w[q] = ww < 0.f ? 0.f : (ww > 1.f ? 1.f : ww);
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < kU; ++q) {
      const int i = base + q * nthr;
      if (i < T.nobs) {
        const float m1 = nanv[q] ? nanf_() : a[q] + (b[q] - a[q]) * w[q];
        float pm1 = 0.f, p = 1.f;
        if (cheb) {   // numpy.polynomial.chebyshev.chebval (Clenshaw), fitutils.py:11-20
          const double x = xc[q];
          double c0, c1;
          const int nc = T.npoly;
          if (nc == 1) { c0 = S.poly[0]; c1 = 0.0; }
          else if (nc == 2) { c0 = S.poly[0]; c1 = S.poly[1]; }
          else {
            const double x2 = 2.0 * x;
            c0 = S.poly[nc - 2]; c1 = S.poly[nc - 1];
            for (int r = 3; r <= nc; ++r) { double t = c0; c0 = S.poly[nc - r] - c1; c1 = t + c1 * x2; }
          }
          const double pv = c0 + c1 * x;
          p = (float)pv; pm1 = (float)(pv - 1.0);
        }
        if (out) out[i] = (out_stage == 3) ? (m1 + kBase) * p : (m1 + kBase);   // genspec / getspec
        if (T.obs_f1) {
          const float d = cheb ? (m1 * p + (pm1 - of1[q])) : (m1 - of1[q]);
          acc += (double)(d * d * iv[q]);
        }
      }
    }
  }
  return acc;
}

}  // namespace payne
