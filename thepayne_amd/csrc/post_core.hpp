// post_core.hpp -- per-candidate spectrum pipeline of the likelihood hot path,
// written as barrier-separated PHASES so that the same source runs
//   * on gfx950 inside payne_post_kernel (one 256-thread workgroup per candidate, the
//     spectrum resident in LDS from the ANN output to chi^2), and
//   * on the host (tests/emul/cpu_emul.cpp: phases executed for tid = 0..NT-1 in turn)
//     as the CPU-side check of the kernel logic (also under ASan/UBSan).
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
// ln(lambda): a pixel position is t = (ln x - ln x_0)/dln evaluated with ONE fp64 fma,
// k = floor(t), f = t - k, and the np.interp weight (x-x_k)/(x_{k+1}-x_k) =
// expm1(f v)/expm1(v) ~ f (1 + v (f-1)/2); no per-pixel exp(), no fp32 wavelength
// (SURVEY.md 7.3-1).  The vsini taper (catastrophic cancellation in fp32, 7.3-2) comes
// from a table built on the host in fp64.
//
// The kernel is VALU-issue bound (2 waves per SIMD, ~25 barrier phases): every phase is
// written to minimise instructions per pixel -- compile-time FFT geometry, no integer
// multiplies or fp64 divisions in loops, wave ballots instead of atomics.
#pragma once
#include <math.h>
#include <stdint.h>

#ifdef __HIPCC__
#define PAYNE_HD __host__ __device__ __forceinline__
#define PAYNE_HD_COLD __host__ __device__ inline __attribute__((noinline))
#else
#define PAYNE_HD inline
#define PAYNE_HD_COLD inline
#endif

namespace payne {

constexpr double kCkms = 2.998e5;            // smoothing.py:16
constexpr double kCDoppler = 299792.458;     // ystpred.py:11-12
constexpr double kPi = 3.141592653589793;
constexpr float kBase = 1.0f;                // the flux shift
constexpr double kVsTabStep = 1.0 / 64.0;    // vsini taper table spacing in u
constexpr double kVsTabMax = 256.0;
constexpr double kPosMagic = 1572864.0;      // 1.5 * 2^20: a position t in [0, 2^19) added to it has ulp 2^-32 (magic_locate, vsini_tab_pos)
constexpr unsigned kPosMagicHi = 0x41380000u; // its high dword: the high dword of t + kPosMagic is kPosMagicHi + floor(t) for 0 <= t < 2^19
constexpr int kObsPad = 8192;                // PostTables::obs_rec holds a multiple of this many records (host_tables.hpp build_obs_tables)
// Threads of the per-candidate workgroup, and how many of them own a butterfly in the radix-8 passes.
// One wave issues a vector instruction every ~5 cycles whatever its kind (tools/exp/pk_rate.hip) while a SIMD
// keeps four waves going at that rate each, so the per-pixel phases (tapers, resampling, observed grid) are
// bound by INSTRUCTIONS PER WAVE: eight waves halve them.  The transform keeps radix-8 passes on the first four
// waves (the other four only take part in the barriers): radix-4 passes on all eight (6 barriers per transform
// instead of 4) cost more than the occupancy gains -- measured at C2 (512 candidates, 2 groups per CU):
// 256 / 256 threads 21.8 us per batch, 512 / 512 (radix 4) 32.2 us, 512 / 256 19.5 us.
#ifndef PAYNE_POST_THREADS
#define PAYNE_POST_THREADS 512
#endif
constexpr int kPostThreads = PAYNE_POST_THREADS;
constexpr int kFftThreads = 256;
// unroll factor of the per-pixel loops for `ppt` pixels per thread (loads of one unrolled body are in flight together)
PAYNE_HD constexpr int unroll_for(int ppt) { return ppt >= 16 ? 16 : (ppt >= 8 ? 8 : 4); }

// (8-byte aligned: a complex value is then ONE 8-byte LDS access -- ds_read_b64 / ds_write_b64, conflict-free for consecutive lanes --;
//  as two floats of 4-byte alignment the taper phases read and wrote it as ds_read2_b32 / ds_write2_b32, whose lanes i and i + 16
//  share a bank: a third of the kernel's bank-conflict cycles.  Every complex buffer of this build sits on an 8-byte boundary.)
struct alignas(8) c32 { float x, y; };
#ifdef __HIP_DEVICE_COMPILE__
// Complex arithmetic on the packed-fp32 unit, one instruction per line (the compiler's own selection from the scalar
// statements spends four instructions on a complex product -- both halves computed twice, merged by a move -- and a
// quarter of a transform's vector instructions on moves that re-pair halves): a complex value is a 64-bit register pair
// from the LDS read to the LDS write; which half of a source feeds which half of the result (op_sel / op_sel_hi) and its
// sign (neg_lo / neg_hi) are part of the instruction, so multiplications by +-i and conjugations cost nothing.
typedef float pk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk2 to_pk(c32 a) { pk2 t; t.x = a.x; t.y = a.y; return t; }
__device__ __forceinline__ c32 un_pk(pk2 t) { return {t.x, t.y}; }
__device__ __forceinline__ pk2 pk_add(pk2 a, pk2 b) { pk2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ pk2 pk_sub(pk2 a, pk2 b) { pk2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
// a + (-i) b = (a.x + b.y, a.y - b.x)       a - (-i) b = a + i b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ pk2 pk_add_mi(pk2 a, pk2 b) { pk2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ pk2 pk_add_pi(pk2 a, pk2 b) { pk2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
// a + conj(b), a - conj(b), conj(a + b)
__device__ __forceinline__ pk2 pk_add_cj(pk2 a, pk2 b) { pk2 r; asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ pk2 pk_sub_cj(pk2 a, pk2 b) { pk2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ pk2 pk_cj_add(pk2 a, pk2 b) { pk2 r; asm("v_pk_add_f32 %0, %1, %2 neg_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
// c + a h, c - a h with the constant pair hh = (h, h) in scalar registers
__device__ __forceinline__ pk2 pk_fma_c(pk2 a, pk2 hh, pk2 c) { pk2 r; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(hh), "v"(c)); return r; }
__device__ __forceinline__ pk2 pk_fnma_c(pk2 a, pk2 hh, pk2 c) { pk2 r; asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "s"(hh), "v"(c)); return r; }
// (several instructions per asm statement where one feeds the next: the compiler cannot see into a statement and separates two
//  DEPENDENT statements by an s_nop -- it has to assume the first might write half a register --, which costs the issue slot
//  the packed form was meant to save; inside a statement the hardware's own interlock applies)
// a w  and  a conj(w)
__device__ __forceinline__ pk2 pk_cmul(pk2 a, pk2 w) {
  pk2 t, r;
  asm("v_pk_mul_f32 %1, %2, %3 op_sel:[1,1] op_sel_hi:[1,0]\n\t"                                   // (a.y w.y, a.y w.x)
      "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]"                // (a.x w.x - t.x, a.x w.y + t.y)
      : "=v"(r), "=&v"(t) : "v"(a), "v"(w));
  return r;
}
__device__ __forceinline__ pk2 pk_cmul_cj(pk2 a, pk2 w) {
  pk2 t, r;
  asm("v_pk_mul_f32 %1, %2, %3 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"                // (a.x w.x + t.x, -a.x w.y + t.y)
      : "=v"(r), "=&v"(t) : "v"(a), "v"(w));
  return r;
}
// two products at once (the second product's first instruction sits between the two dependent ones of the first)
__device__ __forceinline__ void pk_cmul2(pk2& a, pk2 wa, pk2& b, pk2 wb) {
  pk2 t, s;
  asm("v_pk_mul_f32 %2, %0, %4 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %3, %1, %5 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %0, %0, %4, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]\n\t"
      "v_pk_fma_f32 %1, %1, %5, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]"
      : "+v"(a), "+v"(b), "=&v"(t), "=&v"(s) : "v"(wa), "v"(wb));
}
// nan_to_num(nan = 1.0) of eight shifted fluxes (NaN -> 0, smoothing.py:138).  The compiler's form of one -- v_cmp_o into vcc, the
// two wait states a vector read of a freshly written condition register needs (s_nop 1), v_cndmask -- costs a wave 20 ticks, four
// instructions' worth (tools/exp/nop_rate.hip).  Eight comparisons into eight scalar register pairs, then eight selects: two
// instructions a value, the wait states covered by the comparisons in between.
__device__ __forceinline__ void nan_scrub8(float& a0, float& a1, float& a2, float& a3, float& a4, float& a5, float& a6, float& a7) {
  unsigned long long m0, m1, m2, m3, m4, m5, m6, m7;
  asm("v_cmp_o_f32 %8, %0, %0\n\t" "v_cmp_o_f32 %9, %1, %1\n\t" "v_cmp_o_f32 %10, %2, %2\n\t" "v_cmp_o_f32 %11, %3, %3\n\t"
      "v_cmp_o_f32 %12, %4, %4\n\t" "v_cmp_o_f32 %13, %5, %5\n\t" "v_cmp_o_f32 %14, %6, %6\n\t" "v_cmp_o_f32 %15, %7, %7\n\t"
      "v_cndmask_b32 %0, 0, %0, %8\n\t" "v_cndmask_b32 %1, 0, %1, %9\n\t" "v_cndmask_b32 %2, 0, %2, %10\n\t" "v_cndmask_b32 %3, 0, %3, %11\n\t"
      "v_cndmask_b32 %4, 0, %4, %12\n\t" "v_cndmask_b32 %5, 0, %5, %13\n\t" "v_cndmask_b32 %6, 0, %6, %14\n\t" "v_cndmask_b32 %7, 0, %7, %15"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
        "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(m4), "=&s"(m5), "=&s"(m6), "=&s"(m7));
}
// (a, b) *= (wa la, wb lb): the two table products and the two data products in ONE statement (wa, wb come back as the products)
__device__ __forceinline__ void pk_cmul2x2(pk2& a, pk2& b, pk2& wa, pk2 la, pk2& wb, pk2 lb) {
  pk2 t, s;
  asm("v_pk_mul_f32 %4, %2, %6 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %5, %3, %7 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %2, %2, %6, %4 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]\n\t"
      "v_pk_fma_f32 %3, %3, %7, %5 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]\n\t"
      "v_pk_mul_f32 %4, %0, %2 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %5, %1, %3 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %0, %0, %2, %4 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]\n\t"
      "v_pk_fma_f32 %1, %1, %3, %5 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]"
      : "+v"(a), "+v"(b), "+v"(wa), "+v"(wb), "=&v"(t), "=&v"(s) : "v"(la), "v"(lb));
}
// a (c - i s) with the constant pair cs = (c, s) in scalar registers: (a.x c + a.y s, a.y c - a.x s)
__device__ __forceinline__ pk2 pk_cmul_k(pk2 a, pk2 cs) {
  pk2 t, r;
  asm("v_pk_mul_f32 %1, %2, %3 op_sel_hi:[1,0]\n\t"                                               // (a.x c, a.y c)
      "v_pk_fma_f32 %0, %2, %3, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"                // (a.y s + t.x, -a.x s + t.y)
      : "=v"(r), "=&v"(t) : "v"(a), "s"(cs));
  return r;
}
__device__ __forceinline__ void pk_cmul_k2(pk2& a, pk2 ca, pk2& b, pk2 cb) {
  pk2 t, s;
  asm("v_pk_mul_f32 %2, %0, %4 op_sel_hi:[1,0]\n\t"
      "v_pk_mul_f32 %3, %1, %5 op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %0, %0, %4, %2 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]\n\t"
      "v_pk_fma_f32 %1, %1, %5, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
      : "+v"(a), "+v"(b), "=&v"(t), "=&v"(s) : "s"(ca), "s"(cb));
}
__device__ __forceinline__ c32 cmul(c32 a, c32 b) { return un_pk(pk_cmul(to_pk(a), to_pk(b))); }
#else
PAYNE_HD c32 cmul(c32 a, c32 b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
#endif
PAYNE_HD c32 cadd(c32 a, c32 b) { return {a.x + b.x, a.y + b.y}; }
PAYNE_HD c32 csub(c32 a, c32 b) { return {a.x - b.x, a.y - b.y}; }
PAYNE_HD c32 cconj(c32 a) { return {a.x, -a.y}; }
PAYNE_HD c32 cscale(c32 a, float s) { return {a.x * s, a.y * s}; }
PAYNE_HD c32 cmul_negi(c32 a) { return {a.y, -a.x}; }   // a * (-i)
PAYNE_HD c32 cmul_posi(c32 a) { return {-a.y, a.x}; }   // a * (+i)
PAYNE_HD float nanf_() { return __builtin_nanf(""); }
constexpr double kEdgeTol = 1e-12;   // ln(lambda): a pixel this close to an end of the model grid counts as on it
PAYNE_HD float nan_to_zero(float v) { return (v != v) ? 0.0f : v; }
PAYNE_HD int pow2ceil(int n) { int p = 1; while (p < n) p <<= 1; return p; }

// ---------------------------------------------------------------------------
// Static (per-context) tables, all device-resident; built once on the host
// (host_tables.hpp).
// ---------------------------------------------------------------------------
// observed pixel as the chi^2 loop reads it (f1 = ivar = 0 when no flux is bound)
struct alignas(16) ObsRec { double lnw; float f1, ivar; };

struct PostTables {
  int npix;            // ANN pixels
  int nobs;            // observed pixels (0: no obs grid bound)
  int n1;              // pow2ceil(npix): vsini FFT length (real points)
  int nmax;            // twiddle table length (== n1)
  const double* lnlam;     // [npix]  ln(lambda_ANN)
  const double* lam;       // [npix]  lambda_ANN (exact mask test)
  const c32* tw;           // [nmax]  exp(-2 pi i j / nmax), full circle (runtime-geometry FFT)
  const c32* twf;          // [twf_n] pass-ordered twiddles of the n1/2-point FFT + exp(-2 pi i k/n1), k < n1/4
  int twf_n;
  // vsini stage maps (theta-independent, smoothing.py:649-668 + :311)
  const int* rs1_idx;      // [n1]   source pixel k of resampled point j
  const float* rs1_frac;   // [n1]   weight of pixel k+1
  const int* bk1_idx;      // [npix] source point j of ANN pixel i (-1: NaN)
  const float* bk1_frac;   // [npix]
  double vs_val;           // 1/(n1*dv1): rfftfreq spacing of the vsini grid
  // observed grid
  const double* lnobs;     // [nobs] ln(obs wave)
  const struct ObsRec* obs_rec;   // [nobs] {ln(obs wave), flux - 1, 1/eflux^2} in one 16-byte record (one load per pixel)
  const double* xcheb;     // [nobs] polycalc abscissa in [-1,1]
  const float* obs_f1;     // [nobs] obs flux - 1
  const float* obs_ivar;   // [nobs] 1/eflux^2
  double obs_min, obs_max; // min/max of obs wave (mask_wave limits)
  double ln_obs_min, ln_obs_max;   // their logarithms (host, fp64): prep_candidate's mask counts by arithmetic on geometric grids
  double r_ann;            // sigma-based R of the ANN
  double geo_inv_dln;      // 1/mean(d lnlam)
  int npoly;               // blaze coefficients (0: off)
  // geometric ANN grid (readc3k construction): ln lam_k = ln0 + k*dln to < 1e-12, so
  // pixel positions come from arithmetic instead of dependent loads of lnlam[]
  int geo;
  double ln0, dln, ln_last;   // ln0 = lnlam[0], ln_last = lnlam[npix-1] (always set)
  const float* vs_tab;        // vsini taper sb(i*kVsTabStep), fp32 copy of the fp64 host table
  int vs_tab_n;
  int rot_identity;    // the vsini resampling maps are the identity (to fp32): skip them
  float inv_lam0, inv_dln32;  // 1/lam[0], 1/dln in fp32: the +-31-pixel position guess of the mask probe
  // rot_back on a geometric grid (phase_rot_back): model pixel i sits at i * bk_r of the rotation stage's grid; weight constants of
  // f (1 + hs (f - 1)) on the 32-bit fraction F as F (bk_c1 + bk_c2 F) -- made once on the host (two fp64 divisions a thread otherwise)
  double bk_r;
  float bk_c1, bk_c2;
  int obs_sorted;      // the observed wavelengths ascend (the on-chip instrumental stage walks them half a spectrum at a time)
  int raw_freq;        // this launch's rows are the half transform of the spectra in pair layout (host_tables.hpp freq_rows): the output
                       // layer carried the first stage's forward transform in its weights (identity vsini maps, compile-time geometry)
};

// Per-candidate scalars from theta, shared by the workgroup (lives in LDS).
// The R-stage window of one candidate, derived (redundantly, by every thread) from the
// mask bounds: resample_wave's grid (smoothing.py:654-661) and the two position maps.
struct Window {
  double lnmin, lnmax;   // ln of the first / last masked, Doppler-shifted ANN pixel
  double rsA, rsB;       // ANN-pixel position of resampled point j: t = j*rsA + rsB   (geo grids)
  double step;           // d ln(lambda) of the resampled grid
  double obA, obB;       // resampled-grid position of obs pixel i: t = lnobs_i*obA + obB
  float hs_ann, hs_step; // half grid steps: weight = f (1 + hs (f-1))
  float g_c2;            // Gaussian taper = exp2(g_c2 k^2)
  int i0, i1;            // masked pixels [i0, i1)
  int n2;                // FFT length of the R stage
  int bad;               // window too small for an FFT -> NaN result
};

struct CandState {
  double one_plus;   // 1 + rv/c
  double dop;        // ln(one_plus)
  double vs_a;       // 2 pi sigma        (vsini)
  double g_a;        // -2 pi^2 sigma^2   (gauss)
  double wl, wh;     // mask limits (smoothing.py:631-647)
  double poly[12];
  int do_rot, do_smooth;
  int win_ready;     // the two mask counts below come from phase_setup's probe (geometric grids)
  int win_below, win_notabove;
  int w_ready;       // W below is valid (filled by prep_candidate, ahead of the kernel)
  Window W;
};

#ifdef __HIP_DEVICE_COMPILE__
constexpr int kSlotShift = 6;        // one partial per wave (ballot / shuffle reductions)
#else
constexpr int kSlotShift = 0;        // host emulation: one partial per thread
#endif
PAYNE_HD int n_slots(int nthr) { return nthr >> kSlotShift; }
// did some thread of the wave see it?  (host emulation: a "wave" is one thread)
PAYNE_HD bool wave_any(bool p) {
#ifdef __HIP_DEVICE_COMPILE__
  return __ballot(p) != 0ull;
#else
  return p;
#endif
}

// ---------------------------------------------------------------------------
// FFT: Stockham radix-2/4/8 passes on M complex points held in LDS.
// Thread i owns butterfly i of M/R: reads src[i + r*M/R], twiddles by
// exp(-2 pi i k r/(pR)) (k = i mod p), writes dst[(i-k)R + k + r p].
// ---------------------------------------------------------------------------
PAYNE_HD void dft2(c32* u) { c32 a = u[0], b = u[1]; u[0] = cadd(a, b); u[1] = csub(a, b); }
#ifdef __HIP_DEVICE_COMPILE__
// 8 and 26 packed instructions, no instruction next to the one it depends on
__device__ __forceinline__ void pk_dft4(pk2& a0, pk2& a1, pk2& a2, pk2& a3) {
  pk2 t0, t2;
  asm("v_pk_add_f32 %4, %0, %2\n\t"                                                   // t0 = a0 + a2
      "v_pk_add_f32 %5, %1, %3\n\t"                                                   // t2 = a1 + a3
      "v_pk_add_f32 %2, %0, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"                         // t1 = a0 - a2      (in a2's place)
      "v_pk_add_f32 %3, %1, %3 neg_lo:[0,1] neg_hi:[0,1]\n\t"                         // d  = a1 - a3      (in a3's place)
      "v_pk_add_f32 %0, %4, %5\n\t"                                                   // a0 = t0 + t2
      "v_pk_add_f32 %1, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"         // a1 = t1 - i d
      "v_pk_add_f32 %3, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"         // a3 = t1 + i d
      "v_pk_add_f32 %2, %4, %5 neg_lo:[0,1] neg_hi:[0,1]"                               // a2 = t0 - t2
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t2));
}
// ONE statement (26 instructions): between two dependent statements the compiler puts an `s_nop` -- it cannot see that the first one
// writes whole registers --, and an s_nop costs a wave 3.7 of the ~5 ticks a vector instruction costs it (tools/exp/nop_rate.hip);
// the two half-transforms are interleaved instruction by instruction so that nothing sits next to what it depends on (a dependent
// neighbour waits 3.3 ticks more).
__device__ __forceinline__ void pk_dft8(pk2* u) {
  pk2 e0 = u[0], e1 = u[2], e2 = u[4], e3 = u[6], o0 = u[1], o1 = u[3], o2 = u[5], o3 = u[7];
  const pk2 hh = {0.70710678118654752f, 0.70710678118654752f};
  pk2 tA, tB, tC, tD;
  asm("v_pk_add_f32 %8, %0, %2\n\t"                                                   // TE0 = e0 + e2
      "v_pk_add_f32 %10, %4, %6\n\t"                                                  // TO0 = o0 + o2
      "v_pk_add_f32 %9, %1, %3\n\t"                                                   // TE2 = e1 + e3
      "v_pk_add_f32 %11, %5, %7\n\t"                                                  // TO2 = o1 + o3
      "v_pk_add_f32 %2, %0, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"                         // e2 := e0 - e2
      "v_pk_add_f32 %6, %4, %6 neg_lo:[0,1] neg_hi:[0,1]\n\t"                         // o2 := o0 - o2
      "v_pk_add_f32 %3, %1, %3 neg_lo:[0,1] neg_hi:[0,1]\n\t"                         // e3 := e1 - e3
      "v_pk_add_f32 %7, %5, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t"                         // o3 := o1 - o3
      "v_pk_add_f32 %0, %8, %9\n\t"                                                   // E0 = TE0 + TE2
      "v_pk_add_f32 %4, %10, %11\n\t"                                                 // O0 = TO0 + TO2
      "v_pk_add_f32 %1, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"         // E1 = e2 - i e3
      "v_pk_add_f32 %5, %6, %7 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"         // O1 = o2 - i o3
      "v_pk_add_f32 %3, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"         // E3 = e2 + i e3
      "v_pk_add_f32 %7, %6, %7 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"         // O3 = o2 + i o3
      "v_pk_add_f32 %2, %8, %9 neg_lo:[0,1] neg_hi:[0,1]\n\t"                         // E2 = TE0 - TE2
      "v_pk_add_f32 %6, %10, %11 neg_lo:[0,1] neg_hi:[0,1]\n\t"                       // O2 = TO0 - TO2
      "v_pk_add_f32 %8, %5, %5 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"         // s1 = O1 (1 - i)
      "v_pk_add_f32 %9, %7, %7 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"         // s3 = O3 (1 + i)
      "v_pk_add_f32 %10, %0, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t"                        // u4 = E0 - O0
      "v_pk_add_f32 %0, %0, %4\n\t"                                                   // u0 = E0 + O0
      "v_pk_fma_f32 %5, %8, %12, %1 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"                // u5 = E1 - h s1    (in O1's place)
      "v_pk_fma_f32 %1, %8, %12, %1\n\t"                                              // u1 = E1 + h s1
      "v_pk_add_f32 %11, %2, %6 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"        // u6 = E2 + i O2
      "v_pk_add_f32 %2, %2, %6 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"         // u2 = E2 - i O2
      "v_pk_fma_f32 %7, %9, %12, %3\n\t"                                              // u7 = E3 + h s3    (in O3's place)
      "v_pk_fma_f32 %3, %9, %12, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]"                       // u3 = E3 - h s3
      : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3), "=&v"(tA), "=&v"(tB), "=&v"(tC), "=&v"(tD)
      : "s"(hh));
  u[0] = e0; u[1] = e1; u[2] = e2; u[3] = e3; u[4] = tC; u[5] = o1; u[6] = tD; u[7] = o3;
}
__device__ __forceinline__ void dft4(c32& a0, c32& a1, c32& a2, c32& a3) {
  pk2 p0 = to_pk(a0), p1 = to_pk(a1), p2 = to_pk(a2), p3 = to_pk(a3);
  pk_dft4(p0, p1, p2, p3);
  a0 = un_pk(p0); a1 = un_pk(p1); a2 = un_pk(p2); a3 = un_pk(p3);
}
__device__ __forceinline__ void dft8(c32* u) {
  pk2 p[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) p[r] = to_pk(u[r]);
  pk_dft8(p);
#pragma unroll
  for (int r = 0; r < 8; ++r) u[r] = un_pk(p[r]);
}
#else
PAYNE_HD void dft4(c32& a0, c32& a1, c32& a2, c32& a3) {
  c32 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmul_negi(csub(a1, a3));
  a0 = cadd(t0, t2); a1 = cadd(t1, t3); a2 = csub(t0, t2); a3 = csub(t1, t3);
}
#endif
#ifndef __HIP_DEVICE_COMPILE__
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
#endif
// u[r] *= w[r], r = 1 .. R-1 (GPU: two products per statement)
template <int R> PAYNE_HD void twiddle_all(c32* u, const c32* w) {
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll
  for (int r = 1; r + 1 < R; r += 2) {
    pk2 a = to_pk(u[r]), b = to_pk(u[r + 1]);
    pk_cmul2(a, to_pk(w[r]), b, to_pk(w[r + 1]));
    u[r] = un_pk(a); u[r + 1] = un_pk(b);
  }
  if (((R - 1) & 1) != 0) u[R - 1] = cmul(u[R - 1], w[R - 1]);
#else
  for (int r = 1; r < R; ++r) u[r] = cmul(u[r], w[r]);
#endif
}
template <int R> PAYNE_HD void dftR(c32* u) {
  if (R == 8) dft8(u);
  else if (R == 4) dft4(u[0], u[1], u[2], u[3]);
  else dft2(u);
}

// Runtime-geometry pass (any M, p; any thread count).
template <int R>
PAYNE_HD void fft_pass(int tid, int nthr, const c32* __restrict__ src, c32* __restrict__ dst, int M, int p,
                       const c32* __restrict__ tw, int tw_n, bool conj_out) {
  const int nb = M / R;
  for (int i = tid; i < nb; i += nthr) {
    const int k = i & (p - 1);
    c32 u[R];
#pragma unroll
    for (int r = 0; r < R; ++r) u[r] = src[i + r * nb];
    if (p > 1) {
      const int ts = tw_n / (p * R);
#pragma unroll
      for (int r = 1; r < R; ++r) u[r] = cmul(u[r], tw[(k * r) * ts]);
    }
    dftR<R>(u);
    const int j = (i - k) * R + k;
#pragma unroll
    for (int r = 0; r < R; ++r) dst[j + r * p] = conj_out ? cconj(u[r]) : u[r];
  }
}
PAYNE_HD int pass_radix(int M, int p) { int rem = M / p; return rem >= 8 ? 8 : rem; }

// ---------------------------------------------------------------------------
// Four-step form of the same transform for spectra that live in a GLOBAL workspace (payne_post_big_kernel):
// M = 512 B.  Step 1: the B columns x[c + B m] get a 512-point transform each, 16 columns at a time in an LDS
// tile (three radix-8 passes: global -> X -> Y -> global), written as y[512 c + q].  Step 2: for every k the
// B values y[k + 512 m] are multiplied by W_M^(k m) and get a B-point transform, 64 k at a time (two or three
// passes through LDS), written to out[k + 512 q].  Two round trips through memory instead of log8(M):
// the runtime-geometry passes of fft_pass move the whole spectrum once per radix-8 pass.
// Tile layouts: step 1 X/Y[cc * kTileLd + m] (column-major, +1 pad: conflict-free column-strided access);
// step 2 X/Y[row * kTileK + kk].
// ---------------------------------------------------------------------------
constexpr int kTileA = 512;          // sub-transform length of step 1
constexpr int kTileC = 16;           // columns per step-1 tile
constexpr int kTileLd = kTileA + 1;
constexpr int kTileK = 64;           // k values per step-2 tile
PAYNE_HD constexpr int fft_tile_complex() { return kTileC * kTileLd; }          // per LDS buffer (>= kTileK * 128)
PAYNE_HD bool fft_tiled_ok(int M) { const int B = M / kTileA; return M % kTileA == 0 && (B == 32 || B == 64 || B == 128); }

// Accessors: GP / LP / TP are the pointer types Ex::buf / Ex::lds / Ex::twid hand out (address space in the
// type on the device); pairs of adjacent complex values travel as one 16-byte access where the index is even.
#ifdef __HIP_DEVICE_COMPILE__
typedef float f2q __attribute__((ext_vector_type(2)));
typedef float f4q __attribute__((ext_vector_type(4)));
#define PAYNE_Q_LDS __attribute__((address_space(3)))
#define PAYNE_Q_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ c32 q_ld(const PAYNE_Q_GLOBAL f2q* p, size_t i) { const f2q v = p[i]; return {v.x, v.y}; }
__device__ __forceinline__ c32 q_ld(const PAYNE_Q_LDS f2q* p, size_t i) { const f2q v = p[i]; return {v.x, v.y}; }
__device__ __forceinline__ c32 q_ld(PAYNE_Q_GLOBAL f2q* p, size_t i) { const f2q v = p[i]; return {v.x, v.y}; }
__device__ __forceinline__ c32 q_ld(PAYNE_Q_LDS f2q* p, size_t i) { const f2q v = p[i]; return {v.x, v.y}; }
__device__ __forceinline__ void q_st(PAYNE_Q_GLOBAL f2q* p, size_t i, c32 v) { f2q t; t.x = v.x; t.y = v.y; p[i] = t; }
__device__ __forceinline__ void q_st(PAYNE_Q_LDS f2q* p, size_t i, c32 v) { f2q t; t.x = v.x; t.y = v.y; p[i] = t; }
__device__ __forceinline__ void q_ld2(const PAYNE_Q_GLOBAL f2q* p, size_t i, c32& a, c32& b) {
  const f4q v = *reinterpret_cast<const PAYNE_Q_GLOBAL f4q*>(p + i); a = {v.x, v.y}; b = {v.z, v.w};
}
__device__ __forceinline__ void q_ld2(PAYNE_Q_GLOBAL f2q* p, size_t i, c32& a, c32& b) {
  const f4q v = *reinterpret_cast<const PAYNE_Q_GLOBAL f4q*>(p + i); a = {v.x, v.y}; b = {v.z, v.w};
}
__device__ __forceinline__ void q_ld2(PAYNE_Q_LDS f2q* p, size_t i, c32& a, c32& b) {
  const f4q v = *reinterpret_cast<const PAYNE_Q_LDS f4q*>(p + i); a = {v.x, v.y}; b = {v.z, v.w};
}
__device__ __forceinline__ void q_st2(PAYNE_Q_GLOBAL f2q* p, size_t i, c32 a, c32 b) {
  f4q t; t.x = a.x; t.y = a.y; t.z = b.x; t.w = b.y; *reinterpret_cast<PAYNE_Q_GLOBAL f4q*>(p + i) = t;
}
__device__ __forceinline__ void q_st2(PAYNE_Q_LDS f2q* p, size_t i, c32 a, c32 b) {
  f4q t; t.x = a.x; t.y = a.y; t.z = b.x; t.w = b.y; *reinterpret_cast<PAYNE_Q_LDS f4q*>(p + i) = t;
}
#else
inline c32 q_ld(const c32* p, size_t i) { return p[i]; }
inline void q_st(c32* p, size_t i, c32 v) { p[i] = v; }
inline void q_ld2(const c32* p, size_t i, c32& a, c32& b) { a = p[i]; b = p[i + 1]; }
inline void q_st2(c32* p, size_t i, c32 a, c32 b) { p[i] = a; p[i + 1] = b; }
#endif

// step 1, first pass of a tile: global (strided columns, two adjacent columns per thread) -> X.
// SCRUB: the source is the raw ANN row read as (even, odd) pairs: NaN -> 0 (nan_to_num of the shifted flux) on the way.
template <bool SCRUB = false, class GP, class LP>
PAYNE_HD void fft4_s1_load(int tid, int nthr, GP src, LP X, int B, int c0) {
  for (int idx = tid; idx < (kTileC / 2) * 64; idx += nthr) {
    const int cp = idx % (kTileC / 2), i = idx / (kTileC / 2), cc = 2 * cp;
    c32 u[8], v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) q_ld2(src, (size_t)(c0 + cc) + (size_t)B * (i + 64 * r), u[r], v[r]);
    if (SCRUB) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        u[r] = {nan_to_zero(u[r].x), nan_to_zero(u[r].y)};
        v[r] = {nan_to_zero(v[r].x), nan_to_zero(v[r].y)};
      }
    }
    dft8(u);
    dft8(v);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      q_st(X, (size_t)cc * kTileLd + 8 * i + r, u[r]);
      q_st(X, (size_t)(cc + 1) * kTileLd + 8 * i + r, v[r]);
    }
  }
}
// step 1, middle pass (sub-length 8): X -> Y
template <class LP, class TP>
PAYNE_HD void fft4_s1_mid(int tid, int nthr, LP X, LP Y, TP tw, int tw_n, int ncols = kTileC) {
  const int ts = tw_n / 64;
  for (int idx = tid; idx < ncols * 64; idx += nthr) {
    const int i = idx & 63, cc = idx >> 6, k = i & 7;
    c32 u[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) u[r] = q_ld(X, (size_t)cc * kTileLd + i + 64 * r);
#pragma unroll
    for (int r = 1; r < 8; ++r) u[r] = cmul(u[r], q_ld(tw, (size_t)(k * r) * ts));
    dft8(u);
    const int j = (i - k) * 8 + k;
#pragma unroll
    for (int r = 0; r < 8; ++r) q_st(Y, (size_t)cc * kTileLd + j + 8 * r, u[r]);
  }
}
// step 1, last pass (sub-length 64): Y -> global, natural order within the column's 512 outputs; two adjacent
// outputs per thread
// (tile column cc holds column colmap(cc) of the transform; < 0: an empty slot)
struct ColRun { int c0; PAYNE_HD int operator()(int cc) const { return c0 + cc; } };
template <class LP, class GP, class TP, class CM>
PAYNE_HD void fft4_s1_store(int tid, int nthr, LP Y, GP dst, TP tw, int tw_n, CM colmap, int ncols = kTileC) {
  const int ts = tw_n / 512;
  for (int idx = tid; idx < ncols * 32; idx += nthr) {
    const int i = 2 * (idx & 31), cc = idx >> 5;
    const int c0 = colmap(cc) - cc;                        // (so that c0 + cc below is the mapped column)
    if (c0 + cc < 0) continue;
    c32 u[8], v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { u[r] = q_ld(Y, (size_t)cc * kTileLd + i + 64 * r); v[r] = q_ld(Y, (size_t)cc * kTileLd + i + 1 + 64 * r); }
#pragma unroll
    for (int r = 1; r < 8; ++r) { u[r] = cmul(u[r], q_ld(tw, (size_t)(i * r) * ts)); v[r] = cmul(v[r], q_ld(tw, (size_t)((i + 1) * r) * ts)); }
    dft8(u);
    dft8(v);
#pragma unroll
    for (int r = 0; r < 8; ++r) q_st2(dst, (size_t)(c0 + cc) * kTileA + i + 64 * r, u[r], v[r]);
  }
}
// step 2, first pass of a tile: global (stride 512, two adjacent k per thread) x W_M^(k m) -> X, radix 8
template <class GP, class LP, class TP>
PAYNE_HD void fft4_s2_load(int tid, int nthr, GP src, LP X, int B, int k0, TP tw, int tw_n) {
  const int nb = B / 8, tsM = tw_n / (kTileA * B);
  for (int idx = tid; idx < (kTileK / 2) * nb; idx += nthr) {
    const int kk = 2 * (idx % (kTileK / 2)), i = idx / (kTileK / 2), k = k0 + kk;
    c32 u[8], v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int m = i + nb * r;
      q_ld2(src, (size_t)k + (size_t)kTileA * m, u[r], v[r]);
      u[r] = cmul(u[r], q_ld(tw, (size_t)(k * m) * tsM));
      v[r] = cmul(v[r], q_ld(tw, (size_t)((k + 1) * m) * tsM));
    }
    dft8(u);
    dft8(v);
#pragma unroll
    for (int r = 0; r < 8; ++r) q_st2(X, (size_t)(8 * i + r) * kTileK + kk, u[r], v[r]);
  }
}
// step 2, a later pass (sub-length p, radix R): X -> Y (LDS) or, as the last pass, -> global (stride 512)
template <int R, bool TO_GLOBAL, class LP, class GP, class TP>
PAYNE_HD void fft4_s2_pass(int tid, int nthr, LP X, LP Y, GP gdst, int B, int p, int k0, TP tw, int tw_n, bool conj_out) {
  const int nb = B / R, ts = tw_n / (p * R);
  for (int idx = tid; idx < (kTileK / 2) * nb; idx += nthr) {
    const int kk = 2 * (idx % (kTileK / 2)), i = idx / (kTileK / 2), k = i & (p - 1);
    c32 u[R], v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) q_ld2(X, (size_t)(i + nb * r) * kTileK + kk, u[r], v[r]);
#pragma unroll
    for (int r = 1; r < R; ++r) { const c32 w = q_ld(tw, (size_t)(k * r) * ts); u[r] = cmul(u[r], w); v[r] = cmul(v[r], w); }
    dftR<R>(u);
    dftR<R>(v);
    const int j = (i - k) * R + k;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int q = j + r * p;
      if (TO_GLOBAL) {
        if (conj_out) q_st2(gdst, (size_t)(k0 + kk) + (size_t)kTileA * q, cconj(u[r]), cconj(v[r]));
        else q_st2(gdst, (size_t)(k0 + kk) + (size_t)kTileA * q, u[r], v[r]);
      } else {
        q_st2(Y, (size_t)q * kTileK + kk, u[r], v[r]);
      }
    }
  }
}
// Compile-time geometry ("plan") of an M-point FFT on kPostThreads threads.
// Its twiddles are stored PASS-ORDERED: for every pass with sub-length P > 1 and radix R, the powers w^r = exp(-2 pi i k r/(P R))
// that are READ -- r = 1, 2, 4 for radix 8 (w^3 = w w^2, w^5 = w w^4, w^6 = w^2 w^4, w^7 = w^3 w^4 are products: one or two more
// roundings, four packed multiplies a butterfly), r = 1 for radix 4 (w^2 = w w, w^3 = w^2 w) and radix 2 --
// twf[off(P) + j P + k], j = 0 .. plan_tw_rows(R) - 1  -- the lanes of a wave read consecutive k, so the LDS reads are
// conflict-free (a plain full-circle table is read with stride 2M/(P R): 8- to 16-way bank conflicts) -- followed by the M/2
// factors exp(-2 pi i k/2M) of the real-FFT split.  (All R - 1 powers stored: 24.5 KB at 2048 points, which every workgroup
// pulls from L2 at its start -- with the row, 82 KB per CU at ~22 B/clk --, and 15 instead of 11 LDS reads per radix-8 butterfly;
// 14 KB this way.)
// radix: 8 while that still gives every thread a butterfly (M/8 >= threads), else 4, else 2
constexpr int plan_radix(int M, int P) {
  return (M / P >= 8 && M / 8 >= kFftThreads) ? 8 : ((M / P >= 4) ? 4 : 2);
}
constexpr int plan_tw_rows(int R) { return R == 8 ? 3 : 1; }
constexpr int plan_offset(int M, int P) {
  int off = 0, p = 1;
  while (p < P) { const int r = plan_radix(M, p); if (p > 1) off += plan_tw_rows(r) * p; p *= r; }
  return off;
}
constexpr int plan_total(int M) { return plan_offset(M, M); }
constexpr int plan_table_len(int M) { return plan_total(M) + M / 2; }

// Complex loads/stores through a pointer whose address space is part of its type: the
// out-of-line FFT body is handed LDS (or global) pointers explicitly, otherwise its
// accesses compile to flat_load/flat_store (slower, and counted on both wait counters).
#ifdef __HIP_DEVICE_COMPILE__
typedef float f2v __attribute__((ext_vector_type(2)));
#define PAYNE_AS_LDS __attribute__((address_space(3)))
#define PAYNE_AS_GLOBAL __attribute__((address_space(1)))
template <class Ptr> __device__ __forceinline__ c32 ldc(Ptr p, int i) { const f2v v = p[i]; return {v.x, v.y}; }
template <class Ptr> __device__ __forceinline__ void stc(Ptr p, int i, c32 v) { f2v t; t.x = v.x; t.y = v.y; p[i] = t; }
// An LDS read the compiler must not pair with its neighbour: two ds_read_b64 take 2 + 2 LDS cycles, the ds_read2_b64 /
// ds_read2st64_b64 it merges them into takes 8 (MI355X_MICROARCH.md, LDS table) -- a volatile access is left alone.
__device__ __forceinline__ c32 ldc1(const PAYNE_AS_LDS f2v* p, int i) { const f2v v = *(const volatile PAYNE_AS_LDS f2v*)(p + i); return {v.x, v.y}; }
__device__ __forceinline__ c32 ldc1(PAYNE_AS_LDS f2v* p, int i) { const f2v v = *(const volatile PAYNE_AS_LDS f2v*)(p + i); return {v.x, v.y}; }
__device__ __forceinline__ c32 ldc1(const PAYNE_AS_GLOBAL f2v* p, int i) { const f2v v = p[i]; return {v.x, v.y}; }
__device__ __forceinline__ c32 ldc1(PAYNE_AS_GLOBAL f2v* p, int i) { const f2v v = p[i]; return {v.x, v.y}; }
__device__ __forceinline__ c32 ld1(const PAYNE_AS_LDS f2v* p, int i) { return ldc1(p, i); }
__device__ __forceinline__ c32 ld1(PAYNE_AS_LDS f2v* p, int i) { return ldc1(p, i); }
__device__ __forceinline__ c32 ld1(const PAYNE_AS_GLOBAL f2v* p, int i) { return ldc1(p, i); }
__device__ __forceinline__ c32 ld1(PAYNE_AS_GLOBAL f2v* p, int i) { return ldc1(p, i); }
__device__ __forceinline__ void st1(PAYNE_AS_LDS f2v* p, int i, c32 v) { stc(p, i, v); }
__device__ __forceinline__ void st1(PAYNE_AS_GLOBAL f2v* p, int i, c32 v) { stc(p, i, v); }
#else
inline c32 ldc1(const c32* p, int i) { return p[i]; }
inline c32 ldc(const c32* p, int i) { return p[i]; }
inline void stc(c32* p, int i, c32 v) { p[i] = v; }
#endif

// Layout of the buffer BETWEEN two passes.  A pass with sub-length P and radix R writes, per
// lane i (k = i mod P, g = i div P), the R elements j + r P with j = g P R + k: for P < 32 the
// lanes of a half-wave land in a few banks (P = 1: lane stride R complex = 64 B -> 4 distinct
// 8-byte slots; P = 8: 64-byte runs 512 B apart -> 4-way conflicts).  Padding the intermediate
// buffer makes those writes conflict-free while the next pass still reads consecutive lanes
// from consecutive (or once-shifted) addresses:
//   P R < 32 : phys(j) = j + (j >> 5)               one complex per 32 (the R-runs of 4 lanes rotate by 8 B)
//   P < 32   : phys(j) = j + P (j div P R)          group stride P R + P  ==  P (mod 32 complex = 256 B)
//   P >= 32  : phys(j) = j
// The first pass reads and the last pass writes the plain layout.  Both padded forms are additive
// over multiples of their period, so the per-element offsets stay compile-time constants.
template <int PW, int RW>
PAYNE_HD int fft_lay(int j) {
  if constexpr (PW == 0 || PW >= 32) return j;
  else if constexpr (PW * RW < 32) return j + (j >> 5);
  else return j + PW * (j / (PW * RW));
}
// sub-length of the pass that precedes the pass with sub-length P (P > 1)
constexpr int plan_prev_p(int M, int P) {
  int p = 1;
  while (p * plan_radix(M, p) < P) p *= plan_radix(M, p);
  return p;
}
// floats each of the two ping-pong buffers needs for n-point spectra (M = n/2 complex + padding <= M/R)
PAYNE_HD constexpr int fft_buf_floats(int n) { return n + ((n / 2) / 8 >= kFftThreads ? n / 8 : n / 4); }

// One pass: M points, sub-length P, NT threads; `sign` = 0x80000000 conjugates the output.
// SP/DP/TP: pointer types as produced by Ex::buf / Ex::twid.
// `edge` (last pass only): the result is a real spectrum packed as (even, odd) pairs; apply
// spec[0] = spec[1], spec[-1] = spec[-2] (ystpred.py:223-224) on the way out.
template <int R, int M, int P, int NT, class SP, class DP, class TP>
PAYNE_HD void fft_pass_fixed(int tid, SP src, DP dst, TP twf, unsigned sign, bool edge = false) {
  constexpr int NB = M / R, OFF = plan_offset(M, P);
  constexpr int PI = (P > 1) ? plan_prev_p(M, P) : 0;              // layout we read: written by pass (PI, RI)
  constexpr int RI = (P > 1) ? plan_radix(M, PI) : 0;
  constexpr bool last = (P * R >= M);
  constexpr int PO = last ? 0 : P;                                 // layout we write
  static_assert(!last || P >= 32 || M < 64, "the last pass must write the plain layout");
  static_assert(NB % 32 == 0 || PI == 0 || PI >= 32, "padded reads need NB to be a multiple of the pad period");
  // LDS stores are banked in groups of SIXTEEN lanes over 32 dwords (MI355X_MICROARCH.md, LDS table; tools/exp/lds_floor.hip reads
  // 4 conflict cycles on every store of the plain lane order here).  The first radix-8 pass writes slot 8 i + (i >> 2) + r: sixteen
  // consecutive butterflies land on eight slots mod 16.  With butterfly i = [m2 m1 m0 h b] on lane [h m2 m1 m0 b] a group holds
  // i = 4 m + 2 h + b, m = 0..7, b = 0..1: slots 8 b + m -- all sixteen; the reads stay a permutation of 32 consecutive slots.
  int lane_i = tid;
  if constexpr (P == 1 && R == 8 && !last) lane_i = (tid & ~31) | ((tid & 14) << 1) | ((tid >> 3) & 2) | (tid & 1);
#pragma unroll
  for (int i0 = 0; i0 < NB; i0 += NT) {
    const int i = i0 + lane_i;
    if ((NB % NT) != 0 && i >= NB) break;
    const int k = i & (P - 1);
    const int ib = fft_lay<PI, RI>(i);
    c32 u[R];
#pragma unroll
    for (int r = 0; r < R; ++r) u[r] = ldc1(src, ib + fft_lay<PI, RI>(r * NB));
    if (P > 1) {
      c32 w[R];
      w[1] = ldc1(twf, OFF + k);
      if constexpr (R == 8) {
        w[2] = ldc1(twf, OFF + P + k); w[4] = ldc1(twf, OFF + 2 * P + k);
        w[3] = cmul(w[1], w[2]); w[5] = cmul(w[1], w[4]); w[6] = cmul(w[2], w[4]); w[7] = cmul(w[3], w[4]);
      } else if constexpr (R == 4) {
        w[2] = cmul(w[1], w[1]); w[3] = cmul(w[2], w[1]);
      }
      twiddle_all<R>(u, w);
    }
    dftR<R>(u);
    const int j = fft_lay<PO, R>((i - k) * R + k);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      c32 v = u[r];
      union { float f; unsigned b; } cv;
      cv.f = v.y; cv.b ^= sign; v.y = cv.f;
      if constexpr (last) {
        if (r == 0 && edge && i == 0) v.x = v.y;                    // element 0     = (spec[0], spec[1])
        if (r == R - 1 && edge && i == NB - 1) v.y = v.x;           // element M - 1 = (spec[n-2], spec[n-1])
      }
      stc(dst, j + r * P, v);
    }
  }
}

// ---------------------------------------------------------------------------
// Tapers.
// ---------------------------------------------------------------------------
// sb(ub) of smoothing.py:616-617 evaluated directly in fp64 (host table build, far tail).
// Cold on the GPU (only beyond the table): kept out of line so its fp64 j1/sincos bodies
// are not replicated into every taper call site.
PAYNE_HD_COLD double vsini_sb_exact(double ub) {
  double s, c;
#ifdef __HIP_DEVICE_COMPILE__
  sincos(ub, &s, &c);
#else
  s = sin(ub); c = cos(ub);
#endif
  double u2 = ub * ub;
  return j1(ub) / ub - 3.0 * c / (2.0 * u2) + 3.0 * s / (2.0 * (u2 * ub));
}
// smoothing.py:612-620 for bin k: 4-point Lagrange interpolation in the host-built table
// (fp64-accurate values stored as fp32; the factor multiplies Fourier amplitudes of the
// SHIFTED spectrum, so 6e-8 on it is ~1e-9 in flux); sb is even in u; sb[0] = 1.
// Branch-free (loads from a clamped index): a guarded load costs a branch and a wait and
// serialises the caller's unrolled evaluations.  `far` reports bins beyond the table (or a
// NaN argument); the caller re-evaluates those with vsini_sb_exact.
// (tab[-1] exists and equals tab[1] -- sb is even --: the four values are ONE 16-byte access at tab + i - 1, no select on i = 0.)
// The table position is split like the pixel positions (magic_locate): t + kPosMagic has the integer part in its high dword and the
// fraction, in units of 2^-32, in its low dword -- an fma, a mask and one conversion where floor / convert / subtract / convert cost
// four fp64-rate instructions per bin, a quarter of the on-chip rotation stage's vector instructions with the rest of this function.
typedef float tap2 __attribute__((vector_size(8)));      // two bins side by side: the pair (k, M - k) is always evaluated together
struct TapPos { int i; float f; };
PAYNE_HD TapPos vsini_tab_pos(int tab_n, double vs_c64, int k, bool& far) {
  const double tm = fma((double)k, vs_c64, kPosMagic);   // u / kVsTabStep (+ magic); 0 <= t < 2^19
  const bool in = tm < kPosMagic + (double)(tab_n - 3);  // false for NaN
  far = far || !in;
  union { double d; unsigned long long u; } cv;
  cv.d = in ? tm : kPosMagic;
  TapPos p;
  p.i = (int)((unsigned)(cv.u >> 32) & 0x7FFFFu);
  p.f = (float)(unsigned)cv.u * 2.3283064365386963e-10f;  // 2^-32
  return p;
}
// 4-point Lagrange weights of two bins at once (packed fp32 on the GPU)
PAYNE_HD void vsini_taper_fast2(const float* __restrict__ tab, int tab_n, double vs_c64, int ka, int kb, bool& far, float& ta_, float& tb_) {
  const TapPos pa = vsini_tab_pos(tab_n, vs_c64, ka, far), pb = vsini_tab_pos(tab_n, vs_c64, kb, far);
  const float* __restrict__ qa = tab + pa.i - 1;
  const float* __restrict__ qb = tab + pb.i - 1;
  const tap2 ym = {qa[0], qb[0]}, y0 = {qa[1], qb[1]}, y1 = {qa[2], qb[2]}, y2 = {qa[3], qb[3]};
  const tap2 f = {pa.f, pb.f};
  const tap2 fm1 = f - 1.0f, fm2 = f - 2.0f, fp1 = f + 1.0f;
  const tap2 a = f * fm1, b = fp1 * fm2;                  // f (f - 1), (f + 1)(f - 2)
  const tap2 v = (a * fm2 * (-1.0f / 6.0f)) * ym + (b * fm1 * 0.5f) * y0 + (b * f * (-0.5f)) * y1 + (a * fp1 * (1.0f / 6.0f)) * y2;
  ta_ = ka == 0 ? 1.0f : v[0];
  tb_ = kb == 0 ? 1.0f : v[1];
}
PAYNE_HD float vsini_taper_fast(const float* __restrict__ tab, int tab_n, double vs_c64, int k, bool& far) {
  float a, b;
  vsini_taper_fast2(tab, tab_n, vs_c64, k, k, far, a, b);
  return a;
}
// smoothing.py:598-599: exp(-2 pi^2 sigma^2 ss^2), ss = k/(n dv), as exp2(c2 k^2)
PAYNE_HD float gauss_taper(float g_c2, int k) {
  const float kf = (float)k;
#ifdef __HIP_DEVICE_COMPILE__
  return __builtin_amdgcn_exp2f(g_c2 * (kf * kf));      // v_exp_f32
#else
  return exp2f(g_c2 * (kf * kf));
#endif
}

struct TaperArgs {
  const float* vs_tab; int vs_tab_n; double vs_c64, vs_c;   // vsini
  float g_c2;                                               // gauss
};
template <bool VSINI> PAYNE_HD float taper_at(const TaperArgs& a, int k, bool& far) {
  return VSINI ? vsini_taper_fast(a.vs_tab, a.vs_tab_n, a.vs_c64, k, far) : gauss_taper(a.g_c2, k);
}
// the two bins of a conjugate pair
template <bool VSINI> PAYNE_HD void taper_at2(const TaperArgs& a, int ka, int kb, bool& far, float& ta_, float& tb_) {
  if (VSINI) vsini_taper_fast2(a.vs_tab, a.vs_tab_n, a.vs_c64, ka, kb, far, ta_, tb_);
  else { ta_ = gauss_taper(a.g_c2, ka); tb_ = gauss_taper(a.g_c2, kb); }
}
// exact re-evaluation of a bin the table does not cover (rare: u >= 256 or NaN)
PAYNE_HD float taper_far(const TaperArgs& a, int k, float fast) {
  const double t = (double)k * a.vs_c64;
  return (k != 0 && !(t < (double)(a.vs_tab_n - 3))) ? (float)vsini_sb_exact((double)k * a.vs_c) : fast;
}

// The convolution's middle step for ONE conjugate pair: zk = Z[k], zmk = Z[M - k] (not conjugated), w = exp(-2 pi i k/2M),
// tkg / tmg = taper(k) g, taper(M - k) g with g = 1/(4M).  Out: Y[k], Y[M - k] (see rfft_taper_phase).
#ifdef __HIP_DEVICE_COMPILE__
// thirteen packed instructions; T = (tkg, tmg); zk / zmk are replaced by Y[k] / Y[M - k]
#define PAYNE_TAPER_PAIR_ASM(ZK, ZM, W, T, A, D, X)                                                                           \
      "v_pk_add_f32 " A ", " ZK ", " ZM " neg_hi:[0,1]\n\t"                                 /* A = zk + conj(zmk) */          \
      "v_pk_add_f32 " D ", " ZK ", " ZM " neg_lo:[0,1]\n\t"                                 /* D = zk - conj(zmk) */          \
      "v_pk_mul_f32 " X ", " W ", " D " op_sel:[1,0] op_sel_hi:[1,1]\n\t"                   /* (w.y D.x, w.y D.y) */          \
      "v_pk_fma_f32 " X ", " W ", " D ", " X " op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]\n\t"   /* C = w (-i D) */       \
      "v_pk_add_f32 " D ", " A ", " X " neg_lo:[0,1] neg_hi:[0,1]\n\t"                      /* Q = A - C */                   \
      "v_pk_add_f32 " A ", " A ", " X "\n\t"                                                /* P = A + C */                   \
      "v_pk_mul_f32 " D ", " D ", " T " op_sel:[0,1] op_sel_hi:[1,1]\n\t"                   /* S2 = Q T.y */                  \
      "v_pk_fma_f32 " X ", " A ", " T ", " D " op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"   /* F = P T.x - S2 */ \
      "v_pk_fma_f32 " A ", " A ", " T ", " D " op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"        /* E = P T.x + S2 */              \
      "v_pk_mul_f32 " D ", " W ", " X " op_sel:[1,0] op_sel_hi:[1,1]\n\t"                   /* (w.y F.x, w.y F.y) */          \
      "v_pk_fma_f32 " D ", " W ", " X ", " D " op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[0,1,0]\n\t"   /* iO = i conj(w) F */   \
      "v_pk_add_f32 " ZK ", " A ", " D " neg_hi:[1,1]\n\t"                                  /* Y[k] = conj(E + iO) */         \
      "v_pk_add_f32 " ZM ", " A ", " D " neg_lo:[0,1] neg_hi:[0,1]"                          /* Y[M-k] = E - iO */
__device__ __forceinline__ void pk_taper_pair(pk2& zk, pk2& zmk, pk2 w, pk2 T) {
  pk2 A, D, X;
  asm(PAYNE_TAPER_PAIR_ASM("%0", "%1", "%5", "%6", "%2", "%3", "%4")
      : "+v"(zk), "+v"(zmk), "=&v"(A), "=&v"(D), "=&v"(X) : "v"(w), "v"(T));
}
__device__ __forceinline__ void taper_pair(c32 zk, c32 zmk, c32 w, float tkg, float tmg, c32& yk, c32& ymk) {
  pk2 T, a = to_pk(zk), b = to_pk(zmk);
  T.x = tkg; T.y = tmg;
  pk_taper_pair(a, b, to_pk(w), T);
  yk = un_pk(a); ymk = un_pk(b);
}
#else
PAYNE_HD void taper_pair(c32 zk, c32 zmk, c32 w, float tkg, float tmg, c32& yk, c32& ymk) {
  const c32 zm = cconj(zmk);
  const c32 A = cadd(zk, zm);
  const c32 C = cmul(w, cmul_negi(csub(zk, zm)));
  const c32 S1 = cscale(cadd(A, C), tkg), S2 = cscale(csub(A, C), tmg);
  const c32 E = cadd(S1, S2);
  const c32 iO = cmul_posi(cmul(cconj(w), csub(S1, S2)));
  yk = cconj(cadd(E, iO));
  ymk = csub(E, iO);
}
#endif

// The two neighbours base[k], base[k + 1] of U gathered points: two 4-byte reads each (the compiler pairs them in one ds_read2_b32).
// (Measured and dropped: ONE 8-byte read at the 4-byte-aligned address -- the hardware serves it and SQ_LDS_BANK_CONFLICT reads 0
//  where the 4-byte reads of lanes 1.14 words apart read 2 cycles each, but it stalls on alignment instead: the post kernel went
//  from 13.4 to 19.2 us.  tools/exp/lds_floor.hip, NOTES R5.)
template <int U>
PAYNE_HD void ld_pairs(const float* __restrict__ base, const unsigned (&k)[U], float (&a)[U], float (&b)[U]) {
#pragma unroll
  for (int q = 0; q < U; ++q) { a[q] = base[k[q]]; b[q] = base[k[q] + 1]; }
}
PAYNE_HD c32 ld1(const c32* p, int i) { return p[i]; }
PAYNE_HD c32 ld1(c32* p, int i) { return p[i]; }
PAYNE_HD void st1(c32* p, int i, c32 v) { p[i] = v; }
// Middle step of a real convolution done with a half-length complex FFT.
// In: Z = FFT_M(z), z[n] = s[2n] + i s[2n+1].  Out (in place): Y with
// FFT_M(Y) = conj(z'), z'[n] = s'[2n] + i s'[2n+1], s' = irfft(rfft(s) * taper).
// A thread owns conjugate pairs (k, M-k), k = 1..M/2-1, PU at a time (all loads of the PU
// pairs are issued before any store: the pairs are disjoint, so this is safe in place);
// the two self-conjugate bins k = 0 and k = M/2 go to the last two threads.
// tw_step = (twiddle table length)/(2M): exp(-2 pi i k/2M) = tw[k*tw_step].
// (ZP / TP: plain pointers, or -- the kernel's compile-time geometry -- the typed LDS / global pointers of Ex::buf / Ex::twid, whose
//  reads are then single ds_read_b64: ld1 / st1 above)
template <bool VSINI, int PU = 4, class ZP = c32*, class TP = const c32*>
PAYNE_HD void rfft_taper_phase(int tid, int nthr, ZP Z, int M, TP tw, int tw_step,
                               const TaperArgs& ta) {
  const float invM = 1.0f / (float)M, g = 0.25f * invM;
  const int npair = M / 2 - 1;                        // k = 1 .. M/2-1
  // The two self-conjugate bins (0 with M, and M/2) need taper(M) and taper(M/2): they ride in the
  // one slot of the loop that falls on k0 == M/2 (past the last pair), so their table loads are in
  // flight with everybody else's instead of forming a second, serial round trip at the end.
  bool special_done = false;
  for (int base = tid; base < npair; base += PU * nthr) {
    c32 zk[PU], zm[PU], w[PU];
    float tk[PU], tm[PU];
    bool far = false;
#pragma unroll
    for (int q = 0; q < PU; ++q) {                     // loads from clamped indices: no branches here
      const int k0 = 1 + base + q * nthr;
      const bool pair = k0 <= npair;
      const int k = pair ? k0 : npair;
      zk[q] = ld1(Z, k); zm[q] = ld1(Z, M - k); w[q] = ld1(tw, k * tw_step);
      taper_at2<VSINI>(ta, pair ? k : M / 2, pair ? M - k : M, far, tk[q], tm[q]);
    }
    if (VSINI && far) {
#pragma unroll
      for (int q = 0; q < PU; ++q) {
        const int k0 = 1 + base + q * nthr;
        const bool pair = k0 <= npair;
        const int k = pair ? k0 : npair;
        tk[q] = taper_far(ta, pair ? k : M / 2, tk[q]);
        tm[q] = taper_far(ta, pair ? M - k : M, tm[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < PU; ++q) {
      const int k = 1 + base + q * nthr;
      c32 yk, ymk;
      taper_pair(zk[q], zm[q], w[q], tk[q] * g, tm[q] * g, yk, ymk);
      if (k <= npair) {
        st1(Z, k, yk);
        st1(Z, M - k, ymk);
      } else if (k == M / 2) {                         // tk = taper(M/2), tm = taper(M); taper(0) = 1
        const c32 z0 = ld1(Z, 0), zh = ld1(Z, M / 2);
        const float x0 = z0.x + z0.y, xm = tm[q] * (z0.x - z0.y);
        st1(Z, 0, c32{0.5f * (x0 + xm) * invM, -0.5f * (x0 - xm) * invM});
        st1(Z, M / 2, cscale(cconj(zh), tk[q] * invM));
        special_done = true;
      }
    }
  }
  // does some thread's loop contain the slot k0 == M/2 ?  (uniform arithmetic; true for every
  // power-of-two M >= 2 PU nthr, e.g. the kernel's 2048 points on 256 threads)
  bool covered = false;
  for (int q = 1; q < PU; ++q) {
    const int b = npair - q * nthr;
    covered = covered || (b >= 0 && (b % (PU * nthr)) < nthr);
  }
  (void)special_done;
  if (!covered) {
    if (tid == nthr - 1) {                               // k = 0 with k = M (real bins X[0], X[M])
      bool far = false;
      const float t0 = taper_at<VSINI>(ta, 0, far);
      float tM = taper_at<VSINI>(ta, M, far);
      if (VSINI && far) tM = taper_far(ta, M, tM);
      const c32 z0 = ld1(Z, 0);
      const float x0 = t0 * (z0.x + z0.y), xm = tM * (z0.x - z0.y);
      st1(Z, 0, c32{0.5f * (x0 + xm) * invM, -0.5f * (x0 - xm) * invM});
    }
    if (tid == (nthr > 1 ? nthr - 2 : 0) && M >= 2) {    // k = M/2 (self-conjugate)
      const int k = M / 2;
      bool far = false;
      float th = taper_at<VSINI>(ta, k, far);
      if (VSINI && far) th = taper_far(ta, k, th);
      st1(Z, k, cscale(cconj(ld1(Z, k)), th * invM));
    }
  }
}

template <bool VSINI> PAYNE_HD void taper_full2(const TaperArgs& ta, int ka, int kb, float& ta_, float& tb_) {
  bool far = false;
  taper_at2<VSINI>(ta, ka, kb, far, ta_, tb_);
  if (VSINI && far) { ta_ = taper_far(ta, ka, ta_); tb_ = taper_far(ta, kb, tb_); }
}
template <bool VSINI> PAYNE_HD float taper_full(const TaperArgs& ta, int k) {
  bool far = false;
  float t = taper_at<VSINI>(ta, k, far);
  if (VSINI && far) t = taper_far(ta, k, t);
  return t;
}

// ---------------------------------------------------------------------------
// Rows handed over in the frequency domain (PostTables::raw_freq) come in PAIR layout: slot j (16 bytes) = (Z[j], Z[M - j]) for
// j = 1 .. M/2 - 1, slot 0 = (Z[0], Z[M/2]) (host_tables.hpp freq_rows).  The conjugate pair the middle step combines arrives in
// ONE load, and the step is done on the way from global memory to LDS: no commit of the row, no barrier, no phase of its own.
// taper_slot: the two factors of a slot with the normalisation rfft_taper_phase applies (slot 0: taper(M/2)/M and taper(M)); they
// depend on theta[5] alone.  (Made ahead by riders of the hidden-layer launch they cost that launch 1.6 us and saved this kernel
// 0.3: NOTES R4.14.)
PAYNE_HD TaperArgs vsini_taper_args(const PostTables& T, double vrot) {
  TaperArgs ta{};
  ta.vs_tab = T.vs_tab; ta.vs_tab_n = T.vs_tab_n;
  // u_k = 2 pi sigma k/(n dv) (smoothing.py:612-614, :297); a candidate that does not rotate: u = 0, the taper is 1 in every bin
  // (sqrt(vrot^2 - 0) of smoothing.py:297 IS |vrot|: a correctly rounded square root of a correctly rounded square gives the
  //  magnitude back exactly; the fp64 square root is a twenty-instruction Newton chain in front of every workgroup's table look-ups)
  ta.vs_c = (vrot != 0.0) ? (2.0 * kPi * fabs(vrot)) * T.vs_val : 0.0;
  ta.vs_c64 = ta.vs_c * (1.0 / kVsTabStep);
  return ta;
}
PAYNE_HD void taper_slot(const TaperArgs& ta, int M, int j, float& a, float& b) {
  const float invM = 1.0f / (float)M, g = 0.25f * invM;
  float ta_, tb_;
  taper_full2<true>(ta, j ? j : M / 2, j ? M - j : M, ta_, tb_);
  if (j) { a = ta_ * g; b = tb_ * g; }
  else { a = ta_ * invM; b = tb_; }
}
template <int SU> struct SlotRegs { float z[SU][4]; c32 w[SU]; float t[SU][2]; };
// `w`: exp(-2 pi i j/2M), j < M/2, in GLOBAL memory (the kernel's LDS copy of the table is still on its way in this phase)
template <int SU>
PAYNE_HD void slots_issue(int tid, int nthr, int M, const float* __restrict__ row, const c32* __restrict__ w, SlotRegs<SU>& R) {
  const int ns = M / 2;
#pragma unroll
  for (int q = 0; q < SU; ++q) {                     // clamped index: unconditional loads
    const int j0 = tid + q * nthr, j = j0 < ns ? j0 : ns - 1;
#ifdef __HIP_DEVICE_COMPILE__
    R.z[q][0] = __builtin_nontemporal_load(&row[4 * j]); R.z[q][1] = __builtin_nontemporal_load(&row[4 * j + 1]);
    R.z[q][2] = __builtin_nontemporal_load(&row[4 * j + 2]); R.z[q][3] = __builtin_nontemporal_load(&row[4 * j + 3]);
#else
    R.z[q][0] = row[4 * j]; R.z[q][1] = row[4 * j + 1]; R.z[q][2] = row[4 * j + 2]; R.z[q][3] = row[4 * j + 3];
#endif
    R.w[q] = w[j];
  }
}
// `scrub`: NaN -> 0 first (the rotating branch's nan_to_num; a row is all NaN or not at all)
// The tapers of a thread's SU slots when every bin is inside the table (the caller has checked the thread's LARGEST bin, M - tid,
// or M for thread 0: the table position grows with the bin): no per-bin range test, no selects, the position of bin M - j from
// the position of bin j by one subtraction and that of the next slot by one addition (positions carry 2^-32 of a table step;
// three roundings instead of one move them by < 1e-9 step).
template <int SU>
PAYNE_HD void taper_slots_fast(const TaperArgs& ta, int tid, int nthr, int M, SlotRegs<SU>& R) {
  const int ns = M / 2;
  const float invM = 1.0f / (float)M, g = 0.25f * invM;
  const double tmM = fma((double)M, ta.vs_c64, 2.0 * kPosMagic), dq = (double)nthr * ta.vs_c64;
  double tma = fma((double)(tid ? tid : M / 2), ta.vs_c64, kPosMagic);     // slot 0: bins M/2 and M
#pragma unroll
  for (int q = 0; q < SU; ++q) {
    const int j0 = tid + q * nthr;
    union { double d; unsigned long long u; } ca, cb;
    ca.d = (j0 < ns) ? tma : fma((double)(ns - 1), ta.vs_c64, kPosMagic);  // (a slot past the end: the last one again, never stored)
    cb.d = tmM - ca.d;
    if (q == 0 && tid == 0) cb.d = tmM - kPosMagic;                        // bin M itself
    const float* __restrict__ qa = ta.vs_tab + (int)((unsigned)(ca.u >> 32) - kPosMagicHi) - 1;
    const float* __restrict__ qb = ta.vs_tab + (int)((unsigned)(cb.u >> 32) - kPosMagicHi) - 1;
    const tap2 ym = {qa[0], qb[0]}, y0 = {qa[1], qb[1]}, y1 = {qa[2], qb[2]}, y2 = {qa[3], qb[3]};
    const tap2 f = {(float)(unsigned)ca.u * 2.3283064365386963e-10f, (float)(unsigned)cb.u * 2.3283064365386963e-10f};
    const tap2 fm1 = f - 1.0f, fm2 = f - 2.0f, fp1 = f + 1.0f;
    const tap2 a = f * fm1, b = fp1 * fm2;
    const tap2 v = (a * fm2 * (-1.0f / 6.0f)) * ym + (b * fm1 * 0.5f) * y0 + (b * f * (-0.5f)) * y1 + (a * fp1 * (1.0f / 6.0f)) * y2;
    const bool first = (q == 0 && tid == 0);
    R.t[q][0] = v[0] * (first ? invM : g);
    R.t[q][1] = first ? v[1] : v[1] * g;
    tma = (q == 0 && tid == 0) ? fma((double)nthr, ta.vs_c64, kPosMagic) : tma + dq;
  }
}
template <int SU>
PAYNE_HD void slots_tapers(int tid, int nthr, int M, SlotRegs<SU>& R, const TaperArgs& ta) {
  const int ns = M / 2;
  // is the thread's largest bin (M - tid, M for thread 0) inside the table?  (false for a NaN rotation: the general look-up
  // produces the NaN tapers)
  const bool in_tab = fma((double)(M - tid), ta.vs_c64, kPosMagic) < kPosMagic + (double)(ta.vs_tab_n - 3);
  if (!wave_any(!in_tab)) taper_slots_fast<SU>(ta, tid, nthr, M, R);
  else {
#pragma unroll
    for (int q = 0; q < SU; ++q) {                   // (all the table loads before any of the pairs)
      const int j0 = tid + q * nthr, j = j0 < ns ? j0 : ns - 1;
      taper_slot(ta, M, j, R.t[q][0], R.t[q][1]);
    }
  }
}
template <int SU, class YP>
PAYNE_HD void slots_store(int tid, int nthr, int M, SlotRegs<SU>& R, YP Y, bool scrub) {
  const int ns = M / 2;
  const float invM = 1.0f / (float)M;
  // nan_to_num of the rotating branch: a row in the frequency domain is NaN in every bin or in none (any NaN pixel of the spectrum
  // reaches every bin of its transform), so the test is one value and the branch is the same for the whole workgroup
  const bool rownan = scrub && wave_any(R.z[0][0] != R.z[0][0]);
#pragma unroll
  for (int q = 0; q < SU; ++q) {
    const int j = tid + q * nthr;
    if (j >= ns) break;
    c32 zk{R.z[q][0], R.z[q][1]}, zm{R.z[q][2], R.z[q][3]};
    if (rownan) { zk = c32{0.f, 0.f}; zm = c32{0.f, 0.f}; }
    if (j) {
      c32 yk, ymk;
      taper_pair(zk, zm, R.w[q], R.t[q][0], R.t[q][1], yk, ymk);
      st1(Y, j, yk);
      st1(Y, M - j, ymk);
    } else {                                         // the two self-conjugate bins (see rfft_taper_phase)
      const float x0 = zk.x + zk.y, xm = R.t[q][1] * (zk.x - zk.y);
      st1(Y, 0, c32{0.5f * (x0 + xm) * invM, -0.5f * (x0 - xm) * invM});
      st1(Y, M / 2, cscale(cconj(zm), R.t[q][0]));
    }
  }
}
template <int SU, class YP>
PAYNE_HD void slots_commit(int tid, int nthr, int M, SlotRegs<SU>& R, YP Y, const TaperArgs& ta, bool scrub) {
  slots_tapers<SU>(tid, nthr, M, R, ta);
  slots_store<SU>(tid, nthr, M, R, Y, scrub);
}

// ---------------------------------------------------------------------------
// Grid positions.
// ---------------------------------------------------------------------------
// k in [lo, hi-2] with x[k] <= v < x[k+1] (clamped at the ends), from a guess.
PAYNE_HD int locate(const double* x, int lo, int hi, double v, int guess) {
  int k = guess < lo ? lo : (guess > hi - 2 ? hi - 2 : guess);
  while (k > lo && v < x[k]) --k;
  while (k < hi - 2 && v >= x[k + 1]) ++k;
  return k;
}
// Non-geometric ANN grids: pixel k of [lo,hi) with lnlam[k] <= v < lnlam[k+1] (clamped) and
// the np.interp weight of pixel k+1 for ln-wavelength v (clamped to [0,1]).
PAYNE_HD void search_locate(const PostTables& T, int lo, int hi, double v, int& k, float& w) {
  const int guess = lo + (int)((v - T.lnlam[lo]) * T.geo_inv_dln);
  k = locate(T.lnlam, lo, hi, v, guess);
  const double u = v - T.lnlam[k], dv = T.lnlam[k + 1] - T.lnlam[k];
  const float f = (float)(u / dv);
  const float ww = f * (1.0f + 0.5f * (float)dv * (f - 1.0f));
  w = (u <= 0.0) ? 0.f : ((u >= dv) ? 1.f : ww);
}
// position t on a uniform ln grid -> (k, weight of k+1), k clamped to [lo, hi-2]:  k = floor(t), f = t - k,
// weight = f (1 + hs (f - 1))  [= expm1(f v)/expm1(v), v = 2 hs].
// Done without fp64 conversions (v_cvt_*_f64 run at a fraction of the fma rate and a floor/convert
// formulation needs four per pixel): the caller folds kPosMagic = 1.5 * 2^20 into the constant term of its
// position fma, so that tm = t + kPosMagic (0 <= t < 2^19) has ulp 2^-32: the low dword of tm IS
// the fraction in units of 2^-32, bits 0..18 of the high dword the integer part, and the
// exponent field is the constant 0x413.  Position resolution 2.3e-10 pixel (single rounding).
// Anything else in the exponent field (NaN, negative, huge) gives a NaN weight.
PAYNE_HD void magic_locate(double tm, int lo, int hi, float hs, int& k, float& w) {
  union { double d; unsigned long long u; } cv;
  cv.d = tm;
  const unsigned lo_dw = (unsigned)cv.u, hi_dw = (unsigned)(cv.u >> 32);
  int kk = (int)(hi_dw & 0x7FFFFu);
  float f = (float)lo_dw * 2.3283064365386963e-10f;      // 2^-32
  const bool below = kk < lo, above = kk > hi - 2;
  kk = below ? lo : (above ? hi - 2 : kk);
  f = below ? 0.f : (above ? 1.f : f);
  f = ((hi_dw >> 20) == 0x413u) ? f : nanf_();
  k = kk;
  w = f * (1.0f + hs * (f - 1.0f));
}

// ---------------------------------------------------------------------------
// Phases.  `spec`/`work` are the two LDS spectra buffers (n1 floats each).  Loops that
// read one LDS buffer and write the other are written "load U items, then store U
// items" so that the U gathers are in flight together.
// ---------------------------------------------------------------------------
constexpr int kU = 4;

// The R-stage window from the two mask counts (see phase_mask_count / make_window below).
PAYNE_HD int pow2ceil_fast(int n) {                    // n >= 2
#ifdef __HIP_DEVICE_COMPILE__
  return 1 << (32 - __clz(n - 1));
#else
  return pow2ceil(n);
#endif
}
// Every thread evaluates this between two phases, so it is kept short: one fp64 division (the
// np.linspace step), the rest multiplications by reciprocals that are exact or constant.
PAYNE_HD Window window_from_counts(const PostTables& T, double dop, double g_a, int below, int notabove) {
  Window W;
  W.i0 = below; W.i1 = notabove;
  const int n = W.i1 - W.i0;
  W.bad = (n < 8) ? 1 : 0;
  if (W.bad) { W.i0 = 0; W.i1 = 8; }                                 // any valid range: results are NaN'd
  W.n2 = pow2ceil_fast(W.bad ? 8 : n);
  const double l0 = T.geo ? (T.ln0 + (double)W.i0 * T.dln) : T.lnlam[W.i0];
  const double l1 = T.geo ? (T.ln0 + (double)(W.i1 - 1) * T.dln) : T.lnlam[W.i1 - 1];
  W.lnmin = l0 + dop;
  W.lnmax = l1 + dop;
  const double span = W.lnmax - W.lnmin, nm1 = (double)(W.n2 - 1);
  W.step = span / nm1;                                               // np.linspace
  // 1/step: |step * (nm1/span) - 1| < 2^-52, positions move by < 1e-12 pixel
  const double inv_step = nm1 * (1.0 / span);
  W.rsA = W.step * T.geo_inv_dln;                                    // geo: geo_inv_dln == 1/dln
  W.rsB = (l0 - T.ln0) * T.geo_inv_dln;
  W.obA = inv_step;
  W.obB = -W.lnmin * inv_step;
  W.hs_ann = (float)(0.5 * T.dln);
  W.hs_step = (float)(0.5 * W.step);
  const double g_val = inv_step * ((1.0 / kCkms) / (double)W.n2);    // rfftfreq: 1/(n2 dv), dv = ckms*step; 1/n2 exact
  W.g_c2 = (float)(g_a * (g_val * g_val) * 1.4426950408889634);
  return W;
}
PAYNE_HD int probe_start(const PostTables& T, float op32, double lim) {
  const float ratio = ((float)lim * T.inv_lam0) / op32;
#ifdef __HIP_DEVICE_COMPILE__
  const float pos = (__builtin_amdgcn_logf(ratio) * 0.6931471805599453f) * T.inv_dln32;   // v_log_f32 is log2
#else
  const float pos = logf(ratio) * T.inv_dln32;
#endif
  if (!(pos > -1e9f && pos < 1e9f)) return INT32_MIN;
  return (int)floorf(pos) - 31;
}
template <bool UPPER>
PAYNE_HD int probe_count(const PostTables& T, double op, double lim, int s, int lane, int nl) {
  if (s == INT32_MIN) return -1;
  int cnt = 0, nvalid = 0;
#ifdef __HIP_DEVICE_COMPILE__
  if (nl != 64) return -1;                                              // needs a whole wave
  const int idx = s + lane;
  const bool valid = idx >= 0 && idx < T.npix;
  const double c = T.lam[valid ? idx : 0] * op;
  const bool pr = valid && (UPPER ? (c < lim) : !(c > lim));
  cnt = __popcll(__ballot(pr));
  nvalid = __popcll(__ballot(valid));
#else
  (void)lane; (void)nl;
  for (int l = 0; l < 64; ++l) {
    const int idx = s + l;
    const bool valid = idx >= 0 && idx < T.npix;
    if (!valid) continue;
    const double c = T.lam[idx] * op;
    ++nvalid;
    if (UPPER ? (c < lim) : !(c > lim)) ++cnt;
  }
#endif
  if (s > 0 && cnt == 0) return -1;                                     // boundary is left of the probe
  if (s + 63 < T.npix - 1 && cnt == nvalid) return -1;                  // ... or right of it (also NaN products)
  if (nvalid == 0) return s < 0 ? 0 : T.npix;
  return (s > 0 ? s : 0) + cnt;
}
// Setup-time mask counts (one wave): fills S.win_below / S.win_notabove / S.win_ready.
PAYNE_HD void setup_window(const PostTables& T, const double* th, double wl, double wh, bool smooth,
                           CandState& S, int lane, int nl) {
  int ready = 0, below = 0, notabove = 0;
  if (smooth && T.geo) {
    const double rv = th[4];
    const double op = (rv != 0.0) ? (1.0 + (rv / kCDoppler)) : 1.0;    // as phase_setup's thread 0
    const float op32 = (float)op;
    const int s_lo = probe_start(T, op32, wl), s_hi = probe_start(T, op32, wh);
    below = probe_count<false>(T, op, wl, s_lo, lane, nl);
    notabove = probe_count<true>(T, op, wh, s_hi, lane, nl);
    ready = (below >= 0 && notabove >= 0) ? 1 : 0;
  }
  if (lane == 0) { S.win_below = below; S.win_notabove = notabove; S.win_ready = ready; S.w_ready = 0; }
}

// ---------------------------------------------------------------------------
// Everything phase_setup + the window derivation produce, by ONE thread, ahead of the post
// kernel (extra workgroups of the first dense-layer launch run it while the matrix cores
// work): the kernel then starts from a 256-byte record instead of ~7000 cycles of dependent
// fp64 arithmetic and two dependent memory round trips.
// ---------------------------------------------------------------------------
// #pixels whose Doppler-shifted wavelength passes the (monotone) limit test: the index of the
// first failing pixel.  A guess that is right costs two loads; otherwise a bisection.
template <bool UPPER>
PAYNE_HD int count_search(const PostTables& T, double op, double lim, int guess) {
  const double* __restrict__ lam = T.lam;
  auto pred = [&](int i) { const double c = lam[i] * op; return UPPER ? (c < lim) : !(c > lim); };
  const int n = T.npix;
  int g = guess < 0 ? 0 : (guess > n ? n : guess);
  for (int tries = 0; tries < 3; ++tries) {                         // the guess, then its neighbours
    const bool left_ok = (g == 0) || pred(g - 1), right_ok = (g == n) || !pred(g);
    if (left_ok && right_ok) return g;
    if (!left_ok) { if (g == 0) break; --g; } else { if (g == n) break; ++g; }
  }
  int lo = 0, hi = n;                                               // pred true on [0, lo), false on [hi, n)
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (pred(mid)) lo = mid + 1; else hi = mid; }
  return lo;
}
// ln(1 + x) for the Doppler factor's x = rv / c by its series, no branch: sixteen terms -- |x| < 0.01 (|rv| < 3000 km/s) to an ulp
// or two, 1e-17 relative at |x| = 0.1; NaN stays NaN.  (The library logarithm is ~100 dependent fp64 instructions on the critical path
// of the hidden-layer launch's last workgroups.)
PAYNE_HD double log1p_series(double x) {
  double p = -1.0 / 16.0;
  p = p * x + 1.0 / 15.0; p = p * x - 1.0 / 14.0; p = p * x + 1.0 / 13.0; p = p * x - 1.0 / 12.0; p = p * x + 1.0 / 11.0;
  p = p * x - 1.0 / 10.0; p = p * x + 1.0 / 9.0; p = p * x - 1.0 / 8.0; p = p * x + 1.0 / 7.0; p = p * x - 1.0 / 6.0;
  p = p * x + 1.0 / 5.0; p = p * x - 1.0 / 4.0; p = p * x + 1.0 / 3.0; p = p * x - 1.0 / 2.0;
  return x + (x * x) * p;
}
PAYNE_HD void prep_candidate(const PostTables& T, const double* th, double instr_factor, CandState& S) {
  // the whole row is requested before anything is computed (clamped column for the coefficients past npoly: a guarded
  // load is a branch and a wait of its own, fifteen round trips in a row as first written -- and the two workgroups that
  // run this are the last of the hidden-layer launch to finish)
  // (one thread per candidate: every load instruction of a wave touches 64 rows = 64 cache lines, ~100 cycles of the memory pipe
  //  each -- a fit without a blaze polynomial asks for three values, not fifteen)
  const double rv = th[4], vrot = th[5], r_in = th[7];
  double pc[12];
  if (T.npoly > 0) {                                                  // (uniform)
#pragma unroll
    for (int i = 0; i < 12; ++i) pc[i] = th[i < T.npoly ? 8 + i : 7];   // (column 7 exists in every row)
  } else {
#pragma unroll
    for (int i = 0; i < 12; ++i) pc[i] = 0.0;
  }
  S.one_plus = (rv != 0.0) ? (1.0 + (rv / kCDoppler)) : 1.0;       // ystpred.py:228-232
  S.dop = log1p_series(S.one_plus - 1.0);                  // (= log(one_plus): the difference is exact)
  S.do_rot = (vrot != 0.0);                                         // ystpred.py:214 (NaN passes)
  S.vs_a = 2.0 * kPi * sqrt(vrot * vrot - 0.0);                     // smoothing.py:297,614
  const double Rs = r_in * instr_factor;                            // genmod.py:82-85
  S.do_smooth = (Rs > 0.0);                                         // ystpred.py:238-240 (false for NaN)
  S.g_a = 0.0; S.wl = 0.0; S.wh = 0.0;
  S.win_ready = 0; S.win_below = 0; S.win_notabove = 0; S.w_ready = 0;
#pragma unroll
  for (int i = 0; i < 12; ++i) S.poly[i] = (i < T.npoly) ? pc[i] : 0.0;
  Window W{};
  if (Rs > 0.0 && T.nobs > 0) {
    const double sig_out = kCkms / Rs, inres = kCkms / T.r_ann;     // smoothing.py:107,113
    const double sig = sqrt(sig_out * sig_out - inres * inres);     // :271 (NaN if negative)
    S.g_a = -2.0 * (kPi * kPi) * (sig * sig);
    const double pad = 20.0 / Rs;                                   // mask_wave, smoothing.py:631-647
    S.wl = T.obs_min * (1.0 + pad * -1.0);
    S.wh = T.obs_max * (1.0 + pad * 1.0);
    // Geometric grids (ln lam_i = ln0 + i dln to 1e-12, verified at set-up): the two counts by ARITHMETIC.  lam_i (1 + rv/c) <= wl
    // <=> i <= (ln wl - dop - ln0) / dln =: t, and ln wl = ln(obs_min) + ln(1 - pad) with the first term the context's and the
    // second a short series: no logarithm, no table value, i.e. no second memory round trip in the hidden-layer launch's last
    // workgroups (0.8 us of that launch: the twin PAYNE_EXP_PREP=1).  t is good to ~1e-9 pixel (rounding) + 1e-12 / dln (the grid's
    // own deviation, < 3e-7 pixel): where its fraction is more than 1e-5 from an integer the count floor(t) + 1 is THE count the
    // table gives; the rest (one candidate in 50 000) and every other grid take the table as before.
    bool have = false;
    if (T.geo) {
      const double tl = ((T.ln_obs_min + log1p_series(pad * -1.0)) - S.dop - T.ln0) * T.geo_inv_dln;
      const double th_ = ((T.ln_obs_max + log1p_series(pad * 1.0)) - S.dop - T.ln0) * T.geo_inv_dln;
      const double fl = floor(tl), fh = floor(th_);
      const bool safe = (tl - fl > 1e-5) && (tl - fl < 1.0 - 1e-5) && (th_ - fh > 1e-5) && (th_ - fh < 1.0 - 1e-5);   // (false for NaN)
      if (safe) {
        const double nn = (double)T.npix;
        const double cl = fl + 1.0, ch = fh + 1.0;
        S.win_below = (int)(cl < 0.0 ? 0.0 : (cl > nn ? nn : cl));
        S.win_notabove = (int)(ch < 0.0 ? 0.0 : (ch > nn ? nn : ch));
        have = true;
      }
    }
    if (!have) {
    const float op32 = (float)S.one_plus;
    const int g_lo = probe_start(T, op32, S.wl), g_hi = probe_start(T, op32, S.wh);
    // both guesses checked with ONE round trip (four independent loads); a guess that does not bracket its limit
    // (or sits at an end of the grid) goes through count_search
    const int n = T.npix;
    const int ga = g_lo == INT32_MIN ? 0 : g_lo + 32, gb = g_hi == INT32_MIN ? 0 : g_hi + 32;
    const bool ina = ga >= 1 && ga < n, inb = gb >= 1 && gb < n;
    const double a0 = T.lam[ina ? ga - 1 : 0] * S.one_plus, a1 = T.lam[ina ? ga : 0] * S.one_plus;
    const double b0 = T.lam[inb ? gb - 1 : 0] * S.one_plus, b1 = T.lam[inb ? gb : 0] * S.one_plus;
#if defined(PAYNE_EXP_PREP) && (PAYNE_EXP_PREP & 1)      /* timing twin: the guesses taken on trust (no second memory round trip) */
    const bool oka = ina && a0 == a0 + 0.0 * a1, okb = inb && b0 == b0 + 0.0 * b1;
#else
    const bool oka = ina && !(a0 > S.wl) && (a1 > S.wl);          // pred true at ga-1, false at ga
    const bool okb = inb && (b0 < S.wh) && !(b1 < S.wh);
#endif
    S.win_below = oka ? ga : count_search<false>(T, S.one_plus, S.wl, ga);
    S.win_notabove = okb ? gb : count_search<true>(T, S.one_plus, S.wh, gb);
    }
    S.win_ready = 1;
#if defined(PAYNE_EXP_PREP) && (PAYNE_EXP_PREP & 2)      /* timing twin: the window left to the post kernel */
    S.w_ready = 0;
#else
    W = window_from_counts(T, S.dop, S.g_a, S.win_below, S.win_notabove);
    S.w_ready = 1;
#endif
  }
  S.W = W;
}
// Phase 0 of the kernel when the record exists: a dword copy into the workgroup's state, in two halves so that the load is
// in flight with the row's (request, ..., store): one memory round trip at the start of the workgroup instead of two.
constexpr int kPrepDwords = (int)(sizeof(CandState) / 4);
static_assert(sizeof(CandState) % 4 == 0, "dword copy");
constexpr int kPrepPerThread = (kPrepDwords + 63) / 64;
struct PrepRegs { unsigned v[kPrepPerThread]; };
PAYNE_HD void phase_take_prep_issue(int tid, const CandState* __restrict__ prep, PrepRegs& R) {
  const unsigned* __restrict__ src = reinterpret_cast<const unsigned*>(prep);
  if (tid < 64) {
#pragma unroll
    for (int q = 0; q < kPrepPerThread; ++q) { const int i = tid + 64 * q; R.v[q] = src[i < kPrepDwords ? i : kPrepDwords - 1]; }
  }
}
PAYNE_HD void phase_take_prep_commit(int tid, const PrepRegs& R, CandState& S) {
  unsigned* dst = reinterpret_cast<unsigned*>(&S);
  if (tid < 64) {
#pragma unroll
    for (int q = 0; q < kPrepPerThread; ++q) { const int i = tid + 64 * q; if (i < kPrepDwords) dst[i] = R.v[q]; }
  }
}

// P0: per-candidate scalars from theta.  The independent fp64 chains (log / sqrt / the
// instrument width) go to the first thread of different waves so that they overlap.
// theta columns: 0 Teff 1 logg 2 FeH 3 aFe 4 Vrad 5 Vrot 6 Vmic 7 Inst_R 8.. pc_*
PAYNE_HD void phase_setup(int tid, int nthr, const PostTables& T, const double* th, double instr_factor,
                          CandState& S) {
  const int lanes = nthr >= 256 ? 64 : 0;            // 0: everything on thread 0 (small / emulated groups)
  if (tid == 0) {
    const double rv = th[4];
    S.one_plus = (rv != 0.0) ? (1.0 + (rv / kCDoppler)) : 1.0;   // ystpred.py:228-232
    S.dop = log1p_series(S.one_plus - 1.0);                  // (= log(one_plus): the difference is exact)
  }
  if (tid == lanes) {
    const double vrot = th[5];
    S.do_rot = (vrot != 0.0);                       // ystpred.py:214 (NaN passes)
    S.vs_a = 2.0 * kPi * sqrt(vrot * vrot - 0.0);   // smoothing.py:297,614
  }
  // the instrument chain runs on a whole wave: its first lane publishes the scalars, and all 64
  // lanes probe the mask limits (geometric grids) while the spectrum row is still on its way from
  // memory; otherwise phase_mask_count runs later
  if (lanes ? ((tid >> 6) == 2) : (tid == 0)) {
    const int lane = lanes ? (tid & 63) : 0;
    const double Rs = th[7] * instr_factor;         // genmod.py:82-85
    double g_a = 0.0, wl = 0.0, wh = 0.0;
    if (Rs > 0.0) {
      const double sig_out = kCkms / Rs, inres = kCkms / T.r_ann;   // smoothing.py:107,113
      const double sig = sqrt(sig_out * sig_out - inres * inres);   // :271 (NaN if negative)
      g_a = -2.0 * (kPi * kPi) * (sig * sig);
      const double pad = 20.0 / Rs;                                 // mask_wave, smoothing.py:631-647
      wl = T.obs_min * (1.0 + pad * -1.0);
      wh = T.obs_max * (1.0 + pad * 1.0);
    }
    if (lane == 0) {
      S.do_smooth = (Rs > 0.0);                     // ystpred.py:238-240 (false for NaN)
      S.g_a = g_a; S.wl = wl; S.wh = wh;
    }
    setup_window(T, th, wl, wh, Rs > 0.0, S, lane, lanes ? 64 : 1);
  }
  if (tid == 3 * lanes)
    for (int i = 0; i < T.npoly && i < 12; ++i) S.poly[i] = th[8 + i];
}

// P1: load the raw ANN spectrum (already shifted by -1) into LDS.  SCRUB applies
// nan_to_num(nan=1.0) (0 in shifted flux, smoothing.py:138) on the way: used when the row
// goes straight into the vsini FFT (identity resampling maps).
// Split in two so that the global loads of the first kU*nthr float4 (the whole row at 4096
// pixels) are in flight WHILE phase_setup's fp64 chains run: issue -> setup -> commit.
template <int U> struct RowRegsT { float v[U][4]; };
typedef RowRegsT<kU> RowRegs;
PAYNE_HD bool row_vectorised(int npix, const float* raw) { return ((npix & 3) == 0) && ((((uintptr_t)raw) & 15) == 0); }
template <int U>
PAYNE_HD void phase_load_issue(int tid, int nthr, int npix, const float* __restrict__ raw, RowRegsT<U>& R) {
  if (!row_vectorised(npix, raw)) return;
  const int n4 = npix >> 2;
#pragma unroll
  for (int q = 0; q < U; ++q) {                      // clamped index: unconditional loads
    const int i0 = tid + q * nthr, i = i0 < n4 ? i0 : n4 - 1;
#ifdef __HIP_DEVICE_COMPILE__
    // read once, produced by other XCDs: streaming loads (no L2 allocation)
    R.v[q][0] = __builtin_nontemporal_load(&raw[4 * i]); R.v[q][1] = __builtin_nontemporal_load(&raw[4 * i + 1]);
    R.v[q][2] = __builtin_nontemporal_load(&raw[4 * i + 2]); R.v[q][3] = __builtin_nontemporal_load(&raw[4 * i + 3]);
#else
    R.v[q][0] = raw[4 * i]; R.v[q][1] = raw[4 * i + 1]; R.v[q][2] = raw[4 * i + 2]; R.v[q][3] = raw[4 * i + 3];
#endif
  }
}
// four consecutive values of the spectrum buffer (16-byte aligned: both buffers start on 16-byte boundaries) as ONE 16-byte store
// (four 4-byte stores put lanes i and i + 8 on the same bank: 4-way conflicts on the whole row)
PAYNE_HD void row_store4(float* __restrict__ spec, int i, float a, float b, float c, float d) {
#ifdef __HIP_DEVICE_COMPILE__
  typedef float row4 __attribute__((ext_vector_type(4)));
  row4 v; v.x = a; v.y = b; v.z = c; v.w = d;
  *reinterpret_cast<row4*>(spec + 4 * i) = v;
#else
  spec[4 * i] = a; spec[4 * i + 1] = b; spec[4 * i + 2] = c; spec[4 * i + 3] = d;
#endif
}
template <int U>
PAYNE_HD void phase_load_commit(int tid, int nthr, int npix, const float* __restrict__ raw, const RowRegsT<U>& R,
                                float* __restrict__ spec, bool scrub) {
  if (row_vectorised(npix, raw)) {
    const int n4 = npix >> 2;
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int i = tid + q * nthr;
      if (i < n4) row_store4(spec, i, scrub ? nan_to_zero(R.v[q][0]) : R.v[q][0], scrub ? nan_to_zero(R.v[q][1]) : R.v[q][1],
                             scrub ? nan_to_zero(R.v[q][2]) : R.v[q][2], scrub ? nan_to_zero(R.v[q][3]) : R.v[q][3]);
    }
    for (int base = tid + U * nthr; base < n4; base += U * nthr) {      // rows longer than U*nthr float4
      float v[U][4];
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const int i0 = base + q * nthr, i = i0 < n4 ? i0 : n4 - 1;
        v[q][0] = raw[4 * i]; v[q][1] = raw[4 * i + 1]; v[q][2] = raw[4 * i + 2]; v[q][3] = raw[4 * i + 3];
      }
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const int i = base + q * nthr;
        if (i < n4) row_store4(spec, i, scrub ? nan_to_zero(v[q][0]) : v[q][0], scrub ? nan_to_zero(v[q][1]) : v[q][1],
                               scrub ? nan_to_zero(v[q][2]) : v[q][2], scrub ? nan_to_zero(v[q][3]) : v[q][3]);
      }
    }
  } else {
    for (int i = tid; i < npix; i += nthr) spec[i] = scrub ? nan_to_zero(raw[i]) : raw[i];
  }
}

// vsini a: resample onto the pow-2 log grid (static map) into `work`; identity maps
// (geometric grid, npix a power of two) degenerate to a NaN-scrubbing copy.
template <bool IDENT>
PAYNE_HD void rot_resample_loop(int tid, int nthr, const PostTables& T, const float* __restrict__ spec,
                                float* __restrict__ work) {
  for (int base = tid; base < T.n1; base += kU * nthr) {
    float a[kU], b[kU], f[kU];
#pragma unroll
    for (int q = 0; q < kU; ++q) {
      const int j0 = base + q * nthr, j = j0 < T.n1 ? j0 : T.n1 - 1;
      if (IDENT) { a[q] = spec[j]; b[q] = a[q]; f[q] = 0.f; }
      else { const int k = T.rs1_idx[j]; f[q] = T.rs1_frac[j]; a[q] = spec[k]; b[q] = spec[k + 1]; }
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
// (workgroup-uniform conditions are resolved OUTSIDE the unrolled loops everywhere below: a
// branch inside the body, even a uniform one, stops the compiler from batching the loads of
// the unrolled iterations, and each iteration then pays its own memory latency)
PAYNE_HD void phase_rot_resample(int tid, int nthr, const PostTables& T, const float* __restrict__ spec,
                                 float* __restrict__ work) {
  if (T.rot_identity) rot_resample_loop<true>(tid, nthr, T, spec, work);
  else rot_resample_loop<false>(tid, nthr, T, spec, work);
}
// vsini c: back onto the ANN grid (left/right = NaN), into `spec` (skipped for identity maps:
// the convolved buffer then IS the spectrum on the ANN grid).
// `edges`: apply spec[0] = spec[1], spec[-1] = spec[-2] (ystpred.py:223-224; phase_rot_edges) on the way: the threads that own the
// first and the last pixel evaluate their neighbours' values instead of their own -- no phase of its own for two stores.
// On a geometric grid with the edge rule riding along the positions are arithmetic (pixel i sits at i (n1 - 1)/(npix - 1) of the
// stage's grid, which spans the same ends; the two end pixels -- the only ones np.interp can find outside, by a rounding of
// exp(log(.)) -- take their neighbours' values anyway): no map loads, a phase without a global round trip.
template <int U = kU>
PAYNE_HD void phase_rot_back(int tid, int nthr, const PostTables& T, const float* __restrict__ work,
                             float* __restrict__ spec, bool edges = false) {
  if (T.geo && edges) {
    const double r = T.bk_r;
    const float c1 = T.bk_c1, c2 = T.bk_c2;
    if (U * nthr == T.n1) {
      // one block, U points a thread, nothing conditional: points past the last pixel are copies of it in a part of the buffer
      // (n1 floats) that nothing reads
      const int hi = T.npix - 2;
      float a[U], b[U], F[U];
      unsigned k[U];
#pragma unroll
      for (int q = 0; q < U; ++q) {
        int i = tid + q * nthr;
        i = i < hi ? i : hi;
        i = i > 1 ? i : 1;
        union { double d; unsigned long long u; } cv;
        cv.d = fma((double)i, r, kPosMagic);
        k[q] = (unsigned)(cv.u >> 32) - kPosMagicHi;
        F[q] = (float)(unsigned)cv.u;
      }
      ld_pairs<U>(work, k, a, b);
#pragma unroll
      for (int q = 0; q < U; ++q) spec[tid + q * nthr] = fmaf(b[q] - a[q], F[q] * fmaf(F[q], c2, c1), a[q]);
      return;
    }
    for (int base = tid; base < T.npix; base += U * nthr) {
      float a[U], b[U], F[U];
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const int i0 = base + q * nthr;
        int i = i0 < T.npix ? i0 : T.npix - 1;
        i = (i == 0) ? 1 : ((i == T.npix - 1) ? T.npix - 2 : i);
        union { double d; unsigned long long u; } cv;
        cv.d = fma((double)i, r, kPosMagic);
        const int k = (int)((unsigned)(cv.u >> 32) - kPosMagicHi);
        F[q] = (float)(unsigned)cv.u;
        a[q] = work[k]; b[q] = work[k + 1];
      }
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const int i = base + q * nthr;
        if (i < T.npix) spec[i] = fmaf(b[q] - a[q], F[q] * fmaf(F[q], c2, c1), a[q]);
      }
    }
    return;
  }
  for (int base = tid; base < T.npix; base += U * nthr) {
    float a[U], b[U], f[U];
    int jj[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int i0 = base + q * nthr;
      int i = i0 < T.npix ? i0 : T.npix - 1;
      if (edges) i = (i == 0) ? 1 : ((i == T.npix - 1) ? T.npix - 2 : i);
      jj[q] = T.bk1_idx[i]; f[q] = T.bk1_frac[i];
      const int j = jj[q] < 0 ? 0 : jj[q];
      a[q] = work[j]; b[q] = work[j + 1];
    }
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int i = base + q * nthr;
      if (i < T.npix) spec[i] = (jj[q] < 0) ? nanf_() : a[q] + (b[q] - a[q]) * f[q];
    }
  }
}
// vsini d: spec[0]=spec[1]; spec[-1]=spec[-2]  (ystpred.py:223-224)
PAYNE_HD void phase_rot_edges(int tid, int npix, float* spec) {
  if (tid == 0) spec[0] = spec[1];
  if (tid == 1) spec[npix - 1] = spec[npix - 2];
}

// R a: data-dependent mask (smoothing.py:631-647), exact fp64 products as numpy.  The ANN
// grid is increasing, so the mask is the run [#(lam' <= wl), #(lam' < wh)): two counts.
// On the GPU the counts are wave ballots + popcounts (scalar unit, no atomics); each
// wave leaves one packed partial in cnt[] (host emulation: one per thread).
PAYNE_HD void phase_mask_count(int tid, int nthr, const PostTables& T, const CandState& S, int* cnt) {
  const double wl = S.wl, wh = S.wh, op = S.one_plus;
  int cb = 0, ca = 0;
  constexpr int MU = 16;                                 // global loads: keep a whole thread-share in flight
  for (int base = tid; base < T.npix; base += MU * nthr) {
    double wc[MU];
#pragma unroll
    for (int q = 0; q < MU; ++q) {
      const int i0 = base + q * nthr;
      wc[q] = T.lam[i0 < T.npix ? i0 : T.npix - 1];
    }
#pragma unroll
    for (int q = 0; q < MU; ++q) {
      const bool valid = (base + q * nthr) < T.npix;
      const double c = wc[q] * op;
      const bool below = valid && !(c > wl), notabove = valid && (c < wh);
#ifdef __HIP_DEVICE_COMPILE__
      cb += __popcll(__ballot(below));
      ca += __popcll(__ballot(notabove));
#else
      cb += below ? 1 : 0;
      ca += notabove ? 1 : 0;
#endif
    }
  }
  if ((tid & ((1 << kSlotShift) - 1)) == 0) cnt[tid >> kSlotShift] = cb | (ca << 16);
}

// R b: every thread derives the window from the partial counts (no serial section).
// ln(lam_k (1+rv/c)) is taken as lnlam[k] + dop (differs from log of the rounded product by
// < 2e-16, i.e. < 1e-10 pixel: grid and interpolation depend on it continuously).
PAYNE_HD Window make_window(const PostTables& T, const CandState& S, const int* cnt, int nslots) {
  int below = 0, notabove = 0;
  for (int s = 0; s < nslots; ++s) { const int v = cnt[s]; below += v & 0xffff; notabove += v >> 16; }
  return window_from_counts(T, S.dop, S.g_a, below, notabove);
}

// R c: resample the masked, Doppler-shifted spectrum onto its pow-2 log grid.
template <bool GEO, int RU>
PAYNE_HD void R_resample_loop(int tid, int nthr, const PostTables& T, const CandState& S, const Window& W,
                              const float* __restrict__ spec, float* __restrict__ work) {
  // RU: a whole thread-share of gathers in flight
  const double rsBm = W.rsB + kPosMagic, rsD = (double)nthr * W.rsA;
  // GEO: the POSITION is clamped to [first pixel, one position ulp below the last] -- its integer part is then in [i0, i1 - 2] and its
  // fraction in [0, 1): what magic_locate's integer clamps, fraction selects and exponent test produce, for two instructions instead
  // of three round trips through a condition register (20 ticks each for a wave: tools/exp/nop_rate.hip).  Positions of a window that
  // is not `bad` are finite.
  const double tlo = kPosMagic + (double)W.i0, thi = kPosMagic + (double)(W.i1 - 1) - 2.3283064365386963e-10;
  for (int base = tid; base < W.n2; base += RU * nthr) {
    float a[RU], b[RU], w[RU];
    const double tm0 = fma((double)base, W.rsA, rsBm);   // point base + q nthr sits at tm0 + q rsD
#pragma unroll
    for (int q = 0; q < RU; ++q) {
      const int j0 = base + q * nthr, j = j0 < W.n2 ? j0 : W.n2 - 1;
      int k; float ww;
      if (GEO) {                                         // (clamped: j0 >= n2 is harmless)
        union { double d; unsigned long long u; } cv;
        cv.d = fmin(fmax(fma((double)q, rsD, tm0), tlo), thi);
        k = (int)((unsigned)(cv.u >> 32) & 0x7FFFFu);
        const float f = (float)(unsigned)cv.u * 2.3283064365386963e-10f;      // 2^-32
        ww = f * (1.0f + W.hs_ann * (f - 1.0f));
      }
      else {
        const double lw = (j == W.n2 - 1) ? W.lnmax : ((double)j * W.step + W.lnmin);
        search_locate(T, W.i0, W.i1, lw - S.dop, k, ww);
      }
      a[q] = spec[k]; b[q] = spec[k + 1]; w[q] = ww;
    }
#pragma unroll
    for (int q = 0; q < RU; ++q) {
      const int j = base + q * nthr;
      if (j < W.n2) {
        const float aa = nan_to_zero(a[q]), bb = nan_to_zero(b[q]);   // nan_to_num, smoothing.py:138
        work[j] = aa + (bb - aa) * w[q];
      }
    }
  }
}
// The usual case of the loop above with half its instructions: geometric grid and a window of exactly RU x nthr points (every
// thread owns RU of them: no bounds).  Only point 0 can round to a position below the first masked pixel and only point n2 - 1
// above the last: the clamps sit on the first and the last row alone.  nan_to_num is not applied value by value: a NaN input shows
// in the interpolated value, the function reports it and the caller runs the general loop instead (a NaN row; the NaN edges a
// rotation stage with resampling maps leaves).  Weight f (1 + hs (f - 1)) from the integer fraction F = f 2^32 as F (c1 + c2 F).
template <int RU>
PAYNE_HD bool R_resample_fast(int tid, int nthr, const Window& W, const float* __restrict__ spec, float* __restrict__ work) {
  const double rsBm = W.rsB + kPosMagic, rsD = (double)nthr * W.rsA;
  const double tlo = kPosMagic + (double)W.i0, thi = kPosMagic + (double)(W.i1 - 1) - 2.3283064365386963e-10;
  const float c1 = 2.3283064365386963e-10f * (1.0f - W.hs_ann), c2 = 5.421010862427522e-20f * W.hs_ann;
  const double tm0 = fma((double)tid, W.rsA, rsBm);
  float a[RU], b[RU], F[RU];
  unsigned k[RU];
#pragma unroll
  for (int q = 0; q < RU; ++q) {
    union { double d; unsigned long long u; } cv;
    cv.d = fma((double)q, rsD, tm0);
    if (q == 0) cv.d = fmax(cv.d, tlo);
    if (q == RU - 1) cv.d = fmin(cv.d, thi);
    k[q] = (unsigned)(cv.u >> 32) - kPosMagicHi;
    F[q] = (float)(unsigned)cv.u;
  }
  ld_pairs<RU>(spec, k, a, b);
  bool nan = false;
#pragma unroll
  for (int q = 0; q < RU; ++q) {
    const float r = fmaf(b[q] - a[q], F[q] * fmaf(F[q], c2, c1), a[q]);
    nan = nan || (r != r);
    work[tid + q * nthr] = r;
  }
  return nan;
}
template <int RU = 16>
PAYNE_HD void phase_R_resample(int tid, int nthr, const PostTables& T, const CandState& S, const Window& W,
                               const float* __restrict__ spec, float* __restrict__ work) {
  if (T.geo) {
    if (W.n2 == RU * nthr && !wave_any(R_resample_fast<RU>(tid, nthr, W, spec, work))) return;
    R_resample_loop<true, RU>(tid, nthr, T, S, W, spec, work);
  }
  else R_resample_loop<false, RU>(tid, nthr, T, S, W, spec, work);
}

// Final: interpolate onto the observed grid, blaze, chi^2 partial per thread.
// `conv` = smoothed spectrum on the candidate's log grid (do_smooth) or the
// (rotated) spectrum on the ANN grid (plain np.interp branch, ystpred.py:271-272).
// MODE: 0 = smoothed spectrum on the candidate's uniform log grid; 1 = plain interpolation on a
// geometric ANN grid; 2 = plain interpolation with a search (non-geometric grid).
template <int MODE, bool CHEB, bool HASF, bool OUT, int OU>
PAYNE_HD float obs_loop(int tid, int nthr, const PostTables& T, const CandState& S, const Window& W,
                        const float* __restrict__ conv, float* __restrict__ out, int out_stage) {
  float acc = 0.f;                                       // <= ~16 terms per thread: fp32 is ample; the cross-thread reduction is fp64
  const double piA = T.geo_inv_dln, piBm = -(S.dop + T.ln0) * T.geo_inv_dln + kPosMagic;   // MODE 1: t = (lnobs - dop - ln0)/dln
  const double obBm = W.obB + kPosMagic;
  // MODE 0: positions (+ kPosMagic) of the window's ends with and without the edge tolerance
  const double obTA = fma(W.lnmin - kEdgeTol, W.obA, obBm), obTD = fma(W.lnmax + kEdgeTol, W.obA, obBm);
  const double obTT = fmax(fma(W.lnmax, W.obA, obBm), kPosMagic);
  const float hs_ann = (float)(0.5 * T.dln);
  const int nc = T.npoly;
  for (int base = tid; base < T.nobs; base += OU * nthr) {
    float a[OU], b[OU], w[OU], of1[OU], iv[OU];
    double xc[OU];
    bool nanv[OU];
#pragma unroll
    for (int q = 0; q < OU; ++q) {                       // clamped index: every load unconditional
      const int i0 = base + q * nthr, i = i0 < T.nobs ? i0 : T.nobs - 1;
      const ObsRec rec = T.obs_rec[i];                   // one 16-byte load
      const double lo = rec.lnw;
      if (HASF) { of1[q] = rec.f1; iv[q] = rec.ivar; }
      if (CHEB) xc[q] = T.xcheb[i];
      int k = 0; float ww = 0.f;
      // np.interp(left=nan, right=nan).  A pixel that coincides with an end of the model grid (output on
      // the model grid itself) is inside or outside by one rounding of exp(log(.)) in the reference; here it
      // is always inside (kEdgeTol in ln lambda, ~1e-7 pixel), the host applies numpy's verdict for that case.
      if (MODE == 0) {
        // the window test and the clamp on the POSITION (one fma of the pixel's ln lambda, monotone in it): inside iff tA <= tm <= tD
        // (the positions of lnmin - tol and lnmax + tol), clamped to [position 0, position of lnmax] -- the same pixels as the test
        // on ln lambda itself up to one rounding of a bound that carries a 1e-12 tolerance for exactly that; a clamped position
        // needs neither magic_locate's lower clamp nor its exponent test
        const double tm = fma(lo, W.obA, obBm);
        nanv[q] = !(tm >= obTA && tm <= obTD);             // (true for NaN)
        union { double d; unsigned long long u; } cv;
        cv.d = fmin(fmax(tm, kPosMagic), obTT);
        const unsigned lo_dw = (unsigned)cv.u;
        const int kk = (int)((unsigned)(cv.u >> 32) & 0x7FFFFu);
        const bool above = kk > W.n2 - 2;
        k = above ? W.n2 - 2 : kk;
        const float f = above ? 1.f : (float)lo_dw * 2.3283064365386963e-10f;      // 2^-32
        ww = f * (1.0f + W.hs_step * (f - 1.0f));
      } else {
        const double v0 = lo - S.dop;
        nanv[q] = (v0 < T.ln0 - kEdgeTol) || (v0 > T.ln_last + kEdgeTol);
        const double v = fmin(fmax(v0, T.ln0), T.ln_last);
        if (MODE == 1) magic_locate(nanv[q] ? kPosMagic : fmax(fma(v + S.dop, piA, piBm), kPosMagic), 0, T.npix, hs_ann, k, ww);
        else if (!nanv[q]) search_locate(T, 0, T.npix, v, k, ww);
      }
      a[q] = conv[k]; b[q] = conv[k + 1]; w[q] = ww;
    }
#pragma unroll
    for (int q = 0; q < OU; ++q) {
      const int i = base + q * nthr;
      const bool valid = i < T.nobs;
      const float m1 = nanv[q] ? nanf_() : a[q] + (b[q] - a[q]) * w[q];
      float pm1 = 0.f, p = 1.f;
      if (CHEB) {   // numpy.polynomial.chebyshev.chebval (Clenshaw), fitutils.py:11-20
        const double x = xc[q];
        double c0, c1;
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
      if (OUT && valid) out[i] = (out_stage == 3) ? (m1 + kBase) * p : (m1 + kBase);   // genspec / getspec
      if (HASF) {
        const float d = CHEB ? (m1 * p + (pm1 - of1[q])) : (m1 - of1[q]);
        acc = valid ? fmaf(d * d, iv[q], acc) : acc;
      }
    }
  }
  return acc;
}

// The likelihood's usual case of obs_loop<0, false, true, false, OU> with a third of its instructions (profiles/r5_c2_valu_census.txt):
// smoothed spectrum on the candidate's log grid, no blaze, chi^2 only, every pixel of the thread STRICTLY inside the window
// (position t in [0, n2 - 1): the high dword of t + kPosMagic minus kPosMagicHi is then the left neighbour, one unsigned comparison
// is the whole window test -- NaN, negative and too-large positions all fail it -- and no clamp, select or edge tolerance is
// needed).  Whole blocks of the padded record table (kObsPad): no index clamp, no validity test; a padding record weighs nothing.
// The interpolation weight f (1 + hs (f - 1)) is evaluated on the integer fraction F = f 2^32 as F (c1 + c2 F).
// `bad` comes back true if some pixel failed the test: the caller then runs the general loop for this thread's pixels (on the
// GPU: for the whole wave, a uniform branch) -- same result, the fast loop's value is dropped.
struct ObsFastConsts { double obA, obBm; float c1, c2; unsigned kmax; };
PAYNE_HD ObsFastConsts obs_fast_consts(const Window& W) {
  ObsFastConsts c;
  c.obA = W.obA; c.obBm = W.obB + kPosMagic;
  c.c1 = 2.3283064365386963e-10f * (1.0f - W.hs_step); c.c2 = 5.421010862427522e-20f * W.hs_step;   // 2^-32 (1 - hs), 2^-64 hs
  c.kmax = (unsigned)(W.n2 - 2);
  return c;
}
// is the short loop worth taking for this grid and this block size?  (the padding must not be most of the work)
// (a padded pixel costs the short loop 13 instructions, a real one costs the general loop 30: worth it up to half as many again)
PAYNE_HD bool obs_fast_ok(const PostTables& T, int blk) {
  const int npad = (T.nobs + blk - 1) / blk * blk;
  return 2 * (npad - T.nobs) <= T.nobs && blk <= kObsPad;
}
// the OU records of one block of the thread, requested (one 16-byte load each; the table is padded past nobs)
template <int OU>
PAYNE_HD void obs_fast_issue(int nthr, const PostTables& T, int base, ObsRec (&rec)[OU]) {
#pragma unroll
  for (int q = 0; q < OU; ++q) rec[q] = T.obs_rec[(unsigned)(base + q * nthr)];
}
template <int OU>
PAYNE_HD void obs_fast_block(const ObsRec (&rec)[OU], const ObsFastConsts& c, const float* __restrict__ conv, float& acc, unsigned& worst) {
  float a[OU], b[OU], F[OU];
  unsigned k[OU];
#pragma unroll
  for (int q = 0; q < OU; ++q) {
    union { double d; unsigned long long u; } cv;
    cv.d = fma(rec[q].lnw, c.obA, c.obBm);
    const unsigned kk = (unsigned)(cv.u >> 32) - kPosMagicHi;
    worst = kk > worst ? kk : worst;
    k[q] = kk < c.kmax ? kk : c.kmax;                    // (a pixel that failed: any valid address)
    F[q] = (float)(unsigned)cv.u;
  }
  ld_pairs<OU>(conv, k, a, b);
#pragma unroll
  for (int q = 0; q < OU; ++q) {
    const float w = F[q] * fmaf(F[q], c.c2, c.c1);
    const float d = fmaf(b[q] - a[q], w, a[q]) - rec[q].f1;
    acc = fmaf(d * d, rec[q].ivar, acc);
  }
}
// `first`: the records of the thread's first block, already requested by the caller (ahead of the transform that produces `conv`)
template <int OU>
PAYNE_HD float obs_loop_fast(int tid, int nthr, const PostTables& T, const Window& W, const float* __restrict__ conv, bool& bad,
                             const ObsRec (*first)[OU] = nullptr) {
  float acc = 0.f;
  const ObsFastConsts c = obs_fast_consts(W);
  unsigned worst = 0u;
  int base = tid;
  if (first && base < T.nobs) { obs_fast_block<OU>(*first, c, conv, acc, worst); base += OU * nthr; }
  for (; base < T.nobs; base += OU * nthr) {
    ObsRec rec[OU];
    obs_fast_issue<OU>(nthr, T, base, rec);
    obs_fast_block<OU>(rec, c, conv, acc, worst);
  }
  bad = worst > c.kmax;
  return acc;
}

// Final: interpolate onto the observed grid, blaze, chi^2 partial per thread.
// `conv` = smoothed spectrum on the candidate's log grid (do_smooth) or the
// (rotated) spectrum on the ANN grid (plain np.interp branch, ystpred.py:271-272).
// force_grid: `conv` lives on the uniform log grid W describes whatever S.do_smooth says (the rotation stage's own resampled grid:
// smoothspec('vsini', outwave=...), smoothing.py:293-312)
// `first`: the thread's first block of records, requested ahead by the caller (only with obs_fast_ok and chi^2 alone: see run_candidate)
template <int OU = 16>
PAYNE_HD double phase_obs(int tid, int nthr, const PostTables& T, const CandState& S, const Window& W,
                          const float* __restrict__ conv, float* __restrict__ out, int out_stage, bool force_grid = false,
                          const ObsRec (*first)[OU] = nullptr) {
  const bool cheb = T.npoly > 0 && !force_grid, hasf = T.obs_f1 != nullptr, smooth = S.do_smooth != 0 || force_grid;
  if (!out && !hasf) return 0.0;                         // nothing to produce
  if (smooth && W.bad) {                                 // window too small: every pixel NaN
    if (out) for (int i = tid; i < T.nobs; i += nthr) out[i] = nanf_();
    return hasf ? (double)nanf_() : 0.0;
  }
  // (the blaze variants four pixels at a time: their fp64 recurrences, eight side by side, are the register peak of the whole kernel,
  //  which sits at the 128 registers that still allow two workgroups per CU -- unrelated edits moved it between 126 and 129; a fit
  //  without a blaze polynomial must not pay for that with half its occupancy)
  constexpr int OUC = OU > 4 ? 4 : OU;
#define PAYNE_OBS(MODE_)                                                                              \
  (out ? (cheb ? (hasf ? obs_loop<MODE_, true, true, true, OUC>(tid, nthr, T, S, W, conv, out, out_stage)    \
                       : obs_loop<MODE_, true, false, true, OUC>(tid, nthr, T, S, W, conv, out, out_stage))  \
               : (hasf ? obs_loop<MODE_, false, true, true, OU>(tid, nthr, T, S, W, conv, out, out_stage)   \
                       : obs_loop<MODE_, false, false, true, OU>(tid, nthr, T, S, W, conv, out, out_stage))) \
       : (cheb ? obs_loop<MODE_, true, true, false, OUC>(tid, nthr, T, S, W, conv, out, out_stage)           \
               : obs_loop<MODE_, false, true, false, OU>(tid, nthr, T, S, W, conv, out, out_stage)))
  float acc;
  // chi^2 only, no blaze, smoothed spectrum: the short loop on whole blocks of the padded table (if the padding is not most of
  // the work), the general one for a wave that met a pixel at or outside the window's ends
  bool general = true;
  if (smooth && !cheb && hasf && !out && obs_fast_ok(T, OU * nthr)) {
    bool bad;
    acc = obs_loop_fast<OU>(tid, nthr, T, W, conv, bad, first);
    general = wave_any(bad);
  }
  if (general) {
    if (smooth) acc = PAYNE_OBS(0);
    else if (T.geo) acc = PAYNE_OBS(1);
    else acc = PAYNE_OBS(2);
  }
#undef PAYNE_OBS
  return (double)acc;
}

// chi^2 partial of the thread -> one partial per slot in red[] (GPU: wave shuffle reduction)
#ifdef __HIP_DEVICE_COMPILE__
// Sum of a double over the wave, in lane 63: an inclusive scan inside each row of 16 lanes (row_shr 1, 2, 4, 8), then the rows'
// totals handed on (row_bcast 15, 31) -- data-parallel-primitive moves of the two halves and one fp64 add per level, 18 vector
// instructions and no LDS crossbar (the shuffle form: six dependent ds_bpermute round trips of ~100 cycles each, 40 instructions,
// at the very end of every workgroup's life).
template <int CTRL>
__device__ __forceinline__ double dpp_add_f64(double x) {
  union { double d; int i[2]; } a, b;
  a.d = x;
  b.i[0] = __builtin_amdgcn_update_dpp(a.i[0], a.i[0], CTRL, 0xf, 0xf, true);   // (bound_ctrl: a lane without a source reads 0)
  b.i[1] = __builtin_amdgcn_update_dpp(a.i[1], a.i[1], CTRL, 0xf, 0xf, true);
  return x + b.d;
}
__device__ __forceinline__ double wave_sum_to_last(double x) {
  x = dpp_add_f64<0x111>(x);                             // row_shr:1
  x = dpp_add_f64<0x112>(x);                             // row_shr:2
  x = dpp_add_f64<0x114>(x);                             // row_shr:4
  x = dpp_add_f64<0x118>(x);                             // row_shr:8   -> lane 15 of every row holds the row's sum
  // (every row takes part in the last two levels: rows 0 and 2 collect sums nobody reads, lanes 31 and then 63 the right ones)
  x = dpp_add_f64<0x142>(x);                             // row_bcast:15 -> lane 31 = rows 0 + 1, lane 63 = rows 2 + 3
  x = dpp_add_f64<0x143>(x);                             // row_bcast:31 -> lane 63 holds the wave's sum
  return x;
}
#endif
PAYNE_HD void store_partial(int tid, double acc, double* red) {
#ifdef __HIP_DEVICE_COMPILE__
  acc = wave_sum_to_last(acc);
  if ((tid & 63) == 63) red[tid >> 6] = acc;
#else
  red[tid] = acc;
#endif
}

}  // namespace payne
