// post_onchip2.hpp -- the register-resident convolution stage of post_onchip.hpp for 32 768-pixel spectra, TWO candidates at a time.
//
// A 32 768-point stage is 16 384 complex points = 32 x 16 x 32: not a cube, so the three square 32 x 32 transposes that carry the
// 65 536-point stage (register index <-> a digit of the thread index) do not apply to ONE such spectrum.  They do apply to TWO: the
// candidate index c takes the top bit of the 32-valued thread digit h = 16 c + h', and everything that moves data -- the loads, the
// radix-32 stages over the register index, the HI / LO exchanges, the taper's pair exchange -- is the 65 536-point code unchanged:
//
//   n = t' + 512 a              (virtual thread vt = 32 h + l, h = 16 c + h', t' = 32 h' + l = vt & 511; register a)
//   S1: DFT_32 over a -> k1     T1: x W_16384^(t' k1)        X1: (h, l; reg k1) <-> (k1, l; reg h = 16 c + h')
//   S2: TWO DFT_16 over h' -> k2a (one per candidate, on the two halves of the registers)
//                               T2: x W_512^(l k2a)          X2: (k1, l; reg (c, k2a)) <-> (k1, (c, k2a); reg l)
//   S3: DFT_32 over l -> k2b    => thread (k1, (c, k2a)), register k2b holds Z_c[k], k = k1 + 32 k2a + 512 k2b
//   P : pairs (k, M - k) of ONE candidate: thread low = k1 + 32 k2a meets thread 512 - low of the same c (register 31 - k2b)
//   then the transposed transform, as there.
//
// Both virtual threads of a real thread (h = 2 i and 2 i + 1, i = tid / 32) belong to candidate c = tid / 256: the first four waves
// of the workgroup carry candidate 0, the other four candidate 1; what differs between the candidates on the way in and out
// (pointers, resampling window) is wave-uniform; in the spectrum the candidate is bit 2 of l (a lane property).  Global traffic of
// a stage: each candidate's input once, its output once -- as the 65 536-point kernel.
// Device code only (included by post_kernels.hpp under PAYNE_TU_CHIP2).
#pragma once
#include "post_onchip.hpp"
namespace payne {
constexpr int kChip2M = 16384;                     // complex points of one candidate's stage (n1 = 32768 real)
constexpr int kChip2N1 = 32768;
}
#ifdef __HIP_DEVICE_COMPILE__
namespace payne {

// After S2 the register at PHYSICAL position p holds (c, k2a) with p = 8 (k2a >> 2) + 4 c + (k2a & 3)  [see chip2_dft16x2_fwd]
__device__ constexpr int chip2_k2a(int p) { return 4 * (p >> 3) + (p & 3); }
__device__ constexpr int chip2_c(int p) { return (p >> 2) & 1; }

// position 8 i + 4 c + ka (i, ka = 1..3) x W_16^(i ka)
__device__ __forceinline__ void chip2_tw16(c32 (&u)[32]) {
  // cos / sin of 2 pi m / 16, m = 0..9
  constexpr float C[10] = {1.0f, 0.92387953251128675613f, 0.70710678118654752440f, 0.38268343236508977173f, 0.0f,
                           -0.38268343236508977173f, -0.70710678118654752440f, -0.92387953251128675613f, -1.0f, -0.92387953251128675613f};
  constexpr float S[10] = {0.0f, 0.38268343236508977173f, 0.70710678118654752440f, 0.92387953251128675613f, 1.0f,
                           0.92387953251128675613f, 0.70710678118654752440f, 0.38268343236508977173f, 0.0f, -0.38268343236508977173f};
#pragma unroll
  for (int i = 1; i < 4; ++i) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const pk2 c1 = {C[i], S[i]}, c2 = {C[2 * i], S[2 * i]}, c3 = {C[3 * i], S[3 * i]};
      pk2 a = to_pk(u[8 * i + 4 * c + 1]), b = to_pk(u[8 * i + 4 * c + 2]);
      pk_cmul_k2(a, c1, b, c2);
      u[8 * i + 4 * c + 1] = un_pk(a); u[8 * i + 4 * c + 2] = un_pk(b);
      u[8 * i + 4 * c + 3] = un_pk(pk_cmul_k(to_pk(u[8 * i + 4 * c + 3]), c3));
    }
  }
}
// In: the PERMUTED layout the HI exchange leaves (logical register r = 16 c + h' at chip_pos(true, r) = 8 (h' & 3) + 4 c + (h' >> 2)),
// i.e. with h' = i + 4 j candidate c's value sits at 8 i + 4 c + j.  X[ka + 4 kb] = sum_i W4^(i kb) W16^(i ka) [ sum_j x[i + 4 j] W4^(j ka) ]:
// DFT_4 over j (four contiguous registers), twiddle, DFT_4 over i (stride 8).  Out: k2a = ka + 4 kb at 8 kb + 4 c + ka.
__device__ __forceinline__ void chip2_dft16x2_fwd(c32 (&u)[32]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    dft4(u[8 * i], u[8 * i + 1], u[8 * i + 2], u[8 * i + 3]);
    dft4(u[8 * i + 4], u[8 * i + 5], u[8 * i + 6], u[8 * i + 7]);
    __builtin_amdgcn_sched_barrier(0);
  }
  chip2_tw16(u);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < 8; ++q) {                                  // q = 4 c + ka
    dft4(u[q], u[q + 8], u[q + 16], u[q + 24]);
    if (q & 1) __builtin_amdgcn_sched_barrier(0);
  }
}
// The transposed operator: in k2a = ka + 4 kb at 8 kb + 4 c + ka, out h' = i + 4 j at 8 i + 4 c + j (the PERMUTED layout the HI
// exchange takes):  y[i + 4 j] = sum_ka W4^(j ka) W16^(i ka) [ sum_kb X[ka + 4 kb] W4^(i kb) ].
__device__ __forceinline__ void chip2_dft16x2_back(c32 (&u)[32]) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    dft4(u[q], u[q + 8], u[q + 16], u[q + 24]);
    if (q & 1) __builtin_amdgcn_sched_barrier(0);
  }
  chip2_tw16(u);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    dft4(u[8 * i], u[8 * i + 1], u[8 * i + 2], u[8 * i + 3]);
    dft4(u[8 * i + 4], u[8 * i + 5], u[8 * i + 6], u[8 * i + 7]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Twiddles: the context's full-circle table of a 32 768-pixel model is tw[j] = exp(-2 pi i j / 32768).  Two LDS tables as in the
// 65 536-point kernel -- w1024[j] = W_1024^j = tw[32 j] and the fine steps wfine[j] = W_32768^j = tw[j], j < 32 -- and one product:
// W_32768^e = w1024[e >> 5] wfine[e & 31].
__device__ __forceinline__ void chip2_fill_tables(const ChipLds& L, const c32* __restrict__ tw, int tid) {
  for (int j = tid; j < 1024; j += kChipThreads) stc(L.w1024, j, tw[32 * j]);
  if (tid < 32) stc(L.wfine, tid, tw[tid]);
}
__device__ __forceinline__ c32 chip2_w32768(const ChipLds& L, int e) {           // exp(-2 pi i e / 32768), 0 <= e < 32768
  const c32 a = ldc(L.w1024, (e >> 5) & 1023), b = ldc(L.wfine, e & 31);
  return cmul(a, b);
}
// x W_16384^(t' k1) = W_32768^(2 t' k1), t' = vt & 511 (registers in the layout PERM)
template <bool PERM>
__device__ __forceinline__ void chip2_tw1(const ChipLds& L, c32 (&u)[32], int vt_) {
  const int t2 = 2 * (chip_fresh(vt_) & 511);
#pragma unroll
  for (int k1 = 1; k1 < 31; k1 += 2) {
    const int e0 = (t2 * k1) & 32767, e1 = (t2 * (k1 + 1)) & 32767;
    chip_mul2x2(u[chip_pos(PERM, k1)], u[chip_pos(PERM, k1 + 1)], ldc(L.w1024, (e0 >> 5) & 1023), ldc(L.wfine, e0 & 31),
                ldc(L.w1024, (e1 >> 5) & 1023), ldc(L.wfine, e1 & 31));
    if ((k1 & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
  u[chip_pos(PERM, 31)] = cmul(u[chip_pos(PERM, 31)], chip2_w32768(L, (t2 * 31) & 32767));
}
// x W_512^(l k2a) = W_1024^(2 l k2a) on the register at physical position p = (c, k2a)
__device__ __forceinline__ void chip2_tw2(const ChipLds& L, c32 (&u)[32], int vt_) {
  const int l2 = 2 * (chip_fresh(vt_) & 31);
#pragma unroll
  for (int p = 1; p < 32; ++p) {
    if (chip2_k2a(p) == 0) continue;                              // p = 4: candidate 1's k2a = 0
    u[p] = cmul(u[p], ldc(L.w1024, (l2 * chip2_k2a(p)) & 1023));
    if ((p & 7) == 7) __builtin_amdgcn_sched_barrier(0);
  }
}

// natural (n = t' + 512 a per candidate) -> spectrum (thread (k1, (c, k2a)), register k2b in the PERMUTED layout) and back
__device__ __forceinline__ void chip2_fft_fwd(const ChipLds& L, c32 (&u0)[32], c32 (&u1)[32], int vt0) {
  const int vt1 = vt0 + 32;
  chip_dft32_A(u0); chip_pin(u0); chip2_tw1<true>(L, u0, vt0); chip_pin(u0); chip_dft32_A(u1); chip_pin(u1); chip2_tw1<true>(L, u1, vt1); chip_pin(u1);
  chip_xch<true, true>(L, u0, u1, vt0);
  chip2_dft16x2_fwd(u0); chip_pin(u0); chip2_tw2(L, u0, vt0); chip_pin(u0); chip2_dft16x2_fwd(u1); chip_pin(u1); chip2_tw2(L, u1, vt1); chip_pin(u1);
  chip_xch<false, false>(L, u0, u1, vt0);
  chip_dft32_A(u0); chip_pin(u0); chip_dft32_A(u1); chip_pin(u1);
}
__device__ __forceinline__ void chip2_fft_back(const ChipLds& L, c32 (&u0)[32], c32 (&u1)[32], int vt0) {
  const int vt1 = vt0 + 32;
  chip_dft32_B(u0); chip_pin(u0); chip_dft32_B(u1); chip_pin(u1);
  chip_xch<false, false>(L, u0, u1, vt0);
  chip2_tw2(L, u0, vt0); chip_pin(u0); chip2_dft16x2_back(u0); chip_pin(u0); chip2_tw2(L, u1, vt1); chip_pin(u1); chip2_dft16x2_back(u1); chip_pin(u1);
  chip_xch<true, true>(L, u0, u1, vt0);
  chip2_tw1<true>(L, u0, vt0); chip_pin(u0); chip_dft32_B(u0); chip_pin(u0); chip2_tw1<true>(L, u1, vt1); chip_pin(u1); chip_dft32_B(u1); chip_pin(u1);
}

// ---- the convolution's middle: thread (h = k1, l = (c, k2a)), register r = k2b holds Z_c[low + 512 r], low = k1 + 32 k2a ---------
__device__ __forceinline__ ChipPair chip2_pair(int vt_) {
  const int vt = chip_fresh(vt_);
  ChipPair p;
  const int h = vt >> 5, l = vt & 31;
  const int c = (l >> 2) & 1, k2a = 4 * (l >> 3) + (l & 3);
  p.low = h + 32 * k2a;
  const int plow = (512 - p.low) & 511;
  const int pk2a = plow >> 5;
  p.pt = 32 * (plow & 31) + (8 * (pk2a >> 2) + 4 * c + (pk2a & 3));   // the partner's virtual thread (itself for low = 0 and 256)
  p.t0 = p.low == 0;                                               // (k = 512 r) pairs r with 32 - r; r = 0 and 16 pair with themselves
  p.sh = p.t0 ? 1 : 0;
  return p;
}
// HAVE: the partner's values already sit in registers 16 + r (a row handed over transformed: host_tables.hpp chip2_layout)
template <bool VSINI, bool HAVE = false>
__device__ __forceinline__ void chip2_taper_pairs(const ChipLds& L, c32 (&u)[32], int vt, const TaperArgs& ta) {
  constexpr int M = kChip2M;
  const float invM = 1.0f / (float)M, g = 0.25f * invM;
  const ChipPair P = chip2_pair(vt);
  const c32 z0 = u[chip_pos(true, 0)];
  const c32 zh = HAVE ? u[chip_pos(true, 16)] : ldc(L.xch, chip_fresh(vt));
  if constexpr (!HAVE) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int slot = 15 - r + P.sh;
    slot = slot > 15 ? 15 : slot;
    u[chip_pos(true, 16 + r)] = ldc(L.xch, slot * 1024 + P.pt);
  }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int k0 = P.low + 512 * r;
    const int k = (k0 == 0) ? 1 : k0;
    float tk, tm;
    taper_full2<VSINI>(ta, k, M - k, tk, tm);
    const c32 w = chip2_w32768(L, k);                               // exp(-2 pi i k / 2M), 2M = 32768
    c32 yk, ym;
    c32& ua = u[chip_pos(true, r)];
    c32& ub = u[chip_pos(true, 16 + r)];
    taper_pair(ua, ub, w, tk * g, tm * g, yk, ym);
    ua = yk; ub = ym;
    { f2v t; t.x = ua.x; t.y = ua.y; f2v v2; v2.x = ub.x; v2.y = ub.y; asm volatile("" : "+v"(t), "+v"(v2)); ua = {t.x, t.y}; ub = {v2.x, v2.y}; }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (P.t0) {
    const float tM = taper_full<VSINI>(ta, M), th = taper_full<VSINI>(ta, M / 2);
    const float x0 = z0.x + z0.y, xm = tM * (z0.x - z0.y);
    u[chip_pos(true, 0)] = {0.5f * (x0 + xm) * invM, -0.5f * (x0 - xm) * invM};
    u[chip_pos(true, 16)] = cscale(cconj(zh), th * invM);
  }
}
__device__ __forceinline__ void chip2_taper_recv2(const ChipLds& L, c32 (&u)[32], int vt) {
  const ChipPair P = chip2_pair(vt);
  const c32 yh = u[chip_pos(true, 16)];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    int slot = 15 - j + P.sh;
    slot = slot > 15 ? 15 : slot;
    u[chip_pos(true, 16 + j)] = ldc(L.xch, slot * 1024 + P.pt);
  }
  if (P.t0) u[chip_pos(true, 16)] = yh;
}
// (after S3 the candidate of a virtual thread is bit 2 of l -- NOT the half of the workgroup it started in: the taper's arguments
//  are chosen per virtual thread here, by lane)
template <bool VSINI, bool HAVE = false>
__device__ __forceinline__ void chip2_taper(const ChipLds& L, c32 (&u0)[32], c32 (&u1)[32], int vt0, const TaperArgs& ta0, const TaperArgs& ta1) {
  const int vt1 = vt0 + 32;
  if constexpr (!HAVE) {
    chip_taper_send1(L, u0, vt0); chip_taper_send1(L, u1, vt1);
    __syncthreads();
  }
  // both virtual threads of a thread have the same l, hence the same candidate; what differs between the candidates' tapers are
  // three scalars (the table is the context's)
  const bool second = ((chip_fresh(vt0) >> 2) & 1) != 0;
  TaperArgs ta = ta0;
  ta.vs_c64 = second ? ta1.vs_c64 : ta0.vs_c64;
  ta.vs_c = second ? ta1.vs_c : ta0.vs_c;
  ta.g_c2 = second ? ta1.g_c2 : ta0.g_c2;
  chip2_taper_pairs<VSINI, HAVE>(L, u0, vt0, ta); chip_pin(u0); chip2_taper_pairs<VSINI, HAVE>(L, u1, vt1, ta); chip_pin(u1);
  if constexpr (!HAVE) __syncthreads();                            // (round 1's slots are read no more)
  chip_taper_send2(L, u0, vt0); chip_taper_send2(L, u1, vt1);
  __syncthreads();
  chip2_taper_recv2(L, u0, vt0); chip_pin(u0); chip2_taper_recv2(L, u1, vt1); chip_pin(u1);
  __syncthreads();
}

struct Chip2Io { const float* in[2]; float* out[2]; };
// One stage of BOTH candidates: input (real, global) -> registers -> convolution -> output (real, global).  rs: the two resampling
// windows (LDS) when the stage gathers its own input, else null.
template <bool VSINI>
__device__ __attribute__((noinline)) void chip2_conv(const ChipLds L, const Chip2Io io, const TaperArgs ta0, const TaperArgs ta1,
                                                     bool scrub, bool edge, int tid, const ChipResample* rs, bool zin) {
  typedef float f2g __attribute__((ext_vector_type(2)));
  const int c = __builtin_amdgcn_readfirstlane(tid >> 8);           // this wave's candidate while the data is in "time" order
  const float* in = c ? io.in[1] : io.in[0];
  float* out = c ? io.out[1] : io.out[0];
  const PAYNE_AS_GLOBAL f2g* g = (const PAYNE_AS_GLOBAL f2g*)in;
  const int vt0 = 64 * (tid >> 5) + (tid & 31);
  const int tp0 = vt0 & 511;                                        // t' of the first virtual thread (the second: + 32)
  c32 u0[32], u1[32];
  if (zin) {
    // the rows are the TRANSFORMS of the two spectra in the order the taper wants them (chip2_layout): in the spectrum the
    // candidate of a virtual thread is bit 2 of its lane digit -- each lane reads its own candidate's row
    typedef float f4g __attribute__((ext_vector_type(4)));
    const int lz = vt0 & 31, k2a = 4 * (lz >> 3) + (lz & 3), s0 = 16 * (vt0 >> 5) + k2a;
    const PAYNE_AS_GLOBAL f4g* gz = (const PAYNE_AS_GLOBAL f4g*)(((lz >> 2) & 1) ? io.in[1] : io.in[0]);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const f4g v = gz[512 * r + s0], w = gz[512 * r + s0 + 16];
      u0[chip_pos(true, r)] = {v.x, v.y}; u0[chip_pos(true, 16 + r)] = {v.z, v.w};
      u1[chip_pos(true, r)] = {w.x, w.y}; u1[chip_pos(true, 16 + r)] = {w.z, w.w};
    }
    if (scrub) { chip_scrub(u0); chip_scrub(u1); }
    chip_pin(u0); chip_pin(u1);
    chip2_taper<VSINI, true>(L, u0, u1, vt0, ta0, ta1);
  } else {
  if (rs) {
    const ChipResample R = rs[c];
    chip_gather_t<512>(in, R, tp0, u0);                             // (candidate c's masked, Doppler-shifted spectrum onto its pow-2 log grid)
    chip_gather_t<512>(in, R, tp0 + 32, u1);
  } else {
#pragma unroll
    for (int a = 0; a < 32; ++a) {
      const PAYNE_AS_GLOBAL f2g* ga = g + 512 * a;
      const PAYNE_AS_GLOBAL f2g* gb = g + 512 * a + 32;
      const f2g v = ga[tp0], w = gb[tp0];
      u0[a] = {v.x, v.y}; u1[a] = {w.x, w.y};
    }
  }
  if (scrub && !rs) { chip_scrub(u0); chip_scrub(u1); }
  chip_pin(u0); chip_pin(u1);
  chip2_fft_fwd(L, u0, u1, vt0);
  chip2_taper<VSINI>(L, u0, u1, vt0, ta0, ta1);
  }
  chip2_fft_back(L, u0, u1, vt0);
  chip_pin(u0); chip_pin(u1);
  PAYNE_AS_GLOBAL f2g* o = (PAYNE_AS_GLOBAL f2g*)out;
#pragma unroll
  for (int a = 0; a < 32; ++a) {
    f2g v, w;
    v.x = u0[a].x; v.y = -u0[a].y; w.x = u1[a].x; w.y = -u1[a].y;
    if (edge && a == 0 && tp0 == 0) v.x = v.y;                      // element 0     = (spec[0], spec[1])
    if (edge && a == 31 && tp0 + 32 == 511) w.y = w.x;              // element M - 1 = (spec[n-2], spec[n-1])
    PAYNE_AS_GLOBAL f2g* oa = o + 512 * a;
    PAYNE_AS_GLOBAL f2g* ob = o + 512 * a + 32;
    oa[tp0] = v; ob[tp0] = w;
  }
  __syncthreads();
}

}  // namespace payne
#endif
