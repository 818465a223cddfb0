// post_onchip.hpp -- one FFT convolution stage of a 65 536-pixel spectrum WITHOUT leaving the compute unit.
//
// payne_post_big_kernel keeps the two spectrum buffers of a candidate in a global workspace; its four-step transforms move
// the 256 KB spectrum through that workspace twelve times per convolution stage (25 transfers per candidate in all: the
// HBM / L2-bound regime of SURVEY 8(d)).  Here the 32 768 complex points z[n] = s[2n] + i s[2n+1] of a stage live in the
// REGISTERS of the workgroup from the load of the stage's input to the store of its output, and LDS (128 KB) is only the
// transpose buffer between the three radix-32 register stages of a transform.  The algorithm is written for 1024 "virtual
// threads" of 32 points each; a 512-thread workgroup runs two of them per thread (t and t + 512: 128 of its 256 registers hold
// data -- 1024 real threads would have 128 registers each, and the compiler spills two hundred of them):
//
//   n = t + 1024 a   (thread t = 32 h + l, register a)                                      [layout L0]
//   S1: DFT_32 over a -> k1        T1: x W_32768^(t k1)         X1: (h, l; reg k1)  <-> (k1, l; reg h)
//   S2: DFT_32 over h -> k2a       T2: x W_1024^(l k2a)         X2: (h, l; reg k2a) <-> (h, k2a; reg l)
//   S3: DFT_32 over l -> k2b       => thread (k1, k2a), register k2b holds Z[k], k = k1 + 32 k2a + 1024 k2b
//   P : the partner of thread low = k1 + 32 k2a is thread 1024 - low (register 31 - k2b): each sends half of its
//       registers, so every conjugate pair (k, M - k) meets in one thread, which applies the taper and the real-FFT
//       split / merge (taper_pair of post_core.hpp) and sends the partner's half back
//   then the TRANSPOSED transform (the DFT matrix is symmetric): S3, X2, T2, S2, X1, T1, S1 -> layout L0, conjugate.
//
// X1 / X2 are 32 x 32 transposes between the register index and a digit of the thread index; done in two rounds of 16
// registers chosen by the PARITY of that digit ((q + h) & 1), each thread sends sixteen registers and receives sixteen into
// the same slots: 64 VGPRs of data at every moment, 128 KB of LDS per round.  Global traffic of a stage: its input once, its
// output once.  Twiddles: W_65536^j from two LDS tables (j >> 6: 1024 entries, j & 63: 64 entries), one complex product.
// Device code only (included by post_kernels.hpp under PAYNE_TU_BIG).
#pragma once
namespace payne {
constexpr int kChipThreads = 512;                  // real threads; kChipVT virtual ones
constexpr int kChipVT = 1024;
constexpr int kChipM = 32768;                      // complex points of a stage (n1 = 65536 real)
constexpr int kChipXch = 16 * 1024;                // complex slots of an exchange round
constexpr size_t kChipLdsBytes = 256 + (size_t)kChipXch * 8 + 1024 * 8 + 64 * 8;   // (alignment) | exchange buffer | W_1024 | W_65536 (fine)
}
#ifdef __HIP_DEVICE_COMPILE__
namespace payne {

struct ChipLds {             // (LDS address space in the pointer types: ds_read / ds_write, not flat accesses)
  PAYNE_AS_LDS f2v* xch;     // [16][32][32]
  PAYNE_AS_LDS f2v* w1024;   // exp(-2 pi i j / 1024), j < 1024
  PAYNE_AS_LDS f2v* wfine;   // exp(-2 pi i j / 65536), j < 64
};
__device__ __forceinline__ ChipLds chip_lds(unsigned char* base) {
  ChipLds L;
  // (the exchange buffer on a 256-byte boundary: chip_xlo's slot addresses are then one XOR away from a per-call constant)
  L.xch = (PAYNE_AS_LDS f2v*)(((unsigned)(uintptr_t)(PAYNE_AS_LDS unsigned char*)base + 255u) & ~255u);
  L.w1024 = L.xch + kChipXch;
  L.wfine = L.w1024 + 1024;
  return L;
}
// the two tables from the context's full-circle table tw[j] = exp(-2 pi i j / 65536) (every thread a share; barrier by the caller)
__device__ __forceinline__ void chip_fill_tables(const ChipLds& L, const c32* __restrict__ tw, int tid) {
  for (int j = tid; j < 1024; j += kChipThreads) stc(L.w1024, j, tw[64 * j]);
  if (tid < 64) stc(L.wfine, tid, tw[tid]);
}

__device__ __forceinline__ c32 chip_w65536(const ChipLds& L, int e) {           // exp(-2 pi i e / 65536), 0 <= e < 65536
  const c32 a = ldc(L.w1024, (e >> 6) & 1023), b = ldc(L.wfine, e & 63);
  return cmul(a, b);
}

// Every value of a thread's share named at this point of the instruction stream: computations are neither sunk past it nor
// pulled above it (without it the compiler carries the twiddle products of a stage into the exchange that follows and keeps
// a hundred table values alive meanwhile: 326 registers spilled in the forward transform alone).
__device__ __forceinline__ void chip_pin(c32 (&u)[32]) {
  // (as register PAIRS: the packed instructions want a complex value in two consecutive, even-aligned registers; pinned as
  //  two separate values the halves land anywhere and every packed operation starts with moves)
#pragma unroll
  for (int i = 0; i < 32; ++i) { f2v t; t.x = u[i].x; t.y = u[i].y; asm volatile("" : "+v"(t)); u[i] = {t.x, t.y}; }
}

// ---- DFT_32 in registers (forward sign), in two forms that hand the result to each other WITHOUT moving it: ---------------
//   A  natural in -> "permuted" out: logical index k sits at position P(k) = 8 (k & 3) + (k >> 2)
//        X[k1 + 4 k2] = sum_a2 W32^(a2 k1) [ sum_a1 x[8 a1 + a2] W4^(a1 k1) ] W8^(a2 k2)      (DFT_4 strided, twiddle, DFT_8 contiguous)
//   B  permuted in -> natural out:
//        X[ka + 8 kb] = sum_alo W4^(alo kb) W32^(alo ka) [ sum_ahi x[alo + 4 ahi] W8^(ahi ka) ] (DFT_8 contiguous, twiddle, DFT_4 strided)
// Everything between two transforms (twiddles, exchanges, the taper) addresses the registers through the layout it was
// handed (template parameter PERM): the 64 register moves per transform of a reordering copy are a quarter of a stage's
// vector instructions.
constexpr int chip_pos(bool perm, int k) { return perm ? (8 * (k & 3) + (k >> 2)) : k; }
__device__ __forceinline__ void chip_tw32(c32 (&u)[32]) {             // position 8 i + j (i = 1..3, j = 1..7) x W32^(i j): both forms
  constexpr float C[22] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f,
                           0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f, 0.0f, -0.19509032201612826785f,
                           -0.38268343236508977173f, -0.55557023301960222474f, -0.70710678118654752440f, -0.83146961230254523708f,
                           -0.92387953251128675613f, -0.98078528040323044913f, -1.0f, -0.98078528040323044913f, -0.92387953251128675613f,
                           -0.83146961230254523708f, -0.70710678118654752440f, -0.55557023301960222474f};
  constexpr float S[22] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f, 0.70710678118654752440f,
                           0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f, 1.0f, 0.98078528040323044913f,
                           0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f, 0.55557023301960222474f,
                           0.38268343236508977173f, 0.19509032201612826785f, 0.0f, -0.19509032201612826785f, -0.38268343236508977173f,
                           -0.55557023301960222474f, -0.70710678118654752440f, -0.83146961230254523708f};
#pragma unroll
  for (int i = 1; i < 4; ++i) {
#pragma unroll
    for (int j = 1; j < 7; j += 2) {                             // x exp(-2 pi i m / 32) = cos - i sin: the pairs sit in scalar registers
      const pk2 c0 = {C[i * j], S[i * j]}, c1 = {C[i * (j + 1)], S[i * (j + 1)]};
      pk2 a = to_pk(u[j + 8 * i]), b = to_pk(u[j + 1 + 8 * i]);
      pk_cmul_k2(a, c0, b, c1);
      u[j + 8 * i] = un_pk(a); u[j + 1 + 8 * i] = un_pk(b);
    }
    const pk2 c7 = {C[i * 7], S[i * 7]};
    u[7 + 8 * i] = un_pk(pk_cmul_k(to_pk(u[7 + 8 * i]), c7));
  }
}
// (scheduling fences between the pieces: half of a thread's registers hold the data; left alone the scheduler pulls every table
//  read of a stage forward and spills hundreds of registers)
__device__ __forceinline__ void chip_dft32_A(c32 (&u)[32]) {          // natural -> permuted
#pragma unroll
  for (int a2 = 0; a2 < 8; ++a2) { dft4(u[a2], u[a2 + 8], u[a2 + 16], u[a2 + 24]); if (a2 & 1) __builtin_amdgcn_sched_barrier(0); }
  chip_tw32(u);                                                  // element (a2, k1) at a2 + 8 k1
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) { dft8(&u[8 * k1]); __builtin_amdgcn_sched_barrier(0); }   // -> X[k1 + 4 k2] at 8 k1 + k2
}
__device__ __forceinline__ void chip_dft32_B(c32 (&u)[32]) {          // permuted -> natural
#pragma unroll
  for (int al = 0; al < 4; ++al) { dft8(&u[8 * al]); __builtin_amdgcn_sched_barrier(0); }    // over ahi -> ka at 8 alo + ka
  chip_tw32(u);                                                  // element (alo, ka) at ka + 8 alo
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ka = 0; ka < 8; ++ka) { dft4(u[ka], u[ka + 8], u[ka + 16], u[ka + 24]); if (ka & 1) __builtin_amdgcn_sched_barrier(0); }   // -> X[ka + 8 kb] at ka + 8 kb
}

// ---- the two transposes (each its own inverse) -----------------------------------------------------------------
// HI: (h, l; reg r) <-> (r, l; reg h).  Round q: the registers of parity (q + d) & 1 travel (d = h resp. l): send, workgroup
// barrier, receive, barrier (the barriers are the caller's: it runs two virtual threads between them).  PERM: the layout of the
// register array (chip_pos).  HPAR: the parity of h, known at compile time (a thread's two virtual threads have h = 2 i and
// 2 i + 1): the HI exchange then needs no selects; the LO one picks by the lane's parity.
// (the thread index is laundered at every use: the exchanges' LDS addresses are otherwise computed once for all of them and
//  kept in a hundred registers from the first exchange to the last)
__device__ __forceinline__ int chip_fresh(int v) { asm volatile("" : "+v"(v)); return v; }
template <bool HI, bool PERM, int HPAR>
__device__ __forceinline__ void chip_xch_send(const ChipLds& L, const c32 (&u)[32], int vt_, int q) {
  const int vt = chip_fresh(vt_);
  const int h = vt >> 5, l = vt & 31;
  const bool odd = HI ? (((q + HPAR) & 1) != 0) : (((q + l) & 1) != 0);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    c32 v;
    if constexpr (HI) {
      v = u[chip_pos(PERM, 2 * j + ((q + HPAR) & 1))];           // register r = 2 j + parity
    } else {
      const c32 e = u[chip_pos(PERM, 2 * j)], o = u[chip_pos(PERM, 2 * j + 1)];
      v = {odd ? o.x : e.x, odd ? o.y : e.y};
    }
    // HI: [j][h][l];  LO: [h][j][l ^ 2 j] (the reader's lanes differ in j: the XOR spreads them over the banks)
    const int idx = HI ? ((j * 32 + h) * 32 + l) : ((h * 16 + j) * 32 + (l ^ (2 * j)));
    stc(L.xch, idx, v);
  }
}
template <bool HI, bool PERM, int HPAR>
__device__ __forceinline__ void chip_xch_recv(const ChipLds& L, c32 (&u)[32], int vt_, int q) {
  const int vt = chip_fresh(vt_);
  const int h = vt >> 5, l = vt & 31;
  const int d = HI ? h : l;
  const bool odd = HI ? (((q + HPAR) & 1) != 0) : (((q + l) & 1) != 0);
  const int J = d >> 1;                                          // the slot the senders used for register r = d
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int s = 2 * j + (odd ? 1 : 0);                         // the sender's digit (the parity class this thread just sent)
    const int idx = HI ? ((J * 32 + s) * 32 + l) : ((h * 16 + J) * 32 + (s ^ (2 * J)));
    const c32 v = ldc(L.xch, idx);
    if constexpr (HI) {
      u[chip_pos(PERM, 2 * j + ((q + HPAR) & 1))] = v;
    } else {
      // (selects, not a branch around the store: a conditional store keeps the whole array in scratch memory)
      const c32 e = u[chip_pos(PERM, 2 * j)], o = u[chip_pos(PERM, 2 * j + 1)];
      u[chip_pos(PERM, 2 * j)] = {odd ? e.x : v.x, odd ? e.y : v.y};
      u[chip_pos(PERM, 2 * j + 1)] = {odd ? v.x : o.x, odd ? v.y : o.y};
    }
  }
}
// LO without selects: (h, l; reg k) <-> (h, k; reg l) permutes within one h, so the two virtual threads of a thread (h = 2 i and
// 2 i + 1) belong to two DISJOINT exchanges: one round per set, all 32 registers of a virtual thread in it (512 virtual threads x 32
// registers = the same 128 KB), no parity classes.  Slot [i][k][l ^ k] (8 bytes each): the writers' lanes differ in l, the readers' in
// k -- the XOR spreads both over the 32 bank pairs.  Byte address = (per-call constant) ^ 8 k + 256 k: one vector instruction per access.
template <bool PERM>
__device__ __forceinline__ void chip_xlo_send(const ChipLds& L, const c32 (&u)[32], int vt_) {
  const int vt = chip_fresh(vt_);
  const unsigned a0 = (unsigned)(uintptr_t)L.xch + (unsigned)((vt >> 6) * 8192 + (vt & 31) * 8);      // [i][0][l]
#pragma unroll
  for (int k = 0; k < 32; ++k)
    stc((PAYNE_AS_LDS f2v*)(uintptr_t)((a0 ^ (unsigned)(8 * k)) + (unsigned)(256 * k)), 0, u[chip_pos(PERM, k)]);
}
template <bool PERM>
__device__ __forceinline__ void chip_xlo_recv(const ChipLds& L, c32 (&u)[32], int vt_) {
  const int vt = chip_fresh(vt_);
  const int l = vt & 31;
  const unsigned a0 = ((unsigned)(uintptr_t)L.xch + (unsigned)((vt >> 6) * 8192 + l * 256)) ^ (unsigned)(8 * l);   // [i][k = l][0 ^ l]
#pragma unroll
  for (int r = 0; r < 32; ++r)
    u[chip_pos(PERM, r)] = ldc((PAYNE_AS_LDS f2v*)(uintptr_t)(a0 ^ (unsigned)(8 * r)), 0);                      // [i][l][r ^ l]
}
template <bool PERM>
__device__ __forceinline__ void chip_xlo(const ChipLds& L, c32 (&u0)[32], c32 (&u1)[32], int vt0) {
  chip_xlo_send<PERM>(L, u0, vt0);
  __syncthreads();
  chip_xlo_recv<PERM>(L, u0, vt0); chip_pin(u0);
  __syncthreads();
  chip_xlo_send<PERM>(L, u1, vt0 + 32);
  __syncthreads();
  chip_xlo_recv<PERM>(L, u1, vt0 + 32); chip_pin(u1);
  __syncthreads();
}
template <bool HI, bool PERM>
__device__ __forceinline__ void chip_xch(const ChipLds& L, c32 (&u0)[32], c32 (&u1)[32], int vt0) {
  if constexpr (!HI) { chip_xlo<PERM>(L, u0, u1, vt0); return; }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    chip_xch_send<HI, PERM, 0>(L, u0, vt0, q);
    __builtin_amdgcn_sched_barrier(0);
    chip_xch_send<HI, PERM, 1>(L, u1, vt0 + 32, q);
    __syncthreads();
    chip_xch_recv<HI, PERM, 0>(L, u0, vt0, q); chip_pin(u0);
    chip_xch_recv<HI, PERM, 1>(L, u1, vt0 + 32, q); chip_pin(u1);
    __syncthreads();
  }
}

// ---- twiddles between the stages -----------------------------------------------------------------------------
// (two products per statement -- pk_cmul2 of post_core.hpp: the second product's first instruction sits between the two
//  dependent instructions of the first)
__device__ __forceinline__ void chip_mul2(c32& a, c32 wa, c32& b, c32 wb) {
  pk2 pa = to_pk(a), pb = to_pk(b);
  pk_cmul2(pa, to_pk(wa), pb, to_pk(wb));
  a = un_pk(pa); b = un_pk(pb);
}
// (a, b) *= (wa la, wb lb) -- a two-table twiddle and its application as one statement (pk_cmul2x2)
__device__ __forceinline__ void chip_mul2x2(c32& a, c32& b, c32 wa, c32 la, c32 wb, c32 lb) {
  pk2 pa = to_pk(a), pb = to_pk(b), pwa = to_pk(wa), pwb = to_pk(wb);
  pk_cmul2x2(pa, pb, pwa, to_pk(la), pwb, to_pk(lb));
  a = un_pk(pa); b = un_pk(pb);
}
template <bool PERM>
__device__ __forceinline__ void chip_tw1(const ChipLds& L, c32 (&u)[32], int vt_) {     // x W_32768^(t k1) = W_65536^(2 t k1)
  const int vt = chip_fresh(vt_);
#pragma unroll
  for (int k1 = 1; k1 < 31; k1 += 2) {
    const int e0 = (2 * vt * k1) & 65535, e1 = (2 * vt * (k1 + 1)) & 65535;
    chip_mul2x2(u[chip_pos(PERM, k1)], u[chip_pos(PERM, k1 + 1)], ldc(L.w1024, (e0 >> 6) & 1023), ldc(L.wfine, e0 & 63),
                ldc(L.w1024, (e1 >> 6) & 1023), ldc(L.wfine, e1 & 63));
    if ((k1 & 3) == 3) __builtin_amdgcn_sched_barrier(0);        // four twiddles (eight table reads) in flight at a time
  }
  u[chip_pos(PERM, 31)] = cmul(u[chip_pos(PERM, 31)], chip_w65536(L, (2 * vt * 31) & 65535));
}
template <bool PERM>
__device__ __forceinline__ void chip_tw2(const ChipLds& L, c32 (&u)[32], int vt_) {     // x W_1024^(l k2a)
  const int l = chip_fresh(vt_) & 31;
#pragma unroll
  for (int k = 1; k < 31; k += 2) {
    chip_mul2(u[chip_pos(PERM, k)], ldc(L.w1024, (l * k) & 1023), u[chip_pos(PERM, k + 1)], ldc(L.w1024, (l * (k + 1)) & 1023));
    if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);
  }
  u[chip_pos(PERM, 31)] = cmul(u[chip_pos(PERM, 31)], ldc(L.w1024, (l * 31) & 1023));
}

// natural (layout L0) -> spectrum (thread (k1, k2a), reg k2b, PERMUTED register layout) and back (transposed order: the same
// operators); two virtual threads vt0 and vt0 + 32
__device__ __forceinline__ void chip_fft_fwd(const ChipLds& L, c32 (&u0)[32], c32 (&u1)[32], int vt0) {
  const int vt1 = vt0 + 32;
  chip_dft32_A(u0); chip_pin(u0); chip_tw1<true>(L, u0, vt0); chip_pin(u0); chip_dft32_A(u1); chip_pin(u1); chip_tw1<true>(L, u1, vt1); chip_pin(u1);
  chip_xch<true, true>(L, u0, u1, vt0);
  chip_dft32_B(u0); chip_pin(u0); chip_tw2<false>(L, u0, vt0); chip_pin(u0); chip_dft32_B(u1); chip_pin(u1); chip_tw2<false>(L, u1, vt1); chip_pin(u1);
  chip_xch<false, false>(L, u0, u1, vt0);
  chip_dft32_A(u0); chip_pin(u0); chip_dft32_A(u1); chip_pin(u1);
}
__device__ __forceinline__ void chip_fft_back(const ChipLds& L, c32 (&u0)[32], c32 (&u1)[32], int vt0) {
  const int vt1 = vt0 + 32;
  chip_dft32_B(u0); chip_pin(u0); chip_dft32_B(u1); chip_pin(u1);
  chip_xch<false, false>(L, u0, u1, vt0);
  chip_tw2<false>(L, u0, vt0); chip_pin(u0); chip_dft32_A(u0); chip_pin(u0); chip_tw2<false>(L, u1, vt1); chip_pin(u1); chip_dft32_A(u1); chip_pin(u1);
  chip_xch<true, true>(L, u0, u1, vt0);
  chip_tw1<true>(L, u0, vt0); chip_pin(u0); chip_dft32_B(u0); chip_pin(u0); chip_tw1<true>(L, u1, vt1); chip_pin(u1); chip_dft32_B(u1); chip_pin(u1);
}

// ---- the convolution's middle: conjugate pairs, taper, real-FFT split / merge ------------------------------------
// In: thread (h = k1, l = k2a), register r = k2b holds Z[k], k = h + 32 l + 1024 r.  Out: the same layout holds Y with
// FFT_M(Y) = conj(z'), z' the packed smoothed spectrum (rfft_taper_phase of post_core.hpp, same arithmetic per pair).
struct ChipPair { int low, pt, sh; bool t0; };
__device__ __forceinline__ ChipPair chip_pair(int vt_) {
  const int vt = chip_fresh(vt_);
  ChipPair p;
  const int h = vt >> 5, l = vt & 31;
  p.low = h + 32 * l;
  const int plow = (1024 - p.low) & 1023;
  p.pt = 32 * (plow & 31) + (plow >> 5);                           // the partner's (virtual) thread index (itself for low = 0 and 512)
  p.t0 = vt == 0;                                                  // thread 0 (k = 1024 r) pairs r with 32 - r; r = 0 and 16 pair with themselves
  p.sh = p.t0 ? 1 : 0;
  return p;
}
// round 1: the upper half of the registers goes to the partner (slot j = register 16 + j)
__device__ __forceinline__ void chip_taper_send1(const ChipLds& L, const c32 (&u)[32], int vt_) {
  const int vt = chip_fresh(vt_);
#pragma unroll
  for (int j = 0; j < 16; ++j) stc(L.xch, j * 1024 + vt, u[chip_pos(true, 16 + j)]);
}
// HAVE: the partner's values already sit in registers 16 + r (a row handed over transformed, host_tables.hpp chip_layout:
// register 16 of thread 0 is Z[M/2])
// FAR: a bin beyond the taper table (u >= 256: |vrot| of several tens of km/s on this grid, or NaN) is re-evaluated exactly by a CALL
// (taper_far -> vsini_sb_exact).  A call inside a stage function keeps the stage's sixty-odd live registers in callee-saved ones, and
// the function then saves and restores those through scratch memory on EVERY call (63 dwords a thread each way: 258 KB a candidate,
// a fifth of the bytes the C5 launch moved) -- the likelihood kernel's usual path checks the candidate's largest bin against the
// table first and runs the stage without that call (FAR = false); a candidate beyond the table takes the general sequence.
template <bool VSINI, bool HAVE = false, bool FAR = true>
__device__ __forceinline__ void chip_taper_pairs(const ChipLds& L, c32 (&u)[32], int vt, const TaperArgs& ta) {
  constexpr int M = kChipM;
  const float invM = 1.0f / (float)M, g = 0.25f * invM;
  const ChipPair P = chip_pair(vt);
  // (the registers are in the permuted layout the forward transform leaves: logical r at chip_pos(true, r))
  const c32 z0 = u[chip_pos(true, 0)];                             // thread 0: Z[0] (holds the real bins X[0] and X[M])
  const c32 zh = HAVE ? u[chip_pos(true, 16)] : ldc(L.xch, vt);    // thread 0: Z[M/2] (its register 16, slot 0)
  if constexpr (!HAVE) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int slot = 15 - r + P.sh;                                      // the partner's register 31 - r (thread 0: 32 - r)
    slot = slot > 15 ? 15 : slot;                                  // (thread 0, r = 0: not a pair, the value is not used)
    u[chip_pos(true, 16 + r)] = ldc(L.xch, slot * 1024 + P.pt);
  }
  }
  // the pairs (k, M - k), k = low + 1024 r, r < 16: taper and real-FFT split / merge
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int k0 = P.low + 1024 * r;
    const int k = (k0 == 0) ? 1 : k0;                              // (k = 0 is replaced below; keep its lane's arithmetic ordinary)
    float tk, tm;
    if constexpr (FAR) taper_full2<VSINI>(ta, k, M - k, tk, tm);
    else { bool far_ = false; taper_at2<VSINI>(ta, k, M - k, far_, tk, tm); }
    const c32 w = chip_w65536(L, k);                               // exp(-2 pi i k / 2M)
    c32 yk, ym;
    c32& ua = u[chip_pos(true, r)];
    c32& ub = u[chip_pos(true, 16 + r)];
    taper_pair(ua, ub, w, tk * g, tm * g, yk, ym);
    ua = yk; ub = ym;
    { f2v t; t.x = ua.x; t.y = ua.y; f2v v2; v2.x = ub.x; v2.y = ub.y; asm volatile("" : "+v"(t), "+v"(v2)); ua = {t.x, t.y}; ub = {v2.x, v2.y}; }
    __builtin_amdgcn_sched_barrier(0);                             // one pair (two taper values) at a time
  }
  if (P.t0) {                                                      // the self-conjugate bins (rfft_taper_phase, same statements)
    float tM, th;
    if constexpr (FAR) { tM = taper_full<VSINI>(ta, M); th = taper_full<VSINI>(ta, M / 2); }
    else { bool far_ = false; tM = taper_at<VSINI>(ta, M, far_); th = taper_at<VSINI>(ta, M / 2, far_); }
    const float x0 = z0.x + z0.y, xm = tM * (z0.x - z0.y);
    u[chip_pos(true, 0)] = {0.5f * (x0 + xm) * invM, -0.5f * (x0 - xm) * invM};
    u[chip_pos(true, 16)] = cscale(cconj(zh), th * invM);          // (Y[M/2]: parked in the slot that round 2 skips for thread 0)
  }
}
// round 2: the partner's halves of the pairs go back (slot r = the pair index of the thread that computed it)
__device__ __forceinline__ void chip_taper_send2(const ChipLds& L, const c32 (&u)[32], int vt_) {
  const int vt = chip_fresh(vt_);
#pragma unroll
  for (int r = 0; r < 16; ++r) stc(L.xch, r * 1024 + vt, u[chip_pos(true, 16 + r)]);
}
__device__ __forceinline__ void chip_taper_recv2(const ChipLds& L, c32 (&u)[32], int vt) {
  const ChipPair P = chip_pair(vt);
  const c32 yh = u[chip_pos(true, 16)];                            // thread 0: Y[M/2]
#pragma unroll
  for (int j = 0; j < 16; ++j) {                                   // my register 16 + j was the partner's pair 31 - (16 + j) = 15 - j
    int slot = 15 - j + P.sh;                                      // (thread 0: 32 - (16 + j) = 16 - j; j = 0 is Y[M/2])
    slot = slot > 15 ? 15 : slot;
    u[chip_pos(true, 16 + j)] = ldc(L.xch, slot * 1024 + P.pt);
  }
  if (P.t0) u[chip_pos(true, 16)] = yh;
}
template <bool VSINI, bool FAR = true>
__device__ __forceinline__ void chip_taper(const ChipLds& L, c32 (&u0)[32], c32 (&u1)[32], int tid, const TaperArgs& ta) {
  const int t1 = tid + 32;                                         // (tid: the first virtual thread's index)
  chip_taper_send1(L, u0, tid); chip_taper_send1(L, u1, t1);
  __syncthreads();
  chip_taper_pairs<VSINI, false, FAR>(L, u0, tid, ta); chip_pin(u0); chip_taper_pairs<VSINI, false, FAR>(L, u1, t1, ta); chip_pin(u1);
  __syncthreads();
  chip_taper_send2(L, u0, tid); chip_taper_send2(L, u1, t1);
  __syncthreads();
  chip_taper_recv2(L, u0, tid); chip_pin(u0); chip_taper_recv2(L, u1, t1); chip_pin(u1);
  __syncthreads();
}
// ... of a row that arrived as its transform with every pair side by side: no first round
template <bool VSINI, bool FAR = true>
__device__ __forceinline__ void chip_taper_have(const ChipLds& L, c32 (&u0)[32], c32 (&u1)[32], int tid, const TaperArgs& ta) {
  const int t1 = tid + 32;
  chip_taper_pairs<VSINI, true, FAR>(L, u0, tid, ta); chip_pin(u0); chip_taper_pairs<VSINI, true, FAR>(L, u1, t1, ta); chip_pin(u1);
  chip_taper_send2(L, u0, tid); chip_taper_send2(L, u1, t1);
  __syncthreads();
  chip_taper_recv2(L, u0, tid); chip_pin(u0); chip_taper_recv2(L, u1, t1); chip_pin(u1);
  __syncthreads();
}

// ---- one stage: input (real, global) -> registers -> convolution -> output (real, global) -------------------------
// scrub: NaN -> 0 on the way in (nan_to_num of the shifted flux, smoothing.py:138: the raw row); edge: spec[0] = spec[1],
// spec[-1] = spec[-2] on the way out (ystpred.py:223-224).  Every thread of the 1024 takes part (barriers inside).
// (a CALL: the stage needs 224 of the 256 registers a thread has; inlined into the candidate's phase sequence, what that keeps
//  alive around it costs three hundred spills)
// What the instrumental stage reads is the masked, Doppler-shifted spectrum resampled onto its pow-2 log grid (smoothing.py:649-668;
// R_resample_loop of post_core.hpp, geometric grids): with `rs` the stage's load gathers and interpolates its points itself
// (positions by one fp64 fma each, as there) instead of reading a resampled copy a phase of its own wrote -- two transfers of
// the spectrum and one phase fewer.
// tlo / thi: the positions (+ kPosMagic) of the window's first pixel and of "just below its last" -- a position clamped to
// [tlo, thi] has its integer part in [i0, i1 - 2] and its fraction in [0, 1): what magic_locate's integer clamps and exponent
// test produce, for two instructions instead of a dozen (positions of a valid window are finite: a window that is not is `bad`)
struct ChipResample { double rsA, rsBm, tlo, thi; float hs; int i0, i1; };
__device__ __forceinline__ ChipResample chip_resample_of(const Window& W) {
  ChipResample R;
  R.rsA = W.rsA; R.rsBm = W.rsB + kPosMagic; R.hs = W.hs_ann; R.i0 = W.i0; R.i1 = W.i1;
  R.tlo = kPosMagic + (double)W.i0;
  R.thi = kPosMagic + (double)(W.i1 - 1) - 2.3283064365386963e-10;       // one position ulp (2^-32) below the last pixel
  return R;
}
// STRIDE: complex points between two registers of a virtual thread (1024: one 65 536-point spectrum; 512: one of two 32 768-point ones)
template <int STRIDE>
__device__ __forceinline__ void chip_gather_t(const float* spec, const ChipResample& R, int vt, c32 (&u)[32]) {
  const double step = (double)(2 * STRIDE) * R.rsA;                 // real points 2 n and 2 n + 1 of n = vt + STRIDE a
  double te = fma((double)(2 * vt), R.rsA, R.rsBm), to = fma((double)(2 * vt + 1), R.rsA, R.rsBm);
#pragma unroll
  for (int a0 = 0; a0 < 32; a0 += 4) {                              // four complex points (sixteen gathers) at a time
    float va[8], vb[8], vw[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      double& t = (q & 1) ? to : te;
      union { double d; unsigned long long u; } cv;
      cv.d = fmin(fmax(t, R.tlo), R.thi);
      t += step;                                                    // (thirty-two roundings of 2^-33 pixel: nothing)
      const int k = (int)((unsigned)(cv.u >> 32) & 0x7FFFFu);
      const float f = (float)(unsigned)cv.u * 2.3283064365386963e-10f;
      vw[q] = f * (1.0f + R.hs * (f - 1.0f));
      va[q] = spec[k]; vb[q] = spec[k + 1];
    }
    nan_scrub8(va[0], va[1], va[2], va[3], va[4], va[5], va[6], va[7]);      // nan_to_num, smoothing.py:138
    nan_scrub8(vb[0], vb[1], vb[2], vb[3], vb[4], vb[5], vb[6], vb[7]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      u[a0 + q] = {va[2 * q] + (vb[2 * q] - va[2 * q]) * vw[2 * q], va[2 * q + 1] + (vb[2 * q + 1] - va[2 * q + 1]) * vw[2 * q + 1]};
      f2v t; t.x = u[a0 + q].x; t.y = u[a0 + q].y; asm volatile("" : "+v"(t)); u[a0 + q] = {t.x, t.y};
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
// NaN -> 0 on all sixty-four values of a virtual thread's share
__device__ __forceinline__ void chip_scrub(c32 (&u)[32]) {
#pragma unroll
  for (int a = 0; a < 32; a += 4) nan_scrub8(u[a].x, u[a].y, u[a + 1].x, u[a + 1].y, u[a + 2].x, u[a + 2].y, u[a + 3].x, u[a + 3].y);
}
template <bool VSINI, bool FAR = true>
__device__ __attribute__((noinline)) void chip_conv(const ChipLds L, const float* in, float* out, const TaperArgs ta,   // (in == out is allowed)
                                                    bool scrub, bool edge, int tid, const ChipResample* rs, bool zin) {
  typedef float f2g __attribute__((ext_vector_type(2)));
  const PAYNE_AS_GLOBAL f2g* g = (const PAYNE_AS_GLOBAL f2g*)in;
  // the thread's two virtual threads: vt0 = (h = 2 i, l), vt1 = (h = 2 i + 1, l), i = tid / 32, l = tid % 32
  const int vt0 = 64 * (tid >> 5) + (tid & 31);
  c32 u0[32], u1[32];
  if (zin) {
    // the row is the TRANSFORM of the spectrum (the output layer's weights carried it), in the order the taper wants it
    // (chip_layout): slot r * 1024 + vt = (Z[k], Z[M - k]), k = low + 1024 r -- no forward transform, no first exchange round
    typedef float f4g __attribute__((ext_vector_type(4)));
    const PAYNE_AS_GLOBAL f4g* g4 = (const PAYNE_AS_GLOBAL f4g*)in;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const PAYNE_AS_GLOBAL f4g* ga = g4 + 1024 * r;
      const PAYNE_AS_GLOBAL f4g* gb = g4 + 1024 * r + 32;
      const f4g v = ga[vt0], w = gb[vt0];
      u0[chip_pos(true, r)] = {v.x, v.y}; u0[chip_pos(true, 16 + r)] = {v.z, v.w};
      u1[chip_pos(true, r)] = {w.x, w.y}; u1[chip_pos(true, 16 + r)] = {w.z, w.w};
    }
    if (scrub) { chip_scrub(u0); chip_scrub(u1); }
    chip_pin(u0); chip_pin(u1);
    chip_taper_have<VSINI, FAR>(L, u0, u1, vt0, ta);
  } else {
  if (rs) {
    const ChipResample R = *rs;
    chip_gather_t<1024>(in, R, vt0, u0);
    chip_gather_t<1024>(in, R, vt0 + 32, u1);
  } else {
  // (uniform base + the thread's 32-bit offset: the 64 addresses live in scalar registers, not in 128 vector ones)
#pragma unroll
  for (int a = 0; a < 32; ++a) {
    const PAYNE_AS_GLOBAL f2g* ga = g + 1024 * a;
    const PAYNE_AS_GLOBAL f2g* gb = g + 1024 * a + 32;
    const f2g v = ga[vt0], w = gb[vt0];
    u0[a] = {v.x, v.y}; u1[a] = {w.x, w.y};
  }
  }
  if (scrub && !rs) { chip_scrub(u0); chip_scrub(u1); }
  chip_pin(u0); chip_pin(u1);
  chip_fft_fwd(L, u0, u1, vt0);
  chip_taper<VSINI, FAR>(L, u0, u1, vt0, ta);
  }
  chip_fft_back(L, u0, u1, vt0);
  chip_pin(u0); chip_pin(u1);
  PAYNE_AS_GLOBAL f2g* o = (PAYNE_AS_GLOBAL f2g*)out;
#pragma unroll
  for (int a = 0; a < 32; ++a) {
    f2g v, w;                                                      // z' = conj(FFT(Y))
    v.x = u0[a].x; v.y = -u0[a].y; w.x = u1[a].x; w.y = -u1[a].y;
    if (edge && a == 0 && vt0 == 0) v.x = v.y;                     // element 0     = (spec[0], spec[1])
    if (edge && a == 31 && vt0 + 32 == kChipVT - 1) w.y = w.x;     // element M - 1 = (spec[n-2], spec[n-1])
    PAYNE_AS_GLOBAL f2g* oa = o + 1024 * a;
    PAYNE_AS_GLOBAL f2g* ob = o + 1024 * a + 32;
    oa[vt0] = v; ob[vt0] = w;
  }
  __syncthreads();                                                 // (one CU, one L1: the next phase of this workgroup reads what was just written)
}

// ---- the instrumental stage and the observed grid in ONE call: the smoothed spectrum never leaves the compute unit ---------------
// chip_conv<false> with a gathered input, up to the inverse transform; then, instead of the store of its output and a phase that
// gathers from it (two transfers of the spectrum through the workspace, 2 x nobs 4-byte gathers at L2 distance): the registers hold
// the spectrum in layout L0 (thread t, register a = complex point t + 1024 a), and the exchange buffer holds exactly HALF of it --
// registers 0..15 first (real points 0..32767), the observed pixels that fall there interpolate from LDS, then registers 16..31, the
// rest.  The observed wavelengths ascend (PostTables::obs_sorted) and every one of them lies inside the window (the caller checks the
// two ends): the blocks of 16 x 512 padded records are walked once, the block that straddles the middle keeps its records in
// registers across the switch.  Same statements per pixel, same order of a thread's sum as obs_loop_fast: the same chi^2 to the bit.
// `edge`: one LDS float (real point 32768 for the last pixel of the first half).  Returns the thread's partial sum.
__device__ __attribute__((noinline)) float chip_conv_obs(const ChipLds L, const float* in, const TaperArgs ta, int tid, const ChipResample* rs,
                                                        const ObsRec* __restrict__ recs, int nobs, const ObsFastConsts oc,
                                                        PAYNE_AS_LDS float* edge) {
  const int vt0 = 64 * (tid >> 5) + (tid & 31);
  c32 u0[32], u1[32];
  {
    const ChipResample R = *rs;
    chip_gather_t<1024>(in, R, vt0, u0);
    chip_gather_t<1024>(in, R, vt0 + 32, u1);
  }
  chip_pin(u0); chip_pin(u1);
  chip_fft_fwd(L, u0, u1, vt0);
  chip_taper<false>(L, u0, u1, vt0, ta);
  chip_fft_back(L, u0, u1, vt0);
  chip_pin(u0); chip_pin(u1);
  // z' = conj(FFT(Y)): slot s of the exchange buffer = real points 2 s, 2 s + 1 of the half
#pragma unroll
  for (int a = 0; a < 16; ++a) {
    stc(L.xch, vt0 + 1024 * a, c32{u0[a].x, -u0[a].y});
    stc(L.xch, vt0 + 32 + 1024 * a, c32{u1[a].x, -u1[a].y});
  }
  if (vt0 == 0) *edge = u0[16].x;
  __syncthreads();
  const PAYNE_AS_LDS float* xf = (const PAYNE_AS_LDS float*)L.xch;
  constexpr int OU = 16, HALF = kChipM;                              // real points in a half = complex points of the stage
  float acc = 0.f;
  int round = 0;
  for (int base = tid; base < nobs; base += OU * kChipThreads) {
    ObsRec rec[OU];
#pragma unroll
    for (int q = 0; q < OU; ++q) rec[q] = recs[(unsigned)(base + q * kChipThreads)];
    unsigned kk[OU]; float F[OU];
    bool hi = false;
#pragma unroll
    for (int q = 0; q < OU; ++q) {
      union { double d; unsigned long long u; } cv;
      cv.d = fma(rec[q].lnw, oc.obA, oc.obBm);
      const unsigned k = (unsigned)(cv.u >> 32) - kPosMagicHi;
      kk[q] = k < oc.kmax ? k : oc.kmax;
      F[q] = (float)(unsigned)cv.u;
      hi = hi || kk[q] >= (unsigned)HALF;
    }
    for (;;) {                                                      // (at most twice: the block that straddles the middle)
      const float ev = *edge;
#pragma unroll
      for (int q = 0; q < OU; ++q) {
        const bool mine = (int)(kk[q] >> 15) == round;
        const unsigned idx = kk[q] & (unsigned)(HALF - 1);
        const float av = xf[idx];
        const float bv = idx == (unsigned)(HALF - 1) ? ev : xf[idx + 1 < (unsigned)HALF ? idx + 1 : idx];
        const float w = F[q] * fmaf(F[q], oc.c2, oc.c1);
        const float d = fmaf(bv - av, w, av) - rec[q].f1;
        acc = mine ? fmaf(d * d, rec[q].ivar, acc) : acc;
      }
      if (round == 1 || !__syncthreads_or(hi ? 1 : 0)) break;       // (uniform: everybody has finished with the first half)
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        stc(L.xch, vt0 + 1024 * a, c32{u0[16 + a].x, -u0[16 + a].y});
        stc(L.xch, vt0 + 32 + 1024 * a, c32{u1[16 + a].x, -u1[16 + a].y});
      }
      __syncthreads();
      round = 1;
    }
  }
  __syncthreads();                                                 // (the exchange buffer is the next candidate's)
  return acc;
}

}  // namespace payne
#endif
