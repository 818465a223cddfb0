// payne_hip.hip -- gfx950 kernels + the C ABI of include/payne_hip.h.
//
// Kernels (all hand-written for CDNA4, wave64):
//   payne_dense_kernel  fp32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32 fma chain) dense layer
//                       Y = act(X W^T + b) over the batch of candidates; LDS-tiled 64xBNx32,
//                       register-prefetch double buffering, XCD-aware tile order.  With FUSE_L0
//                       the A operand is produced on the fly from theta: label encoding
//                       (ystpred.py:47-50) + first layer + activation, so a YST1 forward pass
//                       (ystpred.py:52-58) is two launches.
//   payne_post_kernel   one 256-thread workgroup per candidate; the spectrum lives in LDS from
//                       the ANN output to chi^2: vsini FFT stage, Doppler, masked pow-2
//                       resample, Gaussian FFT stage, interpolation to the observed grid,
//                       blaze, chi^2 (phases in post_core.hpp, order in post_seq.hpp).
//   payne_sed_kernel    one wave per (candidate, filter): the stacked photometric nets of
//                       photANN.fastANN + highAv + the magnitude formulae of predictsed.py.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/payne_hip.h"
#include "host_tables.hpp"
#include "post_seq.hpp"
#include "ns_core.hpp"

using namespace payne;

// ============================================================================
// dense layer on the matrix cores
// ============================================================================
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));   // register-resident 16-byte value (HIP's f32x4_t struct arrays end up in scratch)

struct DenseParams {
  const float* X; int ldx;     // [B][ldx] activations (ignored with FUSE_L0)
  const float* W; int K;       // [N][K] row-major, K % 4 == 0
  const float* bias;           // [N]
  float* Y; int ldy;           // [B][ldy]
  int B, N;
  float bias_shift;            // subtracted from the bias (kBase on the output layer)
  int act;
  int grid_m, grid_n;
  // fused first layer (FUSE_L0): A[r][k] = act0(b0[k] + sum_d W0[k][d] * xhat[r][d])
  const double* theta; int ld_theta;
  const float* W0; const float* b0; int n_labels; int act0;
  int K0;                      // real width of the first layer (W0 has K0 rows)
  double xmin[PAYNE_MAX_LABELS], xden[PAYNE_MAX_LABELS];
  // optional second output of the hidden-layer kernel: the activations as three bf16 planes (x = x1 + x2 + x3
  // exactly), operand of payne_dense_bx3dma_kernel
  unsigned short* Yp; int ldyp; size_t yp_plane;   // [3][yp_plane] elements, row pitch ldyp
#ifdef PAYNE_STAMPS
  unsigned long long* stamps;  // diagnostic build: [grid][16] cycle stamps of the hidden-layer kernel
#endif
};
#ifdef PAYNE_STAMPS
#define HK_STAMP(k) do { if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
static unsigned long long* g_hidden_stamps = nullptr;
static unsigned long long* g_dense_stamps = nullptr;
#else
#define HK_STAMP(k) do {} while (0)
#endif

// Activations without per-lane branches: a `z > 0 ? .. : ..` chain compiles to exec-mask branches, and
// an unrolled epilogue then serialises on them (the fused first layer spent 11 500 of 22 500 cycles
// that way).  leaky ReLU(0.01) = max(z, 0.01 z) for every finite z, 0 and NaN (ystpred.py:57-58).
__device__ __forceinline__ float lrelu01(float z) { return fmaxf(z, 0.01f * z); }
__device__ __forceinline__ float act_apply(float z, int act) {
  if (act == PAYNE_ACT_SIGMOID) return 1.0f / (1.0f + expf(-z));          // (uniform: a scalar branch)
  const float l = lrelu01(z);
  return act == PAYNE_ACT_LRELU ? l : z;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is fence + s_barrier and the fence
// waits for EVERY outstanding memory operation (vmcnt(0)), so a global load issued two k-steps ahead
// would be waited for at the very next barrier; here only the LDS counter is drained and the loads
// stay in flight (the compiler still waits on vmcnt before the first use of their registers).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int BM, int BN, int BK>
constexpr size_t dense_lds_bytes() { return (size_t)(2 * (BM + BN) * (BK + 4) + BM * PAYNE_MAX_LABELS) * sizeof(float); }

template <int BM, int BN, int BK, bool FUSE_L0>
__global__ void __launch_bounds__(256) payne_dense_kernel(DenseParams p) {
  constexpr int PITCH = BK + 4;                     // +4 floats: conflict-free ds_read_b128 fragments (BK = 32, 64)
  constexpr int WM = BM / 2, WN = BN / 2;           // 2x2 waves
  constexpr int TM = WM / 32, TN = WN / 32;         // 32x32 MFMA tiles per wave
  constexpr int KQ = BK / 4;                        // f32x4_t per tile row
  constexpr int A_F4 = BM * KQ / 256, B_F4 = BN * KQ / 256;
  extern __shared__ __attribute__((aligned(16))) float dk_sm[];
  float (*As)[BM * PITCH] = reinterpret_cast<float (*)[BM * PITCH]>(dk_sm);
  float (*Bs)[BN * PITCH] = reinterpret_cast<float (*)[BN * PITCH]>(dk_sm + 2 * BM * PITCH);
  float* Xh = dk_sm + 2 * (BM + BN) * PITCH;

  // XCD-aware order: blocks b and b+8 share an XCD (round-robin dispatch), so give each
  // XCD a contiguous run of tiles (m fastest): its L2 then holds 1/8 of W and all of X.
  const int ntiles = p.grid_m * p.grid_n;
  int t = blockIdx.x;
  if ((ntiles & 7) == 0) t = (t & 7) * (ntiles >> 3) + (t >> 3);
  const int m0 = (t % p.grid_m) * BM, n0 = (t / p.grid_m) * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;

  if (FUSE_L0) {
    for (int idx = tid; idx < BM * p.n_labels; idx += 256) {
      const int r = idx / p.n_labels, d = idx - r * p.n_labels, row = m0 + r;
      float v = 0.f;
      if (row < p.B) {
        const double x = p.theta[(size_t)row * p.ld_theta + (d < 4 ? d : 6)];   // label 4 = Vmic (col 6)
        v = (float)((x - p.xmin[d]) / p.xden[d] - 0.5);
      }
      Xh[r * PAYNE_MAX_LABELS + d] = v;
    }
    __syncthreads();
  }

  // Guarded loads (`if (ok) v = *p`) compile to a branch plus a wait per load and serialise
  // the tile fetch; load unconditionally from a clamped (always valid) address and apply the
  // mask when the value is written to LDS.
  // two register stages: a tile is requested TWO k-steps before it is needed (one step is ~2000
  // MFMA cycles, about one L2/Infinity-Cache round trip: with a one-step lead every step waited
  // for its loads -- measured 2900 cycles per step against 2048 of matrix work)
  f32x4_t ra0[A_F4], rb0[B_F4], ra1[A_F4], rb1[B_F4];
  auto load_tiles = [&](f32x4_t (&ra)[A_F4], f32x4_t (&rb)[B_F4], int k0) {
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const int idx = tid + i * 256, r = idx / KQ, k = k0 + (idx % KQ) * 4, row = m0 + r;
      if (FUSE_L0) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kj = (k + j < p.K0) ? k + j : p.K0 - 1;
          float z = p.b0[kj];
          for (int d = 0; d < p.n_labels; ++d) z = fmaf(p.W0[kj * p.n_labels + d], Xh[r * PAYNE_MAX_LABELS + d], z);
          o[j] = (k + j < p.K0) ? act_apply(z, p.act0) : 0.f;
        }
        ra[i] = (f32x4_t){o[0], o[1], o[2], o[3]};
      } else {
        const int rc = row < p.B ? row : p.B - 1, kc = k < p.K ? k : p.K - 4;
        ra[i] = *reinterpret_cast<const f32x4_t*>(p.X + (size_t)rc * p.ldx + kc);
      }
    }
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      const int idx = tid + i * 256, r = idx / KQ, k = k0 + (idx % KQ) * 4, col = n0 + r;
      const int cc = col < p.N ? col : p.N - 1, kc = k < p.K ? k : p.K - 4;
      rb[i] = *reinterpret_cast<const f32x4_t*>(p.W + (size_t)cc * p.K + kc);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto store_tiles = [&](const f32x4_t (&ra)[A_F4], const f32x4_t (&rb)[B_F4], int buf, int k0) {
    const f32x4_t z4 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const int idx = tid + i * 256, r = idx / KQ, k = k0 + (idx % KQ) * 4;
      const bool ok = FUSE_L0 || ((m0 + r) < p.B && k < p.K);
      *reinterpret_cast<f32x4_t*>(&As[buf][r * PITCH + (idx % KQ) * 4]) = ok ? ra[i] : z4;
    }
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      const int idx = tid + i * 256, r = idx / KQ, k = k0 + (idx % KQ) * 4;
      const bool ok = (n0 + r) < p.N && k < p.K;
      *reinterpret_cast<f32x4_t*>(&Bs[buf][r * PITCH + (idx % KQ) * 4]) = ok ? rb[i] : z4;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (p.K + BK - 1) / BK;
  HK_STAMP(0);
  auto compute = [&](int buf) {
    // A lane (row = lane&31, half = lane>>5) reads 4 consecutive k; MFMA step s then
    // contracts k = {8kk + s, 8kk + 4 + s} -- the same k set on both operands.
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      f32x4_t a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = *reinterpret_cast<const f32x4_t*>(&As[buf][(wm0 + i * 32 + (lane & 31)) * PITCH + kk * 8 + 4 * (lane >> 5)]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = *reinterpret_cast<const f32x4_t*>(&Bs[buf][(wn0 + j * 32 + (lane & 31)) * PITCH + kk * 8 + 4 * (lane >> 5)]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
  };
  load_tiles(ra0, rb0, 0);
  store_tiles(ra0, rb0, 0, 0);
  load_tiles(ra0, rb0, BK);                       // tile 1 (addresses are clamped: over-asking is harmless)
  load_tiles(ra1, rb1, 2 * BK);                   // tile 2
  lds_barrier();
  HK_STAMP(1);
  for (int it = 0; it < nk; it += 2) {
    compute(0);                                   // tile it
    if (it + 1 < nk) store_tiles(ra0, rb0, 1, (it + 1) * BK);
    if (it + 3 < nk) load_tiles(ra0, rb0, (it + 3) * BK);
    lds_barrier();
    if (it < 12) HK_STAMP(2 + it);
    if (it + 1 >= nk) break;
    compute(1);                                   // tile it + 1
    if (it + 2 < nk) store_tiles(ra1, rb1, 0, (it + 2) * BK);
    if (it + 4 < nk) load_tiles(ra1, rb1, (it + 4) * BK);
    lds_barrier();
    if (it + 1 < 12) HK_STAMP(3 + it);
  }

  // C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn0 + j * 32 + (lane & 31);
    if (col >= p.N) continue;
    const float bv = p.bias[col] - p.bias_shift;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < p.B) p.Y[(size_t)row * p.ldy + col] = act_apply(acc[i][j][r] + bv, p.act);
      }
  }
  HK_STAMP(15);
}


// ----------------------------------------------------------------------------
// Output layer, LDS-DMA form: the same 64x64x32 tiling and MFMA schedule as payne_dense_kernel, but
// the operand tiles go from global memory straight into a 3-stage LDS ring with
// global_load_lds_dwordx4 (no VGPR staging, no address-clamp/select VALU work, no LDS store
// instructions), requested TWO k-steps ahead and waited for with explicit vmcnt counts.  The
// register-staged kernel cannot keep loads in flight across its barriers (measured: a k-step
// that issues loads takes 2750 cycles, one that does not 1550).
//   * A lane's 16 bytes land at (wave-uniform base) + 16*lane, so padding rows is impossible; bank
//     conflicts of the fragment reads are avoided by an XOR swizzle instead: 16-byte chunk c of
//     tile row r sits at chunk c ^ ((r >> 1) & 7) -- the lane simply FETCHES the chunk that belongs
//     in its slot.
//   * nothing can be masked on the way, so both operands must be zero-padded in k to a multiple
//     of 32 in memory (X: the hidden buffers' pitch; W: ctx->w_out_pad) and rows are clamped.
// ----------------------------------------------------------------------------
constexpr int DM_NS = 3;                                   // ring stages
// WN = wave columns: tile = 64 x (32 WN), 2 WN waves.  WN = 2 is the 64 x 64 / 256-thread form (two workgroups
// per CU); WN = 4 the 64 x 128 / 512-thread form (one per CU, same waves per SIMD): the activation tile is then
// fetched once per 128 columns, 24 KB instead of 2 x 16 KB per k-step and CU -- the kernel is bound by the CU's
// miss throughput, not by the matrix pipes.
// BK = k-depth of a stage: 32 (rows of 128 B, 8 chunks, swizzle by (r >> 1) & 7) or 64 (rows of 256 B = one full
// bank cycle, 16 chunks, swizzle by r & 15): half as many barrier steps for the same bytes.
template <int WN, int BK> constexpr int dm_stage_floats() { return (64 + 32 * WN) * BK; }
template <int WN, int BK> constexpr size_t dm_lds_bytes() { return (size_t)DM_NS * dm_stage_floats<WN, BK>() * sizeof(float); }

template <int WN, int BK>
__global__ void __launch_bounds__(128 * WN) payne_dense_dma_kernel(DenseParams p) {
  constexpr int BN = 32 * WN, NW = 2 * WN;                 // tile columns, waves
  constexpr int STAGE = dm_stage_floats<WN, BK>();
  constexpr int CH = BK / 4, RP = 256 / BK;                // 16-byte chunks per row, rows per 1-KiB piece
  constexpr int NBLK = (64 + BN) / RP, NA = 64 / RP;       // pieces per stage, of which A
  constexpr int PER = NBLK / NW;                           // pieces per wave
  static_assert(NBLK % NW == 0 && (BK == 32 || BK == 64), "pieces divide over the waves");
  extern __shared__ __attribute__((aligned(16))) float dm_sm[];
  const int ntiles = p.grid_m * p.grid_n;
  int t = blockIdx.x;
  if ((ntiles & 7) == 0) t = (t & 7) * (ntiles >> 3) + (t >> 3);      // XCD-aware order (see payne_dense_kernel)
  const int m0 = (t % p.grid_m) * 64, n0 = (t / p.grid_m) * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave / WN) * 32, wn0 = (wave % WN) * 32;
  auto swz = [](int row) { return BK == 32 ? ((row >> 1) & 7) : (row & 15); };

  // the 1-KiB pieces this wave moves per stage: piece q covers RP rows of A (q < NA) or of B
  const float* src[PER];
  int dst[PER];                                            // float offset inside a stage (wave-uniform)
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int q = wave * PER + j;
    const bool isA = q < NA;
    const int blk = isA ? q : q - NA, row = RP * blk + lane / CH;
    const int c = (lane % CH) ^ swz(row);                  // which chunk of the row belongs in this lane's slot
    if (isA) {
      const int r = (m0 + row < p.B) ? m0 + row : p.B - 1;
      src[j] = p.X + (size_t)r * p.ldx + 4 * c;
    } else {
      const int r = (n0 + row < p.N) ? n0 + row : p.N - 1;
      src[j] = p.W + (size_t)r * p.K + 4 * c;              // p.K: padded pitch of the weight copy
    }
    dst[j] = (isA ? 0 : 64 * BK) + blk * 256;
  }
  auto issue = [&](int stage, int k0) {
#pragma unroll
    for (int j = 0; j < PER; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + k0),
                                       (__attribute__((address_space(3))) void*)(dm_sm + stage * STAGE + dst[j]), 16, 0, 0);
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // fragment addresses (floats inside a stage): row R, chunk 2kk + half, swizzled
  const int Ra = wm0 + (lane & 31), Rb = wn0 + (lane & 31), half = lane >> 5;
  const int sa = swz(Ra), sb = swz(Rb);

  const int nk = p.K / BK;                                 // padded: exact
  HK_STAMP(0);
  issue(0, 0);
  if (nk > 1) issue(1, BK);
  for (int it = 0; it < nk; ++it) {
    // my pieces of stage `it` have landed once at most the PER younger loads (stage it+1) are outstanding
    if (it + 1 < nk) {
      if (PER == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (PER == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (PER == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_assert(PER == 3 || PER == 4 || PER == 6 || PER == 8, "vmcnt literal");
    asm volatile("s_barrier" ::: "memory");                // everybody's pieces landed; everybody finished step it-1
    if (it < 13) HK_STAMP(1 + it);
    if (it + 2 < nk) issue((it + 2) % DM_NS, (it + 2) * BK);    // into the buffer step it-1 just released
    const float* Asb = dm_sm + (it % DM_NS) * STAGE;
    const float* Bsb = Asb + 64 * BK;
    f32x4_t a[BK / 8], b[BK / 8];                           // all fragments first (one LDS round trip per step)
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      const int c = 2 * kk + half;
      a[kk] = *reinterpret_cast<const f32x4_t*>(Asb + Ra * BK + 4 * (c ^ sa));
      b[kk] = *reinterpret_cast<const f32x4_t*>(Bsb + Rb * BK + 4 * (c ^ sb));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].x, b[kk].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].y, b[kk].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].z, b[kk].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].w, b[kk].w, acc, 0, 0, 0);
    }
  }
  // C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int col = n0 + wn0 + (lane & 31);
  if (col < p.N) {
    const float bv = p.bias[col] - p.bias_shift;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < p.B) __builtin_nontemporal_store(act_apply(acc[r] + bv, p.act), &p.Y[(size_t)row * p.ldy + col]);   // streamed: next read by other XCDs
    }
  }
  HK_STAMP(15);
}

// ----------------------------------------------------------------------------
// Hidden layers are tiny GEMMs ([B x H] x [H x H], ~0.1 GFLOP): one wave per 16x16 output
// tile (hundreds of independent waves) with v_mfma_f32_16x16x4_f32, fragments read straight
// from L2 as f32x4_t (lane (r, g) holds 4 consecutive k of row r at offset 4g; MFMA step t
// contracts k = {4g + t}, identically on both operands), two accumulators to cover the
// 40-cycle dependent-issue latency.  No LDS, no barriers: latency ~ K/4 MFMAs.
// ----------------------------------------------------------------------------

template <bool FUSE_L0>
__global__ void __launch_bounds__(64) payne_dense_small_kernel(DenseParams p) {
  const int tm = blockIdx.x / p.grid_n, tn = blockIdx.x - tm * p.grid_n;
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  const int row = tm * 16 + r, col = tn * 16 + r;
  const bool rowok = row < p.B, colok = col < p.N;
  float xh[4] = {0.f, 0.f, 0.f, 0.f};
  const bool fast0 = FUSE_L0 && (p.n_labels == 4);            // 4-label nets: W0 rows are f32x4_t
  if (FUSE_L0 && rowok) {
#pragma unroll
    for (int d = 0; d < 4; ++d)
      if (d < p.n_labels) xh[d] = (float)((p.theta[(size_t)row * p.ld_theta + d] - p.xmin[d]) / p.xden[d] - 0.5);
  }
  float xh4 = 0.f;                                            // 5th label (vmic, theta column 6)
  if (FUSE_L0 && rowok && p.n_labels == 5)
    xh4 = (float)((p.theta[(size_t)row * p.ld_theta + 6] - p.xmin[4]) / p.xden[4] - 0.5);
  f32x4_t acc[2];
  acc[0] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  acc[1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const float* wrow = p.W + (size_t)(colok ? col : 0) * p.K;
  const float* xrow = FUSE_L0 ? nullptr : p.X + (size_t)(rowok ? row : 0) * p.ldx;
  // K is walked 64 at a time: the 4 steps' operand loads (B fragment, and W0/b0 rows or the A
  // fragment) are all issued before the first MFMA, so one L2 latency is paid per 64 k, not per 16.
  constexpr int SU = 4;
  const f32x4_t z4 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < p.K; k0 += 16 * SU) {
    f32x4_t b[SU], a[SU], bb[SU], w0[SU][4];
#pragma unroll
    for (int s = 0; s < SU; ++s) {
      const int k = k0 + s * 16 + 4 * g;
      b[s] = (colok && k < p.K) ? *reinterpret_cast<const f32x4_t*>(wrow + k) : z4;
      if (FUSE_L0) {
        if (fast0 && rowok && k + 3 < p.K0) {
          bb[s] = *reinterpret_cast<const f32x4_t*>(p.b0 + k);
          const f32x4_t* wp = reinterpret_cast<const f32x4_t*>(p.W0 + (size_t)k * 4);
          w0[s][0] = wp[0]; w0[s][1] = wp[1]; w0[s][2] = wp[2]; w0[s][3] = wp[3];
        }
      } else {
        a[s] = (rowok && k < p.K) ? *reinterpret_cast<const f32x4_t*>(xrow + k) : z4;
      }
    }
#pragma unroll
    for (int s = 0; s < SU; ++s) {
      const int k = k0 + s * 16 + 4 * g;
      if (FUSE_L0) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (fast0 && rowok && k + 3 < p.K0) {
          const float bq[4] = {bb[s].x, bb[s].y, bb[s].z, bb[s].w};
#pragma unroll
          for (int j = 0; j < 4; ++j)
            o[j] = act_apply(fmaf(w0[s][j].w, xh[3], fmaf(w0[s][j].z, xh[2], fmaf(w0[s][j].y, xh[1], fmaf(w0[s][j].x, xh[0], bq[j])))), p.act0);
        } else if (rowok) {                                   // 5-label nets / ragged tail
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (k + j < p.K0) {
              float z = p.b0[k + j];
              const float* wq = p.W0 + (size_t)(k + j) * p.n_labels;
              for (int d = 0; d < p.n_labels && d < 4; ++d) z = fmaf(wq[d], xh[d], z);
              if (p.n_labels == 5) z = fmaf(wq[4], xh4, z);
              o[j] = act_apply(z, p.act0);
            }
          }
        }
        a[s] = (f32x4_t){o[0], o[1], o[2], o[3]};
      }
      f32x4_t& c = acc[s & 1];
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].x, b[s].x, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].y, b[s].y, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].z, b[s].z, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].w, b[s].w, c, 0, 0, 0);
    }
  }
  // C/D map of the 16x16 tile: col = lane&15, row = 4*(lane>>4) + reg
  if (colok) {
    const float bv = p.bias[col] - p.bias_shift;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int orow = tm * 16 + 4 * g + q;
      if (orow < p.B) p.Y[(size_t)orow * p.ldy + col] = act_apply(acc[0][q] + acc[1][q] + bv, p.act);
    }
  }
}


// ----------------------------------------------------------------------------
// Hidden layers, workgroup form: one 256-thread group per 32x32 output tile, the whole K
// extent (<= 320 per chunk) of both operands staged in LDS by coalesced f32x4_t loads issued
// together (one L2 latency), then the four waves split K between them (v_mfma_f32_16x16x4_f32,
// 2x2 tiles each) and their partial tiles are summed through LDS.  With FUSE_L0 the A tile is
// produced in place from theta (label encoding + first layer + activation).
// ----------------------------------------------------------------------------
constexpr int HK_KC = 320;          // K chunk
constexpr int HK_PITCH = 328;       // 8*odd floats: conflict-free ds_read_b128 for the 16-row x 4-offset lane map
constexpr size_t HK_LDS_BYTES = (size_t)(2 * 32 * HK_PITCH + 32 * PAYNE_MAX_LABELS) * sizeof(float);

// Workgroups past the GEMM tiles (first-layer launch only) compute the per-candidate records of
// the post kernel (prep_candidate: Doppler / rotation / instrument scalars, mask counts, R-stage
// window), one thread per candidate, on compute units the 160 GEMM tiles leave idle.
struct PrepArgs {
  PostTables T;
  CandState* out;            // [B] (null: no records from this launch)
  double instr_factor;
  int n_gemm;                // workgroups that are GEMM tiles
};

// NL: label slots the fused first layer loops over (4 for the usual Teff/logg/FeH/aFe nets, else PAYNE_MAX_LABELS)
template <bool FUSE_L0, int NL>
__global__ void __launch_bounds__(256, 1) payne_dense_hidden_kernel(DenseParams p, const PrepArgs pa) {
  if ((int)blockIdx.x >= pa.n_gemm) {
    if (FUSE_L0 && pa.out) {
      const int cand = ((int)blockIdx.x - pa.n_gemm) * 256 + (int)threadIdx.x;
      if (cand < p.B) prep_candidate(pa.T, p.theta + (size_t)cand * p.ld_theta, pa.instr_factor, pa.out[cand]);
    }
    return;
  }
  HK_STAMP(0);
  extern __shared__ __attribute__((aligned(16))) float hk_sm[];
  float* As = hk_sm;
  float* Bs = As + 32 * HK_PITCH;
  float* Xh = Bs + 32 * HK_PITCH;
  const int tm = blockIdx.x / p.grid_n, tn = blockIdx.x - tm * p.grid_n;
  const int m0 = tm * 32, n0 = tn * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const f32x4_t z4 = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // one K chunk; the usual single-chunk case (K <= 320) is called outside any loop: around a loop the
  // compiler's wait-count bookkeeping turns conservative (a vmcnt(0) right after the first load)
  auto chunk = [&](const int kc) {
    const int kn = (p.K - kc < HK_KC) ? (p.K - kc) : HK_KC;        // multiple of 4
    const int kn16 = (kn + 15) & ~15;
    const int nk4 = kn16 >> 2;
    // ---- stage B (weights) and A (activations or the fused first layer) -------------------
    // every global load of the chunk is issued before the first LDS store (a load->store loop
    // would pay one L2 latency per iteration)
    constexpr int NKI = (HK_KC / 4 + 31) / 32;                   // k4 slots per thread: 3
    f32x4_t vb[4 * NKI], va[FUSE_L0 ? 1 : 4 * NKI];
    // Order of issue = order of arrival (vmcnt counts in order): the few small loads the fused first layer
    // needs (theta, W0, b0) go first, the 12 weight-tile loads after them, so that the first layer is
    // computed WHILE the weight tile is still on its way (it used to wait ~2000 cycles for it first).
    // Columns of the first layer: thread t owns column t (all 32 rows); the columns past 256 (48 of them at
    // H = 300) are spread over all threads -- column 256 + t % nE, rows t / nE, + G, + 2G, .. -- instead of
    // giving 48 threads of wave 0 a second full column each (that wave was the critical path).
    double xlab = 0.0;
    const int xrr = tid / PAYNE_MAX_LABELS, xd = tid - xrr * PAYNE_MAX_LABELS;
    const bool xlive = FUSE_L0 && kc == 0 && (xrr < 32) && (m0 + xrr < p.B) && (xd < p.n_labels);
    if (FUSE_L0 && kc == 0) {
      static_assert(32 * PAYNE_MAX_LABELS <= 256, "one (row, label) pair per thread");
      const int row = (m0 + xrr < p.B) ? m0 + xrr : p.B - 1;
      xlab = p.theta[(size_t)row * p.ld_theta + (xd < 4 ? xd : 6)];
    }
    const int nE = kn16 > 256 ? kn16 - 256 : 0;                  // extra columns
    const int G = nE ? 256 / nE : 1, eg = nE ? tid / nE : 0, ec = nE ? 256 + (tid - eg * nE) : 0;
    const bool eact = nE && eg < G;
    float w0[2][NL], bz[2];
    if (FUSE_L0) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = kc + (h ? ec : tid);
        const int kq = k < p.K0 ? k : p.K0 - 1;
        bz[h] = p.b0[kq];
#pragma unroll
        for (int d = 0; d < NL; ++d) w0[h][d] = p.W0[(size_t)kq * p.n_labels + (d < p.n_labels ? d : 0)];
#pragma unroll
        for (int d = 0; d < NL; ++d) w0[h][d] = (d < p.n_labels) ? w0[h][d] : 0.f;
      }
    }
#pragma unroll
    for (int it = 0; it < 4 * NKI; ++it) {   // unconditional loads from clamped addresses (see payne_dense_kernel)
      const int rr = (it / NKI) * 8 + (tid >> 5), k4 = (tid & 31) + 32 * (it % NKI);
      const int kcl = (4 * k4 < kn) ? kc + 4 * k4 : kc + kn - 4;
      const int nr = (n0 + rr < p.N) ? n0 + rr : p.N - 1;
      vb[it] = *reinterpret_cast<const f32x4_t*>(p.W + (size_t)nr * p.K + kcl);
      if (!FUSE_L0) {
        const int mr = (m0 + rr < p.B) ? m0 + rr : p.B - 1;
        va[it] = *reinterpret_cast<const f32x4_t*>(p.X + (size_t)mr * p.ldx + kcl);
      }
    }
    HK_STAMP(1);
    __builtin_amdgcn_sched_barrier(0);        // keep every load ahead of the first LDS store
    if (FUSE_L0) {
      if (kc == 0) {
        // (static indices + selects: p.xmin[dd] with a per-lane dd is a vector load from the kernarg
        //  segment, queued BEHIND the weight-tile loads -- waiting for it would wait for them)
        double xm = p.xmin[0], xdn = p.xden[0];
#pragma unroll
        for (int d = 1; d < PAYNE_MAX_LABELS; ++d) { xm = (xd == d) ? p.xmin[d] : xm; xdn = (xd == d) ? p.xden[d] : xdn; }
        if (tid < 32 * PAYNE_MAX_LABELS) Xh[tid] = xlive ? (float)((xlab - xm) / xdn - 0.5) : 0.f;
        lds_barrier();                        // LDS only: the weight-tile loads stay in flight
      }
      HK_STAMP(2);
      // the encoded labels into registers first: As and Xh are the same LDS array to the compiler, so a read
      // of Xh cannot move above a store to As, and a loop that alternates them pays one LDS round trip per
      // row (measured: 14 600 of the kernel's 25 000 cycles)
      float xr[32][NL], xe[8][NL];
#pragma unroll
      for (int rr = 0; rr < 32; ++rr)
#pragma unroll
        for (int d = 0; d < NL; ++d) xr[rr][d] = Xh[rr * PAYNE_MAX_LABELS + d];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int rr = eg + G * j, rc = rr < 32 ? rr : 31;
#pragma unroll
        for (int d = 0; d < NL; ++d) xe[j][d] = Xh[rc * PAYNE_MAX_LABELS + d];
      }
      const bool lre = p.act0 == PAYNE_ACT_LRELU, plain = !lre && p.act0 != PAYNE_ACT_SIGMOID;
      {
        const bool live = (kc + tid) < p.K0;
        float zz[32];
#pragma unroll
        for (int rr = 0; rr < 32; ++rr) {
          float z = bz[0];
#pragma unroll
          for (int d = 0; d < NL; ++d) z = fmaf(w0[0][d], xr[rr][d], z);
          zz[rr] = z;
        }
        if (lre) {
#pragma unroll
          for (int rr = 0; rr < 32; ++rr) zz[rr] = lrelu01(zz[rr]);
        } else if (!plain) {
#pragma unroll
          for (int rr = 0; rr < 32; ++rr) zz[rr] = 1.0f / (1.0f + expf(-zz[rr]));
        }
        if (tid < kn16) {
#pragma unroll
          for (int rr = 0; rr < 32; ++rr) As[rr * HK_PITCH + tid] = live ? zz[rr] : 0.f;
        }
      }
      if (nE) {                               // (G * 8 >= 32 for every nE <= 64)
        const bool live = (kc + ec) < p.K0;
        float ze[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float z = bz[1];
#pragma unroll
          for (int d = 0; d < NL; ++d) z = fmaf(w0[1][d], xe[j][d], z);
          ze[j] = lre ? lrelu01(z) : (plain ? z : 1.0f / (1.0f + expf(-z)));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int rr = eg + G * j;
          if (eact && rr < 32) As[rr * HK_PITCH + ec] = live ? ze[j] : 0.f;
        }
      }
    } else {
      HK_STAMP(2);
    }
    HK_STAMP(6);                                          // (first layer done; the weight tile is stored next)
#pragma unroll
    for (int it = 0; it < 4 * NKI; ++it) {
      const int rr = (it / NKI) * 8 + (tid >> 5), k4 = (tid & 31) + 32 * (it % NKI);
      if (k4 < nk4) {
        const bool kok = 4 * k4 < kn;
        *reinterpret_cast<f32x4_t*>(&Bs[rr * HK_PITCH + 4 * k4]) = (kok && n0 + rr < p.N) ? vb[it] : z4;
        if (!FUSE_L0) *reinterpret_cast<f32x4_t*>(&As[rr * HK_PITCH + 4 * k4]) = (kok && m0 + rr < p.B) ? va[it] : z4;
      }
    }
    __syncthreads();
    HK_STAMP(3);
    // ---- the four waves split the K steps of this chunk -------------------------------------
    const int steps = kn16 >> 4;
    for (int s = wave; s < steps; s += 4) {
      const int k = s * 16 + 4 * g;
      f32x4_t a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const f32x4_t*>(&As[(16 * i + r) * HK_PITCH + k]);
        b[i] = *reinterpret_cast<const f32x4_t*>(&Bs[(16 * i + r) * HK_PITCH + k]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  };
  if (p.K <= HK_KC) chunk(0);
  else for (int kc = 0; kc < p.K; kc += HK_KC) chunk(kc);
  HK_STAMP(4);
  // ---- sum the four partial tiles (C/D map: col = lane&15, row = 4*(lane>>4) + reg) ------------
  float* Red = As;                                               // [4][32][33]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) Red[(wave * 32 + 16 * i + 4 * g + q) * 33 + 16 * j + r] = acc[i][j][q];
  __syncthreads();
  for (int idx = tid; idx < 32 * 32; idx += 256) {
    const int rr = idx >> 5, cc = idx & 31, row = m0 + rr, col = n0 + cc;
    if (row < p.B && col < p.N) {
      const float v = Red[rr * 33 + cc] + Red[(32 + rr) * 33 + cc] + Red[(64 + rr) * 33 + cc] + Red[(96 + rr) * 33 + cc];
      const float y = act_apply(v + (p.bias[col] - p.bias_shift), p.act);
      p.Y[(size_t)row * p.ldy + col] = y;
      if (p.Yp) {                                           // the same value as three bf16 planes
        const __bf16 b1 = (__bf16)y;
        const float r1 = y - (float)b1;
        const __bf16 b2 = (__bf16)r1;
        const __bf16 b3 = (__bf16)(r1 - (float)b2);
        const size_t o = (size_t)row * p.ldyp + col;
        p.Yp[o] = __builtin_bit_cast(unsigned short, b1);
        p.Yp[p.yp_plane + o] = __builtin_bit_cast(unsigned short, b2);
        p.Yp[2 * p.yp_plane + o] = __builtin_bit_cast(unsigned short, b3);
      }
    }
  }
  HK_STAMP(5);
}


// ----------------------------------------------------------------------------
// Output layer, K-resident form (K <= 312, i.e. hidden width <= 312): a workgroup keeps its
// 64-candidate activation tile (whole K) in LDS and walks a run of 32-pixel weight tiles
// through a register-staged double buffer, so the only exposed global latency is the first
// tile's; every later tile's loads fly under the previous tile's MFMAs (the streaming-K kernel
// above re-pays the load latency every 32 k).  Measured at C2: 26.8 us against 22.4 us for the
// streaming kernel at 2 workgroups per CU -- both are bound by fp32-MFMA issue at the clock the
// chip holds under matrix load, not by staging -- so this form is kept as an option
// (PAYNE_OUT_TILE=6), not the default.
// Wave w owns rows 16w..16w+15 of the tile and both 16-column halves (v_mfma_f32_16x16x4_f32,
// two accumulators).  LDS: 64 x 312 + 2 x 32 x 312 floats = 156 KiB -> one workgroup per CU,
// grid = (#64-row tiles) x (runs of pixel tiles) ~ one workgroup per CU.
// ----------------------------------------------------------------------------
constexpr int OK_PITCH = 312;                                   // 8*odd floats (conflict-free fragment reads)
constexpr int OK_KMAX = 304;                                    // padded K handled (19 steps of 16)
constexpr size_t OK_LDS_BYTES = (size_t)(64 + 2 * 32) * OK_PITCH * sizeof(float);

__global__ void __launch_bounds__(256, 1) payne_dense_out_kernel(DenseParams p, int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float ok_sm[];
  float* As = ok_sm;                                            // [64][OK_PITCH]
  float* Bs = ok_sm + 64 * OK_PITCH;                            // [2][32][OK_PITCH]
  const int ngroups = p.grid_n, total = p.grid_m * ngroups;
  int t = blockIdx.x;
  if ((total & 7) == 0) t = (t & 7) * (total >> 3) + (t >> 3);  // XCD-contiguous runs (m fastest)
  const int m0 = (t % p.grid_m) * 64;
  const int ntiles = (p.N + 31) >> 5;
  const int tile0 = (t / p.grid_m) * tiles_per_wg;
  const int tile1 = (tile0 + tiles_per_wg < ntiles) ? tile0 + tiles_per_wg : ntiles;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  const int K = p.K, K16 = (K + 15) & ~15, nk4 = K16 >> 2;     // K % 4 == 0
  const f32x4_t z4 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  constexpr int BI = (32 * (OK_KMAX / 4) + 255) / 256;          // f32x4_t per thread per B tile: 10

  // ---- A tile: 64 rows x K, all loads first ---------------------------------------------------
  {
    constexpr int AI = (64 * (OK_KMAX / 4) + 255) / 256;        // 19
    f32x4_t va[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int idx = tid + 256 * i, rr = idx / (OK_KMAX / 4), k4 = idx - rr * (OK_KMAX / 4);
      const int mr = (m0 + rr < p.B) ? m0 + rr : p.B - 1, kq = (4 * k4 < K) ? 4 * k4 : K - 4;   // clamped: no branch
      va[i] = *reinterpret_cast<const f32x4_t*>(p.X + (size_t)mr * p.ldx + kq);
    }
    __builtin_amdgcn_sched_barrier(0);        // keep every load ahead of the first LDS store
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int idx = tid + 256 * i, rr = idx / (OK_KMAX / 4), k4 = idx - rr * (OK_KMAX / 4);
      if (k4 < nk4) *reinterpret_cast<f32x4_t*>(&As[rr * OK_PITCH + 4 * k4]) = (m0 + rr < p.B && 4 * k4 < K) ? va[i] : z4;
    }
  }
  f32x4_t vb[BI];
  auto load_b = [&](int tile) {
    const int n0 = tile << 5;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int idx = tid + 256 * i, rr = idx / (OK_KMAX / 4), k4 = idx - rr * (OK_KMAX / 4);
      const int nr = (rr < 32 && n0 + rr < p.N) ? n0 + rr : p.N - 1, kq = (4 * k4 < K) ? 4 * k4 : K - 4;
      vb[i] = *reinterpret_cast<const f32x4_t*>(p.W + (size_t)nr * K + kq);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto store_b = [&](int buf, int tile) {
    float* B = Bs + buf * 32 * OK_PITCH;
    const int n0 = tile << 5;
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int idx = tid + 256 * i, rr = idx / (OK_KMAX / 4), k4 = idx - rr * (OK_KMAX / 4);
      if (rr < 32 && k4 < nk4) *reinterpret_cast<f32x4_t*>(&B[rr * OK_PITCH + 4 * k4]) = (n0 + rr < p.N && 4 * k4 < K) ? vb[i] : z4;
    }
  };
  if (tile0 < tile1) { load_b(tile0); store_b(0, tile0); }
  __syncthreads();

  const int steps = K16 >> 4;
  for (int tile = tile0; tile < tile1; ++tile) {
    const int buf = (tile - tile0) & 1;
    if (tile + 1 < tile1) load_b(tile + 1);                     // flies under this tile's MFMAs
    const float* B = Bs + buf * 32 * OK_PITCH;
    f32x4_t acc0 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    // software pipeline: the fragments of step s+1 are read from LDS while the 8 MFMAs of step s
    // issue (one wave per SIMD: nothing else would cover the ds_read latency)
    const float* Arow = &As[(16 * wave + r) * OK_PITCH + 4 * g];
    const float* B0row = &B[r * OK_PITCH + 4 * g];
    const float* B1row = &B[(16 + r) * OK_PITCH + 4 * g];
#define OK_READ(A_, B0_, B1_, S_)                                               \
    A_ = *reinterpret_cast<const f32x4_t*>(Arow + (S_) * 16);                     \
    B0_ = *reinterpret_cast<const f32x4_t*>(B0row + (S_) * 16);                   \
    B1_ = *reinterpret_cast<const f32x4_t*>(B1row + (S_) * 16);                   \
    __builtin_amdgcn_sched_barrier(0)      /* the reads are ISSUED here, ahead of the MFMAs below */
#define OK_MFMA8(A_, B0_, B1_)                                                    \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A_.x, B0_.x, acc0, 0, 0, 0);      \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A_.x, B1_.x, acc1, 0, 0, 0);      \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A_.y, B0_.y, acc0, 0, 0, 0);      \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A_.y, B1_.y, acc1, 0, 0, 0);      \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A_.z, B0_.z, acc0, 0, 0, 0);      \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A_.z, B1_.z, acc1, 0, 0, 0);      \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A_.w, B0_.w, acc0, 0, 0, 0);      \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A_.w, B1_.w, acc1, 0, 0, 0);      \
    __builtin_amdgcn_sched_barrier(0)
    f32x4_t a, b0, b1, an, b0n, b1n;
    OK_READ(a, b0, b1, 0);
    int s = 0;
    for (; s + 2 < steps; s += 2) {
      OK_READ(an, b0n, b1n, s + 1);
      OK_MFMA8(a, b0, b1);
      OK_READ(a, b0, b1, s + 2);
      OK_MFMA8(an, b0n, b1n);
    }
    if (s + 1 < steps) {                       // two steps left
      OK_READ(an, b0n, b1n, s + 1);
      OK_MFMA8(a, b0, b1);
      OK_MFMA8(an, b0n, b1n);
    } else {                                   // one step left
      OK_MFMA8(a, b0, b1);
    }
#undef OK_READ
#undef OK_MFMA8
    if (tile + 1 < tile1) store_b(buf ^ 1, tile + 1);           // before the output stores: vmcnt then only covers the loads
    // C/D map: col = lane&15, row = 4*(lane>>4) + reg
    const int n0 = tile << 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + 16 * j + r;
      if (col < p.N) {
        const float bv = p.bias[col] - p.bias_shift;
        const f32x4_t& a4 = j ? acc1 : acc0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = m0 + 16 * wave + 4 * g + q;
          if (row < p.B) p.Y[(size_t)row * p.ldy + col] = act_apply(a4[q] + bv, p.act);
        }
      }
    }
    __syncthreads();
  }
}


// ----------------------------------------------------------------------------
// Output layer on the bf16 matrix pipe at fp32 accuracy ("3 x bf16" split).
// Every fp32 operand is written exactly as x = x1 + x2 + x3 with bf16 parts (8 + 8 + 8
// mantissa bits: x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2); the subtractions are
// exact).  A product a*b is then the six partial products with i + j <= 4,
//   a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1,
// each exact in fp32 (8b x 8b), the dropped terms being < 2^-23 |ab|; accumulation is fp32 in
// the MFMA accumulator exactly as for the f32 MFMA.  v_mfma_f32_32x32x16_bf16 retires 16 k per
// 32 cycles against 2 k per 64 cycles for v_mfma_f32_32x32x2_f32, so six of them cost 3/8 of the
// fp32 issue time -- and the f32 MFMA kernel above is bound exactly by that issue time.
// Weights are split once at context creation ([3][Npad][Kp] bf16, zero padded); activations are
// split while their tile is staged into LDS.  Tile 64 x 128 x 32, 4 waves as 2 x 2, wave tile
// 32 x 64; LDS rows padded to 80 B (conflict-free ds_read_b128 for the 32-row x 2-half lane map).
// ----------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4_t __attribute__((ext_vector_type(4)));
constexpr int BX_BM = 64, BX_BN = 128, BX_BK = 32, BX_PITCH = 80;            // bytes per LDS row
constexpr size_t BX_LDS_BYTES = (size_t)2 * 3 * (BX_BM + BX_BN) * BX_PITCH;

__device__ __forceinline__ unsigned short bf16_bits(__bf16 v) { return __builtin_bit_cast(unsigned short, v); }
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  const __bf16 b1 = (__bf16)x;
  const float r1 = x - (float)b1;
  const __bf16 b2 = (__bf16)r1;
  const float r2 = r1 - (float)b2;
  const __bf16 b3 = (__bf16)r2;
  h = bf16_bits(b1); m = bf16_bits(b2); l = bf16_bits(b3);
}

struct Bx3Params {
  DenseParams d;
  const unsigned short* Wp;     // [3][Npad][Kp] bf16 planes of W
  int Kp, Npad;
  int dbg;                      // timing experiments (PAYNE_BX_DBG): 1 = no output stores, 2 = no MFMAs, 4 = no staging
};

__global__ void __launch_bounds__(256, 1) payne_dense_bf16x3_kernel(Bx3Params q) {
  const DenseParams& p = q.d;
  extern __shared__ __attribute__((aligned(16))) unsigned char bx_sm[];
  // [buf][plane][A rows 64 | B rows 128][80 B]
  auto lds_a = [&](int buf, int pl) { return bx_sm + ((size_t)(buf * 3 + pl) * (BX_BM + BX_BN)) * BX_PITCH; };
  auto lds_b = [&](int buf, int pl) { return lds_a(buf, pl) + (size_t)BX_BM * BX_PITCH; };
  const int ntiles = p.grid_m * p.grid_n;
  int t = blockIdx.x;
  if ((ntiles & 7) == 0) t = (t & 7) * (ntiles >> 3) + (t >> 3);             // XCD-contiguous tile runs (m fastest)
  const int m0 = (t % p.grid_m) * BX_BM, n0 = (t / p.grid_m) * BX_BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 64;
  const int r = lane & 31, h = lane >> 5;

  // staging registers: A = 2 float4 of fp32 per thread, B = 2 x 16 B per plane per thread
  f32x4_t ra[2];
  f32x4_t rb[3][2];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i, row = idx >> 3, k = k0 + (idx & 7) * 4;
      const int mr = (m0 + row < p.B) ? m0 + row : p.B - 1, kc = (k < p.K) ? k : p.K - 4;
      ra[i] = *reinterpret_cast<const f32x4_t*>(p.X + (size_t)mr * p.ldx + kc);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i, row = idx >> 2, c16 = idx & 3;
        rb[pl][i] = *reinterpret_cast<const f32x4_t*>(q.Wp + ((size_t)pl * q.Npad + n0 + row) * q.Kp + k0 + 8 * c16);
      }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto store_tiles = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i, row = idx >> 3, k4 = idx & 7, k = k0 + k4 * 4;
      const bool ok = (m0 + row < p.B) && (k < p.K);
      const float v[4] = {ok ? ra[i].x : 0.f, ok ? ra[i].y : 0.f, ok ? ra[i].z : 0.f, ok ? ra[i].w : 0.f};
      u16x4_t p1, p2, p3;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned short a1, a2, a3;
        split3(v[e], a1, a2, a3);
        p1[e] = a1; p2[e] = a2; p3[e] = a3;
      }
      *reinterpret_cast<u16x4_t*>(lds_a(buf, 0) + row * BX_PITCH + k4 * 8) = p1;
      *reinterpret_cast<u16x4_t*>(lds_a(buf, 1) + row * BX_PITCH + k4 * 8) = p2;
      *reinterpret_cast<u16x4_t*>(lds_a(buf, 2) + row * BX_PITCH + k4 * 8) = p3;
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i, row = idx >> 2, c16 = idx & 3;
        *reinterpret_cast<f32x4_t*>(lds_b(buf, pl) + row * BX_PITCH + c16 * 16) = rb[pl][i];
      }
  };

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  const int nk = (p.K + BX_BK - 1) / BX_BK;
  load_tiles(0);
  store_tiles(0, 0);
  __syncthreads();
  for (int it = 0; it < nk; ++it) {
    const int buf = it & 1;
    if (it + 1 < nk && !(q.dbg & 4)) load_tiles((it + 1) * BX_BK);
    if (!(q.dbg & 2))
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {                    // two 16-deep MFMA steps per 32-deep tile
      bf16x8_t a[3], b[2][3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        a[pl] = *reinterpret_cast<const bf16x8_t*>(lds_a(buf, pl) + (wm0 + r) * BX_PITCH + ks * 32 + h * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          b[j][pl] = *reinterpret_cast<const bf16x8_t*>(lds_b(buf, pl) + (wn0 + 32 * j + r) * BX_PITCH + ks * 32 + h * 16);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {                     // smallest partial products first
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[j][0], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[j][1], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[j][2], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[j][0], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[j][1], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[j][0], acc[j], 0, 0, 0);
      }
    }
    if (it + 1 < nk && !(q.dbg & 4)) store_tiles(buf ^ 1, (it + 1) * BX_BK);
    __syncthreads();
  }
  if (q.dbg & 1) {   // keep the accumulators alive, store one value per wave
    float sacc = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc += acc[j][e];
    if (sacc == 12345.678f) p.Y[0] = sacc;
    return;
  }
  // C/D map of the 32x32 tile: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn0 + 32 * j + r;
    if (col >= p.N) continue;
    const float bv = p.bias[col] - p.bias_shift;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + wm0 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (row < p.B) p.Y[(size_t)row * p.ldy + col] = act_apply(acc[j][e] + bv, p.act);
    }
  }
}

// ----------------------------------------------------------------------------
// Output layer, 3 x bf16 split on the LDS-DMA ring: the arithmetic of payne_dense_bf16x3_kernel (six bf16
// partial products per term, fp32-accurate) with the operand delivery of payne_dense_dma_kernel.  BOTH
// operands arrive pre-split -- the weights at context creation, the activations from the hidden-layer kernel's
// epilogue (DenseParams::Yp) -- so the kernel issues nothing but DMA requests, fragment reads and MFMAs:
// per 32-deep k-step and wave 12 x v_mfma_f32_32x32x16_bf16 = 384 matrix cycles against 1024 for fp32.
// (The register-staged bf16x3 kernel was only ~10 % faster than fp32 because its barriers drained the loads;
// with the ring the steady state is matrix-bound again, at 3/8 of the fp32 time.)
// Stage = 3 planes x (64 A rows + 64 B rows) x 64 B = 24 KB; a 1-KiB DMA piece = 16 rows of one plane;
// 16-byte chunk c of tile row r sits at chunk c ^ ((r >> 2) & 3) (conflict-free 16-lane fragment reads).
// ----------------------------------------------------------------------------
constexpr int BD_STAGE = 3 * (64 + 64) * 64;               // bytes per stage
constexpr size_t BD_LDS_BYTES = (size_t)DM_NS * BD_STAGE;
struct Bd3Params {
  DenseParams d;
  const unsigned short* Ap; int lda; size_t a_plane;       // activation planes [3][a_plane], row pitch lda (elements)
  const unsigned short* Wp; int Kp; int Npad;              // weight planes [3][Npad][Kp]
  int dbg;                                                 // timing experiments (PAYNE_BD_DBG): 1 no MFMAs, 2 no DMA after the prologue, 4 no fragment reads, 8 no output stores
};
__global__ void __launch_bounds__(256) payne_dense_bx3dma_kernel(Bd3Params q) {
  const DenseParams& p = q.d;
  extern __shared__ __attribute__((aligned(16))) unsigned char bd_sm[];
  const int ntiles = p.grid_m * p.grid_n;
  int t = blockIdx.x;
  if ((ntiles & 7) == 0) t = (t & 7) * (ntiles >> 3) + (t >> 3);      // XCD-aware order (see payne_dense_kernel)
  const int m0 = (t % p.grid_m) * 64, n0 = (t / p.grid_m) * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
  // 24 pieces per stage: piece q = (operand, plane, 16-row block); 6 per wave
  const unsigned char* src[6];
  int dst[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int qq = wave * 6 + j, isB = qq / 12, pl = (qq % 12) / 4, blk = qq % 4;
    const int row = 16 * blk + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);           // which 16-byte chunk of the row belongs in this lane's slot
    if (!isB) {
      const int r = (m0 + row < p.B) ? m0 + row : p.B - 1;
      src[j] = reinterpret_cast<const unsigned char*>(q.Ap + (size_t)pl * q.a_plane + (size_t)r * q.lda) + 16 * c;
    } else {
      src[j] = reinterpret_cast<const unsigned char*>(q.Wp + ((size_t)pl * q.Npad + n0 + row) * q.Kp) + 16 * c;
    }
    dst[j] = (isB ? 3 * 64 * 64 : 0) + pl * 64 * 64 + blk * 1024;
  }
  auto issue = [&](int stage, int k0) {                    // k0 in elements: 2 bytes each
#pragma unroll
    for (int j = 0; j < 6; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + 2 * k0),
                                       (__attribute__((address_space(3))) void*)(bd_sm + stage * BD_STAGE + dst[j]), 16, 0, 0);
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int Ra = wm0 + (lane & 31), Rb = wn0 + (lane & 31), h = lane >> 5;
  const int sa = (Ra >> 2) & 3, sb = (Rb >> 2) & 3;
  const int nk = q.Kp / 32;
  issue(0, 0);
  if (nk > 1) issue(1, 32);
  for (int it = 0; it < nk; ++it) {
    if (q.dbg & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (it + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (it + 2 < nk && !(q.dbg & 2)) issue((it + 2) % DM_NS, (it + 2) * 32);
    const unsigned char* As = bd_sm + (it % DM_NS) * BD_STAGE;
    const unsigned char* Bs = As + 3 * 64 * 64;
    // all twelve fragments of the stage first, then the twelve MFMAs back to back: read -> wait -> MFMA per
    // fragment pair exposes one LDS round trip per pair (the compiler places reads next to their use)
    bf16x8_t a[2][3], b[2][3];
    if (q.dbg & 4) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) { a[ks][pl] = (bf16x8_t)(__bf16)(float)it; b[ks][pl] = (bf16x8_t)(__bf16)1.0f; }
    } else
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {                       // two 16-deep MFMA steps per 32-deep stage
      const int c = 2 * ks + h;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        a[ks][pl] = *reinterpret_cast<const bf16x8_t*>(As + pl * 4096 + Ra * 64 + 16 * (c ^ sa));
        b[ks][pl] = *reinterpret_cast<const bf16x8_t*>(Bs + pl * 4096 + Rb * 64 + 16 * (c ^ sb));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (q.dbg & 1) { acc[0] += (float)a[0][0][0] + (float)b[1][2][3]; continue; }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][2], b[ks][0], acc, 0, 0, 0);   // smallest partial products first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][1], b[ks][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][0], b[ks][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][1], b[ks][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][0], b[ks][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][0], b[ks][0], acc, 0, 0, 0);
    }
  }
  if (q.dbg & 8) { float sacc = 0.f; for (int r = 0; r < 16; ++r) sacc += acc[r]; if (sacc == 12345.678f) p.Y[0] = sacc; return; }
  const int col = n0 + wn0 + (lane & 31);
  if (col < p.N) {
    const float bv = p.bias[col] - p.bias_shift;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < p.B) p.Y[(size_t)row * p.ldy + col] = act_apply(acc[r] + bv, p.act);
    }
  }
}

// ============================================================================
// per-candidate spectrum pipeline
// ============================================================================
struct PostArgs {
  const double* theta; int ld_theta;
  double instr_factor;
  const float* raw; int ld_raw;
  float* out; int ld_out; int out_stage;
  double* lnl;                       // [B] or null
  const double* mags; int n_filters; // SED magnitudes of this batch (null: no photometry)
  const double* obs_mag; const double* obs_err;
  unsigned long long* stamps;        // diagnostic build: [B][64] cycle stamps (slot 0 = count)
  const CandState* prep;             // [B] per-candidate records made by the first dense launch (null: none)
};

// BUF_LDS: the two spectrum buffers are LDS (else a global workspace); TW_LDS: so is the twiddle table.
template <bool BUF_LDS, bool TW_LDS>
struct DevExecT {
#ifdef __HIP_DEVICE_COMPILE__
  static __device__ __forceinline__ auto buf(c32* p) {
    if constexpr (BUF_LDS) return (PAYNE_AS_LDS f2v*)p; else return (PAYNE_AS_GLOBAL f2v*)p;
  }
  static __device__ __forceinline__ auto twid(const c32* p) {
    if constexpr (TW_LDS) return (const PAYNE_AS_LDS f2v*)p; else return (const PAYNE_AS_GLOBAL f2v*)p;
  }
#else
  static c32* buf(c32* p) { return p; }
  static const c32* twid(const c32* p) { return p; }
#endif
#ifdef PAYNE_STAMPS
  // diagnostic build only (libpayne_hip_diag.so): cycle stamp after every phase barrier
  unsigned long long* stamps = nullptr;
  int nst = 0;
#endif
  template <class F>
  __device__ __forceinline__ void par(F&& f) {
    f((int)threadIdx.x, (int)blockDim.x);
    __syncthreads();
    mark(0);
  }
  template <class F>
  __device__ __forceinline__ void single(F&& f) { if (threadIdx.x == 0) f((int)blockDim.x); }
  // diagnostic build: an extra cycle stamp inside a phase, written by thread `who`
  __device__ __forceinline__ void mark(int who) {
#ifdef PAYNE_STAMPS
    if (stamps && nst < 62) { ++nst; if ((int)threadIdx.x == who) stamps[nst] = __builtin_amdgcn_s_memtime(); }
#else
    (void)who;
#endif
  }
  __device__ __forceinline__ int nthreads() const { return (int)blockDim.x; }
};

__device__ __forceinline__ double sed_chi2(const double* mags, const double* obs, const double* err, int F) {
  double s = 0.0;   // likelihood.py:109-112
  for (int f = 0; f < F; ++f) { const double d = mags[f] - obs[f]; s += (d * d) / (err[f] * err[f]); }
  return s;
}

// LEAN: the likelihood-only instantiation (out_stage == -1, no spectrum output, per-candidate records present):
// the output variants of the observed-grid loop, the stage branches and the in-kernel setup are compiled out
// (most of the 270 KB of the full kernel).
template <int LOG2N, bool TW_LDS, bool LEAN = false>
__global__ void __launch_bounds__(kPostThreads) payne_post_kernel(const PostTables T, PostArgs a) {
  // T by value: its pointer members then live in the kernarg segment and are known to be
  // global (a struct read through a device pointer yields generic pointers -> flat_load,
  // which also ties every table load to the LDS wait counter)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n1 = T.n1;
  float* bufA = reinterpret_cast<float*>(smem);
  float* bufB = bufA + fft_buf_floats(n1);                     // room for the padded FFT intermediates
  double* red = reinterpret_cast<double*>(bufB + fft_buf_floats(n1));          // scratch_doubles(256)
  CandState* S = reinterpret_cast<CandState*>(red + scratch_doubles(kPostThreads));
  const c32* twf = T.twf;
  if (TW_LDS) {   // the FFT's (pass-ordered) twiddles into LDS: the passes then never leave the CU
    c32* twl = reinterpret_cast<c32*>(reinterpret_cast<unsigned char*>(S) + ((sizeof(CandState) + 15) & ~(size_t)15));
    typedef float f2g __attribute__((ext_vector_type(2)));
    const f2g* __restrict__ g = reinterpret_cast<const f2g*>(T.twf);
    f2g* tl = reinterpret_cast<f2g*>(twl);
    if constexpr (LOG2N > 0) {
      // every load of the table in flight at once (a load -> store loop pays one L2 round trip per
      // iteration: twelve of them at 4096 points, ~3.5 us before the first phase could start)
      constexpr int NTW = plan_table_len((1 << LOG2N) / 2), PER = (NTW + kPostThreads - 1) / kPostThreads;
      f2g tmp[PER];
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int i0 = (int)threadIdx.x + q * kPostThreads;
        tmp[q] = g[i0 < NTW ? i0 : NTW - 1];
      }
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int i0 = (int)threadIdx.x + q * kPostThreads;
        if (i0 < NTW) tl[i0] = tmp[q];
      }
    } else {
      const int nt = T.twf_n;
      for (int i = threadIdx.x; i < nt; i += kPostThreads) tl[i] = g[i];
    }
    twf = twl;                                                 // made visible by the first phase barrier
  }
  const int b = blockIdx.x;
  DevExecT<true, TW_LDS> ex;
#ifdef PAYNE_STAMPS
  if (a.stamps) {
    ex.stamps = a.stamps + (size_t)b * 64;
    if (threadIdx.x == 0) { ex.stamps[1] = __builtin_amdgcn_s_memtime(); }
    ex.nst = 1;
  }
#endif
  double* chi2 = red + scratch_doubles(kPostThreads) - 1;
  const int ostage = LEAN ? -1 : a.out_stage;
  float* outp = LEAN ? nullptr : (a.out ? a.out + (size_t)b * a.ld_out : nullptr);
  const CandState* prep = a.prep ? a.prep + b : nullptr;
  if (LEAN) { prep = a.prep + b; __builtin_assume(prep != nullptr); }     // LEAN is launched only with records
  run_candidate<LOG2N, kPostThreads>(ex, T, twf, a.theta + (size_t)b * a.ld_theta, a.instr_factor,
                                     a.raw + (size_t)b * a.ld_raw, bufA, bufB, *S, red, outp, ostage, chi2, prep);
  if (threadIdx.x == 0 && a.lnl && ostage < 0) {
    double x2 = *chi2;
    if (a.mags) x2 += sed_chi2(a.mags + (size_t)b * a.n_filters, a.obs_mag, a.obs_err, a.n_filters);
    a.lnl[b] = -0.5 * x2;                                       // likelihood.py:117
  }
#ifdef PAYNE_STAMPS
  if (a.stamps && threadIdx.x == 0) ex.stamps[0] = (unsigned long long)ex.nst;
#endif
}


// Spectra that do not fit LDS (n1 > 16384, e.g. the 65k-pixel R~100k grid): the same phase code
// with the two spectrum buffers in a per-workgroup global workspace (L2 / Infinity-Cache resident
// while it is being worked on) and the runtime-geometry FFT.  Workgroups are persistent and walk
// the batch with stride gridDim.x, so the workspace is sized by the grid, not by the batch.  All
// waves of a workgroup share one CU's L1, so __syncthreads() orders the global accesses between
// phases exactly as it orders LDS.  This is the HBM/L2-bandwidth-bound regime of SURVEY.md 8(d).
constexpr int kBigThreads = 512;
__global__ void __launch_bounds__(kBigThreads) payne_post_big_kernel(const PostTables T, PostArgs a, float* ws, int B) {
  __shared__ double red[kBigThreads + kBigThreads / 2 + 2];
  __shared__ CandState S;
  float* bufA = ws + (size_t)blockIdx.x * 2 * T.n1;
  float* bufB = bufA + T.n1;
  DevExecT<false, false> ex;
  double* chi2 = red + scratch_doubles(kBigThreads) - 1;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    run_candidate<0, kBigThreads>(ex, T, T.tw, a.theta + (size_t)b * a.ld_theta, a.instr_factor,
                                  a.raw + (size_t)b * a.ld_raw, bufA, bufB, S, red,
                                  a.out ? a.out + (size_t)b * a.ld_out : nullptr, a.out_stage, chi2);
    if (threadIdx.x == 0 && a.lnl && a.out_stage < 0) {
      double x2 = *chi2;
      if (a.mags) x2 += sed_chi2(a.mags + (size_t)b * a.n_filters, a.obs_mag, a.obs_err, a.n_filters);
      a.lnl[b] = -0.5 * x2;
    }
    __syncthreads();
  }
}

// ============================================================================
// LSF-vector instrumental broadening (inst_R = dispersion in AA per observed pixel):
// ystpred.py:248-269 -> smoothspec(smoothtype='lsf') -> smooth_lsf_fft (smoothing.py:125-151, 482-586).
// Not on the sampler's usual path (Inst_R is a sampled scalar there); one workgroup per candidate,
// fp64 position arithmetic through a global workspace, written for clarity rather than speed:
//   mask (linear, +-2000 AA) -> sigma_i = interp(lambda_i (1+rv/c), obs, lsf) -> r = gradient(w)/sigma ->
//   cdf = cumsum(r)/max -> x_per_sigma = nanmedian(gradient(cdf)/r) -> nx = 2^ceil(log2(2/x_per_sigma)) ->
//   lam = interp(linspace(0,1,nx), cdf, w); s' = interp(lam, w, s) -> Gaussian FFT smoothing of width
//   x_per_sigma (dx = 1/nx) -> np.interp(obs, lam, .) (clamped: no NaN) -> blaze -> chi^2.
// Input: the spectrum after rotational broadening on the ANN grid (post kernel, stage 5), shifted by -1.
// ============================================================================
struct LsfArgs {
  const double* theta; int ld_theta;
  const float* spec; int ld_spec;        // [B][npix] after vsini, shifted
  const double* obs_wave; const double* lsf; // [nobs]
  double* ws; size_t ws_stride;          // per candidate: a[npix] | cdf[npix] | lam[n1]
  float* out; int ld_out; int out_stage; // 2 / 3 / -1
  double* lnl;
  const double* mags; int n_filters; const double* obs_mag; const double* obs_err;
};
// np.interp(x, xp, fp) (arr_interp): clamped outside, slope form inside
__device__ double interp_np(double x, const double* xp, const double* fp, int n) {
  if (x > xp[n - 1]) return fp[n - 1];
  if (x < xp[0]) return fp[0];
  int lo = 0, hi = n;                                   // last j with xp[j] <= x
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (xp[mid] <= x) lo = mid; else hi = mid; }
  const int j = lo;
  if (j == n - 1 || xp[j] == x) return fp[j];
  const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
  return slope * (x - xp[j]) + fp[j];
}
__global__ void __launch_bounds__(256) payne_lsf_kernel(const PostTables T, LsfArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];
  double* sortb = reinterpret_cast<double*>(lsm);                      // [n1] median sort buffer
  float* bufA = reinterpret_cast<float*>(sortb + T.n1);
  float* bufB = bufA + fft_buf_floats(T.n1);
  double* red = reinterpret_cast<double*>(bufB + fft_buf_floats(T.n1));    // [256 + 8]
  __shared__ int cnt_s[2], nvalid_s, nx_s;
  __shared__ double scal_s[4];                                         // 0 max(cdf), 1 x_per_sigma
  const int b = blockIdx.x, tid = threadIdx.x, npix = T.npix, nobs = T.nobs;
  const double* th = a.theta + (size_t)b * a.ld_theta;
  const float* sp = a.spec + (size_t)b * a.ld_spec;
  double* wsA = a.ws + (size_t)b * a.ws_stride;
  double* wsC = wsA + npix;
  double* wsL = wsC + npix;
  const double rv = th[4];
  const double op = (rv != 0.0) ? (1.0 + (rv / kCDoppler)) : 1.0;      // ystpred.py:228-232
  // ---- mask_wave(linear=True, width=100): strict limits obs.min - 2000, obs.max + 2000
  const double wl = T.obs_min + 20.0 * 100.0 * -1.0, wh = T.obs_max + 20.0 * 100.0 * 1.0;
  if (tid < 2) cnt_s[tid] = 0;
  __syncthreads();
  {
    int cb = 0, ca = 0;
    for (int i = tid; i < npix; i += 256) { const double c = T.lam[i] * op; cb += !(c > wl); ca += (c < wh); }
    atomicAdd(&cnt_s[0], cb); atomicAdd(&cnt_s[1], ca);
  }
  __syncthreads();
  const int i0 = cnt_s[0], n = cnt_s[1] - cnt_s[0];
  bool bad = n < 8;
  auto W = [&](int i) { return T.lam[i0 + i] * op; };
  if (!bad) {
    // ---- sigma_i (disparr = np.interp(modwave, outwave, inst_R)), r_i = gradient(w)_i / sigma_i
    for (int i = tid; i < n; i += 256) {
      const double sig = interp_np(W(i), a.obs_wave, a.lsf, nobs);
      const double dw = (i == 0) ? (W(1) - W(0)) : ((i == n - 1) ? (W(n - 1) - W(n - 2)) : (W(i + 1) - W(i - 1)) / 2.0);
      wsA[i] = dw / sig;
    }
    __syncthreads();
    // ---- cdf = cumsum(r): per-thread chunks, then a scan of the 256 chunk sums
    const int chunk = (n + 255) / 256, c0 = tid * chunk, c1 = (c0 + chunk < n) ? c0 + chunk : n;
    double acc = 0.0;
    for (int i = c0; i < c1; ++i) { acc += wsA[i]; wsC[i] = acc; }
    red[tid] = acc;
    __syncthreads();
    if (tid == 0) { double run = 0.0; for (int t = 0; t < 256; ++t) { const double v = red[t]; red[t] = run; run += v; } }
    __syncthreads();
    const double offs = red[tid];
    double mx = -INFINITY; bool anynan = false;
    for (int i = c0; i < c1; ++i) { const double v = wsC[i] + offs; wsC[i] = v; if (v != v) anynan = true; mx = v > mx ? v : mx; }
    __syncthreads();
    red[tid] = anynan ? __builtin_nan("") : mx;
    __syncthreads();
    if (tid == 0) {
      double m = -INFINITY; bool nn = false;
      for (int t = 0; t < 256; ++t) { const double v = red[t]; if (v != v) nn = true; else m = v > m ? v : m; }
      scal_s[0] = nn ? __builtin_nan("") : m;                           // ndarray.max() propagates NaN
      nvalid_s = 0;
    }
    __syncthreads();
    const double cmax = scal_s[0];
    for (int i = tid; i < n; i += 256) wsC[i] = wsC[i] / cmax;           // cdf /= cdf.max()
    __syncthreads();
    // ---- x_per_sigma = nanmedian(gradient(cdf) / r): bitonic sort of the ratios in LDS
    int n2 = 1;
    while (n2 < n) n2 <<= 1;
    int nv = 0;
    for (int i = tid; i < n2; i += 256) {
      double q = INFINITY;
      if (i < n) {
        const double g = (i == 0) ? (wsC[1] - wsC[0]) : ((i == n - 1) ? (wsC[n - 1] - wsC[n - 2]) : (wsC[i + 1] - wsC[i - 1]) / 2.0);
        q = g / wsA[i];
        if (q != q) q = INFINITY; else ++nv;
      }
      sortb[i] = q;
    }
    atomicAdd(&nvalid_s, nv);
    for (int k = 2; k <= n2; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        __syncthreads();
        for (int i = tid; i < n2; i += 256) {
          const int p = i ^ j;
          if (p > i) {
            const double x0 = sortb[i], x1 = sortb[p];
            const bool up = (i & k) == 0;
            if ((x0 > x1) == up) { sortb[i] = x1; sortb[p] = x0; }
          }
        }
      }
    __syncthreads();
    if (tid == 0) {
      const int m = nvalid_s;
      const double xps = m == 0 ? __builtin_nan("") : ((m & 1) ? sortb[m >> 1] : 0.5 * (sortb[(m >> 1) - 1] + sortb[m >> 1]));
      scal_s[1] = xps;
      const double N = 2.0 / xps;                                       // pix_per_sigma = 2
      int nx = 0;
      if (N == N && N > 0.0 && N <= (double)T.n1) { nx = 1; while ((double)nx < N) nx <<= 1; }
      nx_s = nx;
    }
    __syncthreads();
  }
  const int nx = bad ? 0 : nx_s;
  bad = bad || nx < 8 || nx > T.n1;        // x_per_sigma NaN / a grid finer than the context's FFT tables: NaN result
  const float* conv = bufA;
  if (!bad) {
    // ---- lam = np.interp(linspace(0, 1, nx), cdf, w);  newspec = np.interp(lam, w, s)
    const double step = 1.0 / (double)(nx - 1);
    for (int j = tid; j < nx; j += 256) {
      const double x = (j == nx - 1) ? 1.0 : (double)j * step;
      double lamj; int k;
      if (x > wsC[n - 1]) { lamj = W(n - 1); k = n - 2; }
      else if (x < wsC[0]) { lamj = W(0); k = 0; }
      else {
        int lo = 0, hi = n;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (wsC[mid] <= x) lo = mid; else hi = mid; }
        k = lo;
        if (k == n - 1 || wsC[k] == x) lamj = W(k);
        else { const double slope = (W(k + 1) - W(k)) / (wsC[k + 1] - wsC[k]); lamj = slope * (x - wsC[k]) + W(k); }
        if (k > n - 2) k = n - 2;
      }
      wsL[j] = lamj;
      // np.interp(lamj, w, s): w[k2] <= lamj < w[k2+1]; k is right up to rounding at the pixel edges
      int k2 = k;
      while (k2 > 0 && lamj < W(k2)) --k2;
      while (k2 < n - 2 && lamj >= W(k2 + 1)) ++k2;
      const double s0 = (double)nan_to_zero(sp[i0 + k2]), s1 = (double)nan_to_zero(sp[i0 + k2 + 1]);   // nan_to_num(nan=1.0), shifted
      double v;
      if (lamj >= W(n - 1)) v = (double)nan_to_zero(sp[i0 + n - 1]);
      else if (lamj <= W(0)) v = (double)nan_to_zero(sp[i0]);
      else v = (s1 - s0) / (W(k2 + 1) - W(k2)) * (lamj - W(k2)) + s0;
      bufB[j] = (float)v;
    }
    __syncthreads();
    // ---- smooth_fft(dx = 1/nx, newspec, x_per_sigma): taper exp(-2 pi^2 sigma^2 k^2)
    DevExecT<true, false> ex;
    TaperArgs ta{};
    const double xps = scal_s[1];
    ta.g_c2 = (float)(-2.0 * (kPi * kPi) * (xps * xps) * 1.4426950408889634);
    bool no_edge = false;
    conv = conv_stage<0, 256, false>(ex, T, T.tw, bufB, bufA, nx, ta, no_edge);
  }
  // ---- np.interp(outwave, lam, conv) (clamped), blaze, chi^2
  const bool cheb = T.npoly > 0 && a.out_stage != 2, hasf = T.obs_f1 != nullptr;
  double accx = 0.0;
  for (int i = tid; i < nobs; i += 256) {
    float m1 = nanf_();
    if (!bad) {
      const double x = a.obs_wave[i];
      if (x > wsL[nx - 1]) m1 = conv[nx - 1];
      else if (x < wsL[0]) m1 = conv[0];
      else {
        int lo = 0, hi = nx;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (wsL[mid] <= x) lo = mid; else hi = mid; }
        const int j = lo;
        if (j == nx - 1 || wsL[j] == x) m1 = conv[j];
        else m1 = (float)(((double)conv[j + 1] - (double)conv[j]) / (wsL[j + 1] - wsL[j]) * (x - wsL[j]) + (double)conv[j]);
      }
    }
    double pv = 1.0;
    if (cheb) {                                           // chebval, fitutils.py:11-20
      const double xc = T.xcheb[i];
      const int nc = T.npoly;
      double c0, c1;
      if (nc == 1) { c0 = th[8]; c1 = 0.0; }
      else if (nc == 2) { c0 = th[8]; c1 = th[9]; }
      else {
        const double x2 = 2.0 * xc;
        c0 = th[8 + nc - 2]; c1 = th[8 + nc - 1];
        for (int r = 3; r <= nc; ++r) { const double t = c0; c0 = th[8 + nc - r] - c1; c1 = t + c1 * x2; }
      }
      pv = c0 + c1 * xc;
    }
    const double model1 = (double)m1 * pv + (pv - 1.0);   // (m - 1) p + (p - 1) = m p - 1
    if (a.out) a.out[(size_t)b * a.ld_out + i] = (float)(model1 + 1.0);
    if (hasf && a.lnl) { const double d = model1 - (double)T.obs_f1[i]; accx += (d * d) * (double)T.obs_ivar[i]; }
  }
  if (a.lnl) {
    __syncthreads();
    red[tid] = accx;
    __syncthreads();
    if (tid == 0) {
      double x2 = 0.0;
      for (int t = 0; t < 256; ++t) x2 += red[t];
      if (a.mags) x2 += sed_chi2(a.mags + (size_t)b * a.n_filters, a.obs_mag, a.obs_err, a.n_filters);
      a.lnl[b] = -0.5 * x2;
    }
  }
}

typedef void (*post_kernel_fn)(const PostTables, PostArgs);
// compile-time FFT geometry for the common spectrum lengths, runtime geometry otherwise
static post_kernel_fn pick_post_kernel(int n1, bool tw_lds, bool lean = false) {
  if (lean) {                                   // likelihood-only builds of the two LDS-twiddle sizes that matter
    if (tw_lds && n1 == 4096) return payne_post_kernel<12, true, true>;
    if (tw_lds && n1 == 2048) return payne_post_kernel<11, true, true>;
    if (!tw_lds && n1 == 8192) return payne_post_kernel<13, false, true>;
  }
  if (tw_lds) {
    switch (n1) {
      case 1024: return payne_post_kernel<10, true>;
      case 2048: return payne_post_kernel<11, true>;
      case 4096: return payne_post_kernel<12, true>;
      default: return payne_post_kernel<0, true>;
    }
  }
  switch (n1) {
    case 8192: return payne_post_kernel<13, false>;
    default: return payne_post_kernel<0, false>;
  }
}

// photometry-only fits: lnL = -0.5 chi2_sed
__global__ void payne_photonly_kernel(const double* mags, const double* obs, const double* err, int F, int B, double* lnl) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) lnl[b] = -0.5 * sed_chi2(mags + (size_t)b * F, obs, err, F);
}

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
    for (int k = 0; k < H; ++k) z += (double)w[(size_t)k * H] * a1[k];
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

// ============================================================================
// context
// ============================================================================
static std::string g_create_error;

struct payne_ctx {
  int device = 0;
  payne_opts opts{};
  std::string err;
  std::vector<void*> owned;       // freed at destroy
  std::vector<void*> obs_owned;   // freed when the observed grid is re-bound
  // spectral model
  bool has_model = false;
  int n_layers = 0;
  payne_layer layers[PAYNE_MAX_LAYERS];
  int n_labels = 0;
  double xmin[PAYNE_MAX_LABELS], xden[PAYNE_MAX_LABELS];
  HostTables H;
  PostTables T{};
  float* hid[2] = {nullptr, nullptr};
  int ld_hid = 0;
  float* raw = nullptr;
  unsigned short* w_planes = nullptr;   // bf16 x 3 split of the output layer's weights
  const float* w_out_pad = nullptr;     // output layer's weights [N][w_out_kp], k zero-padded to a multiple of 32 (LDS-DMA kernel)
  int w_out_kp = 0;
  bool dma_ok = false;                  // hidden buffers are zero beyond the last hidden width
  unsigned short* a_planes = nullptr;   // [3][b_max][ld_hid] bf16 planes of the last hidden layer's activations (bx3dma kernel)
  int wp_Kp = 0, wp_Npad = 0;
  size_t post_lds = 0;
  void (*post_fn_lean)(const PostTables, PostArgs) = nullptr;   // likelihood-only instantiation (same LDS)
  bool post_tw_lds = false;
  PostTables* d_T = nullptr;          // device copy of T
  post_kernel_fn post_fn = nullptr;
  float* big_ws = nullptr;            // global spectrum buffers of payne_post_big_kernel (n1 > 16384)
  int big_grid = 0;
  // optional continuum network (payne_ctx_set_continuum; ystpred.py:81-85, 191-209)
  bool has_cont = false;
  int cn_layers = 0, cn_npix = 0, cn_ld_hid = 0;
  payne_layer clayers[PAYNE_MAX_LAYERS];
  double cxmin[PAYNE_MAX_LABELS], cxden[PAYNE_MAX_LABELS];
  float* chid[2] = {nullptr, nullptr};
  float* cont_raw = nullptr;            // [b_max][cn_npix] continuum ANN output (F_nu)
  const double* cont_scale = nullptr;   // [cn_npix] (lam_ref / lam_c)^2 : F_nu -> F_lambda up to a constant the median removes
  const int* cont_idx = nullptr;        // [npix] np.interp(modwave, modcontwave, .) map: left pixel (-1: outside -> NaN)
  const double* cont_frac = nullptr;    // [npix] weight of the right pixel
  std::vector<void*> cont_owned;
  // optional LSF vector (payne_ctx_set_lsf): dispersion in AA per bound observed pixel; replaces Inst_R
  bool has_lsf = false;
  const double* d_obs_wave = nullptr;   // [nobs] (obs_owned)
  const double* lsf = nullptr;          // [nobs]
  float* lsf_spec = nullptr;            // [b_max][npix] spectra after vsini, shifted
  double* lsf_ws = nullptr;             // [b_max][2 npix + n1]
  std::vector<void*> lsf_owned;
  bool obs_bound = false;
  CandState* prep = nullptr;      // [b_max] per-candidate records of the post kernel (written by the first dense launch)
  bool prep_valid = false;        // ... as of the last run_ann
  // photometry
  bool has_phot = false, has_obs_phot = false;
  PhotTables P{};
  double* mags_ws = nullptr;
  double *obs_mag = nullptr, *obs_err = nullptr;
  int ncols = 0;
  // optional per-kernel HIP-event timing (payne_profile): kinds 0 output dense layer,
  // 1 post, 2 sed, 3 hidden dense layers
  bool prof = false;
  struct ProfRec { hipEvent_t e0, e1; int kind; };
  std::vector<ProfRec> prof_pool;
  size_t prof_used = 0;
  double prof_ms[4] = {0, 0, 0, 0};
  long long prof_n[4] = {0, 0, 0, 0};
};

// RAII bracket of one timed launch.  The event pair is handed to the launch itself (hipExtLaunchKernelGGL:
// start/stop are the kernel's own dispatch timestamps -- what rocprofv3 reports); events recorded AROUND a
// launch on the stream read ~2 us more (their own packets).  A scope that sees no PAYNE_LAUNCH gives its
// record back; a scope with several launches times the first.
struct ProfScope;
static thread_local ProfScope* g_prof_scope = nullptr;
struct ProfScope {
  payne_ctx* c; hipStream_t s; payne_ctx::ProfRec* r = nullptr; bool used = false; ProfScope* outer = nullptr;
  ProfScope(payne_ctx* c_, hipStream_t s_, int kind) : c(c_), s(s_) {
    outer = g_prof_scope; g_prof_scope = this;
    if (!c->prof) return;
    if (c->prof_used == c->prof_pool.size()) {
      payne_ctx::ProfRec n{};
      if (hipEventCreate(&n.e0) != hipSuccess || hipEventCreate(&n.e1) != hipSuccess) return;
      c->prof_pool.push_back(n);
    }
    r = &c->prof_pool[c->prof_used++];
    r->kind = kind;
  }
  ~ProfScope() {
    g_prof_scope = outer;
    if (r && !used) --c->prof_used;                      // nothing was launched under this scope
  }
};
#include <hip/hip_ext.h>
#define PAYNE_LAUNCH(kernel, grid, block, lds, stream, ...)                                               \
  do {                                                                                                    \
    ProfScope* ps_ = g_prof_scope;                                                                        \
    if (ps_ && ps_->r && !ps_->used) {                                                                    \
      ps_->used = true;                                                                                   \
      hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)(lds), stream, ps_->r->e0, ps_->r->e1, 0, __VA_ARGS__); \
    } else {                                                                                              \
      hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                  \
    }                                                                                                     \
  } while (0)

#define HIPCHK(ctx, call)                                                                 \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                     \
      return PAYNE_E_HIP;                                                                 \
    }                                                                                     \
  } while (0)

template <class V>
static int upload(payne_ctx* c, const std::vector<V>& v, const V** out, std::vector<void*>& bag) {
  void* d = nullptr;
  size_t bytes = v.size() * sizeof(V);
  if (bytes == 0) { *out = nullptr; return PAYNE_OK; }
  HIPCHK(c, hipMalloc(&d, bytes));
  bag.push_back(d);
  HIPCHK(c, hipMemcpy(d, v.data(), bytes, hipMemcpyHostToDevice));
  *out = reinterpret_cast<const V*>(d);
  return PAYNE_OK;
}
template <class V>
static int dev_alloc(payne_ctx* c, size_t n, V** out, std::vector<void*>& bag, bool zero = true) {
  void* d = nullptr;
  HIPCHK(c, hipMalloc(&d, n * sizeof(V)));
  bag.push_back(d);
  if (zero) HIPCHK(c, hipMemset(d, 0, n * sizeof(V)));
  *out = reinterpret_cast<V*>(d);
  return PAYNE_OK;
}

static int fail(payne_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg; else g_create_error = msg;
  return code;
}

static int sync_tables(payne_ctx* c) {
  if (!c->d_T) return PAYNE_OK;
  HIPCHK(c, hipMemcpy(c->d_T, &c->T, sizeof(PostTables), hipMemcpyHostToDevice));
  return PAYNE_OK;
}

static int bind_obs(payne_ctx* c, const payne_obs_desc* obs) {
  for (void* p : c->obs_owned) (void)hipFree(p);
  c->obs_owned.clear();
  c->obs_bound = false;
  for (void* p : c->lsf_owned) (void)hipFree(p);      // an LSF vector belongs to the grid it was given on
  c->lsf_owned.clear(); c->has_lsf = false; c->d_obs_wave = nullptr;
  c->T.nobs = 0; c->T.lnobs = nullptr; c->T.obs_rec = nullptr; c->T.xcheb = nullptr; c->T.obs_f1 = nullptr; c->T.obs_ivar = nullptr;
  if (!obs || obs->nobs <= 0) return sync_tables(c);
  if (!obs->wave) return fail(c, PAYNE_E_INVALID, "obs.wave is NULL");
  if ((obs->flux == nullptr) != (obs->eflux == nullptr)) return fail(c, PAYNE_E_INVALID, "obs.flux and obs.eflux must both be given or both NULL");
  build_obs_tables(obs->wave, obs->flux, obs->eflux, obs->nobs, c->H);
  int rc;
  if ((rc = upload(c, c->H.lnobs, &c->T.lnobs, c->obs_owned))) return rc;
  if ((rc = upload(c, c->H.obs_rec, &c->T.obs_rec, c->obs_owned))) return rc;
  if ((rc = upload(c, c->H.obs_wave, &c->d_obs_wave, c->obs_owned))) return rc;
  if ((rc = upload(c, c->H.xcheb, &c->T.xcheb, c->obs_owned))) return rc;
  if (c->H.has_flux) {
    if ((rc = upload(c, c->H.obs_f1, &c->T.obs_f1, c->obs_owned))) return rc;
    if ((rc = upload(c, c->H.obs_ivar, &c->T.obs_ivar, c->obs_owned))) return rc;
  }
  c->T.nobs = obs->nobs;
  c->T.obs_min = c->H.obs_min;
  c->T.obs_max = c->H.obs_max;
  c->obs_bound = true;
  return sync_tables(c);
}

extern "C" int payne_version(void) { return PAYNE_ABI_VERSION; }

extern "C" const char* payne_last_error(const payne_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int payne_theta_cols(const payne_ctx* ctx) { return ctx ? ctx->ncols : PAYNE_E_INVALID; }

extern "C" const char* payne_kernel_name(int which) {
  switch (which) {
    case 0: return "payne_dense_kernel";
    case 1: return "payne_post_kernel";
    case 2: return "payne_sed_kernel";
    default: return "";
  }
}

extern "C" void payne_ctx_destroy(payne_ctx* c) {
  if (!c) return;
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(c->device);
  for (void* p : c->owned) (void)hipFree(p);
  for (void* p : c->obs_owned) (void)hipFree(p);
  for (void* p : c->cont_owned) (void)hipFree(p);
  for (void* p : c->lsf_owned) (void)hipFree(p);
  for (auto& r : c->prof_pool) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  (void)hipSetDevice(prev);
  delete c;
}

extern "C" int payne_ctx_create(const payne_model_desc* model, const payne_obs_desc* obs, const payne_phot_desc* phot,
                                const payne_opts* opts, int device, payne_ctx** out) {
  if (!out) return fail(nullptr, PAYNE_E_INVALID, "out is NULL");
  *out = nullptr;
  if (!opts || opts->b_max <= 0) return fail(nullptr, PAYNE_E_INVALID, "opts.b_max must be > 0");
  if (opts->npoly < 0 || opts->npoly > PAYNE_MAX_POLY) return fail(nullptr, PAYNE_E_INVALID, "opts.npoly out of range");
  if (!model && !phot) return fail(nullptr, PAYNE_E_INVALID, "need a spectral model and/or a photometric model");
  hipError_t he = hipSetDevice(device);
  if (he != hipSuccess) return fail(nullptr, PAYNE_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(he));
  payne_ctx* c = new payne_ctx();
  c->device = device;
  c->opts = *opts;
  c->ncols = 8 + opts->npoly + 4;
  int rc = PAYNE_OK;
  auto bail = [&](int code) { g_create_error = c->err; payne_ctx_destroy(c); return code; };

  if (model) {
    if (model->n_layers < 2 || model->n_layers > PAYNE_MAX_LAYERS) return bail(fail(c, PAYNE_E_INVALID, "model.n_layers must be 2..8"));
    if (model->n_labels < 1 || model->n_labels > PAYNE_MAX_LABELS) return bail(fail(c, PAYNE_E_INVALID, "model.n_labels must be 1..5"));
    if (!model->xmin || !model->xmax || !model->wavelength) return bail(fail(c, PAYNE_E_INVALID, "model.xmin/xmax/wavelength missing"));
    if (model->layers[0].n_in != model->n_labels) return bail(fail(c, PAYNE_E_INVALID, "first layer n_in != n_labels"));
    if (model->layers[model->n_layers - 1].n_out != model->npix) return bail(fail(c, PAYNE_E_INVALID, "last layer n_out != npix"));
    int maxh = 0;
    for (int l = 0; l < model->n_layers; ++l) {
      const payne_layer& L = model->layers[l];
      if (!L.w || !L.b || L.n_in <= 0 || L.n_out <= 0) return bail(fail(c, PAYNE_E_INVALID, "layer with null weights or bad shape"));
      if (l > 0 && L.n_in != model->layers[l - 1].n_out) return bail(fail(c, PAYNE_E_INVALID, "layer shapes do not chain"));
      c->layers[l] = L;
      if (l > 0 && (L.n_in & 3)) {   // float4 tile loads need K % 4 == 0: keep a zero-padded copy
        const int Kp = (L.n_in + 3) & ~3;
        float* wp = nullptr;
        if ((rc = dev_alloc(c, (size_t)L.n_out * Kp, &wp, c->owned))) return bail(rc);
        he = hipMemcpy2D(wp, (size_t)Kp * 4, L.w, (size_t)L.n_in * 4, (size_t)L.n_in * 4, L.n_out, hipMemcpyDeviceToDevice);
        if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipMemcpy2D: ") + hipGetErrorString(he)));
        c->layers[l].w = wp;
        c->layers[l].n_in = Kp;     // padded K (extra columns are zero)
      }
      if (l + 1 < model->n_layers) maxh = std::max(maxh, L.n_out);
    }
    {   // k-padded copy of the output layer's weights for the LDS-DMA kernel (operands cannot be masked on the way)
      const payne_layer& L = c->layers[model->n_layers - 1];
      const int Kp = (L.n_in + 31) & ~31;
      float* wp = nullptr;
      if ((rc = dev_alloc(c, (size_t)L.n_out * Kp, &wp, c->owned))) return bail(rc);
      he = hipMemcpy2D(wp, (size_t)Kp * 4, L.w, (size_t)L.n_in * 4, (size_t)L.n_in * 4, L.n_out, hipMemcpyDeviceToDevice);
      if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipMemcpy2D: ") + hipGetErrorString(he)));
      c->w_out_pad = wp; c->w_out_kp = Kp;
      // the activations' pad columns are zero only if no wider layer ever wrote them: all hidden widths equal
      bool same = model->n_layers >= 3;
      for (int l = 1; l + 1 < model->n_layers; ++l) same = same && model->layers[l].n_out == model->layers[0].n_out;
      c->dma_ok = same;
    }
    {   // exact 3 x bf16 split of the output layer's weights (payne_dense_bf16x3_kernel)
      const payne_layer& L = c->layers[model->n_layers - 1];
      const int K = L.n_in, N = L.n_out;
      const int Kp = (K + 31) & ~31, Npad = (N + 127) & ~127;
      std::vector<float> w((size_t)N * K);
      he = hipMemcpy(w.data(), L.w, w.size() * 4, hipMemcpyDeviceToHost);
      if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipMemcpy(W out): ") + hipGetErrorString(he)));
      std::vector<unsigned short> pl((size_t)3 * Npad * Kp, 0);
      auto to_bf16 = [](float x) -> unsigned short {            // round to nearest even (NaN kept quiet)
        unsigned u; std::memcpy(&u, &x, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
        return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
      };
      auto from_bf16 = [](unsigned short b) -> float { unsigned u = (unsigned)b << 16; float f; std::memcpy(&f, &u, 4); return f; };
      for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
          const float x = w[(size_t)n * K + k];
          const unsigned short b1 = to_bf16(x);
          const float r1 = x - from_bf16(b1);
          const unsigned short b2 = to_bf16(r1);
          const float r2 = r1 - from_bf16(b2);
          const unsigned short b3 = to_bf16(r2);
          pl[((size_t)0 * Npad + n) * Kp + k] = b1;
          pl[((size_t)1 * Npad + n) * Kp + k] = b2;
          pl[((size_t)2 * Npad + n) * Kp + k] = b3;
        }
      const unsigned short* dp = nullptr;
      if ((rc = upload(c, pl, &dp, c->owned))) return bail(rc);
      c->w_planes = const_cast<unsigned short*>(dp);
      c->wp_Kp = Kp; c->wp_Npad = Npad;
    }
    c->n_layers = model->n_layers;
    c->n_labels = model->n_labels;
    for (int d = 0; d < model->n_labels; ++d) { c->xmin[d] = model->xmin[d]; c->xden[d] = model->xmax[d] - model->xmin[d]; }
    rc = build_model_tables(model->wavelength, model->npix, c->H);
    if (rc == -1) return bail(fail(c, PAYNE_E_INVALID, "model.npix must be >= 16"));
    if (rc == -2) return bail(fail(c, PAYNE_E_INVALID, "model.wavelength must be strictly increasing"));
    if (c->H.n1 > (1 << 20)) return bail(fail(c, PAYNE_E_UNSUPPORTED, "npix > 2^20"));
    PostTables& T = c->T;
    fill_model_scalars(c->H, T);
    T.r_ann = model->resolution; T.npoly = opts->npoly;
    if ((rc = upload(c, c->H.vs_tab32, &T.vs_tab, c->owned))) return bail(rc);
    if ((rc = dev_alloc(c, 1, &c->d_T, c->owned))) return bail(rc);
    if ((rc = upload(c, c->H.lnlam, &T.lnlam, c->owned))) return bail(rc);
    if ((rc = upload(c, c->H.lam, &T.lam, c->owned))) return bail(rc);
    if ((rc = upload(c, c->H.tw, &T.tw, c->owned))) return bail(rc);
    if ((rc = upload(c, c->H.twf, &T.twf, c->owned))) return bail(rc);
    if ((rc = upload(c, c->H.rs1_idx, &T.rs1_idx, c->owned))) return bail(rc);
    if ((rc = upload(c, c->H.rs1_frac, &T.rs1_frac, c->owned))) return bail(rc);
    if ((rc = upload(c, c->H.bk1_idx, &T.bk1_idx, c->owned))) return bail(rc);
    if ((rc = upload(c, c->H.bk1_frac, &T.bk1_frac, c->owned))) return bail(rc);
    c->ld_hid = (maxh + 31) & ~31;
    if (model->n_layers > 2) {
      if ((rc = dev_alloc(c, (size_t)opts->b_max * c->ld_hid, &c->hid[0], c->owned))) return bail(rc);
      if ((rc = dev_alloc(c, (size_t)opts->b_max * c->ld_hid, &c->hid[1], c->owned))) return bail(rc);
    }
    if ((rc = dev_alloc(c, (size_t)opts->b_max * model->npix, &c->raw, c->owned, false))) return bail(rc);
    if (c->dma_ok && (rc = dev_alloc(c, (size_t)3 * opts->b_max * c->ld_hid, &c->a_planes, c->owned))) return bail(rc);
    if ((rc = dev_alloc(c, (size_t)opts->b_max, &c->prep, c->owned, false))) return bail(rc);
    if (T.n1 > 16384) {                // spectrum larger than LDS: global-workspace kernel
      c->big_grid = opts->b_max < 256 ? opts->b_max : 256;
      if ((rc = dev_alloc(c, (size_t)c->big_grid * 2 * T.n1, &c->big_ws, c->owned, false))) return bail(rc);
    }
    c->post_lds = (size_t)(T.n1 > 16384 ? 64 : fft_buf_floats(T.n1)) * 8 + (size_t)scratch_doubles(kPostThreads) * 8 + ((sizeof(CandState) + 15) & ~(size_t)15) + 16;
    // twiddles in LDS while two workgroups still fit a CU (160 KiB); larger spectra read them from L2
    const size_t tw_bytes = c->H.twf.size() * sizeof(c32);         // ~0.75 n1 entries
    c->post_tw_lds = (c->post_lds + tw_bytes) <= 80 * 1024;
    if (getenv("PAYNE_TW_GLOBAL")) c->post_tw_lds = false;
    if (c->post_tw_lds) c->post_lds += tw_bytes;
    c->post_fn = pick_post_kernel(getenv("PAYNE_POST_GENERIC") ? 0 : T.n1, c->post_tw_lds);
    c->post_fn_lean = getenv("PAYNE_POST_FULL") ? c->post_fn : pick_post_kernel(getenv("PAYNE_POST_GENERIC") ? 0 : T.n1, c->post_tw_lds, true);
    he = hipFuncSetAttribute(reinterpret_cast<const void*>(c->post_fn_lean), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->post_lds);
    if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(he)));
    he = hipFuncSetAttribute(reinterpret_cast<const void*>(c->post_fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->post_lds);
    if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(he)));
    c->has_model = true;
    if ((rc = bind_obs(c, obs))) return bail(rc);
  }

  if (phot) {
    if (phot->n_filters <= 0 || phot->hidden <= 0 || phot->hidden > 2048) return bail(fail(c, PAYNE_E_INVALID, "phot.n_filters/hidden out of range"));
    if (!phot->w1 || !phot->b1 || !phot->w2 || !phot->b2 || !phot->w3 || !phot->b3 || !phot->xmin || !phot->xmax)
      return bail(fail(c, PAYNE_E_INVALID, "phot descriptor has NULL members"));
    PhotTables& P = c->P;
    const int F = phot->n_filters, H = phot->hidden;
    P.F = F; P.H = H;
    P.w1 = phot->w1; P.b1 = phot->b1; P.b2 = phot->b2; P.w3 = phot->w3; P.b3 = phot->b3;
    for (int d = 0; d < 6; ++d) { P.xmin[d] = phot->xmin[d]; P.xden[d] = phot->xmax[d] - phot->xmin[d]; }
    {   // w2 -> [F][k][h] so that lanes (h) read consecutive addresses
      std::vector<float> w2((size_t)F * H * H), w2t((size_t)F * H * H);
      he = hipMemcpy(w2.data(), phot->w2, w2.size() * 4, hipMemcpyDeviceToHost);
      if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipMemcpy(w2): ") + hipGetErrorString(he)));
      for (int f = 0; f < F; ++f)
        for (int h = 0; h < H; ++h)
          for (int k = 0; k < H; ++k) w2t[((size_t)f * H + k) * H + h] = w2[((size_t)f * H + h) * H + k];
      if ((rc = upload(c, w2t, &P.w2t, c->owned))) return bail(rc);
    }
    if (phot->hiav) {
      std::vector<double> hv(phot->hiav, phot->hiav + (size_t)F * 5);
      if ((rc = upload(c, hv, &P.hiav, c->owned))) return bail(rc);
    }
    if (phot->obs_mag && phot->obs_err) {
      std::vector<double> m(phot->obs_mag, phot->obs_mag + F), e(phot->obs_err, phot->obs_err + F);
      const double *dm, *de;
      if ((rc = upload(c, m, &dm, c->owned))) return bail(rc);
      if ((rc = upload(c, e, &de, c->owned))) return bail(rc);
      c->obs_mag = const_cast<double*>(dm); c->obs_err = const_cast<double*>(de);
      c->has_obs_phot = true;
    }
    if ((rc = dev_alloc(c, (size_t)opts->b_max * F, &c->mags_ws, c->owned))) return bail(rc);
    if ((size_t)H * 16 > 48 * 1024) {
      he = hipFuncSetAttribute(reinterpret_cast<const void*>(payne_sed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, H * 16);
      if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipFuncSetAttribute(sed): ") + hipGetErrorString(he)));
    }
    c->has_phot = true;
  }
  *out = c;
  return PAYNE_OK;
}

extern "C" int payne_ctx_set_obs(payne_ctx* c, const payne_obs_desc* obs) {
  if (!c) return PAYNE_E_INVALID;
  if (!c->has_model) return fail(c, PAYNE_E_INVALID, "context has no spectral model");
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (prev != c->device) (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();           // no kernel may still read the old tables
  int rc = bind_obs(c, obs);
  if (prev != c->device) (void)hipSetDevice(prev);
  return rc;
}

// Continuum network (ystpred.PayneSpecPredict(Cnnpath=...)): weights as for the spectral model; the label
// set must be the spectral model's.  cont == NULL removes it.
extern "C" int payne_ctx_set_continuum(payne_ctx* c, const payne_model_desc* cont) {
  if (!c) return PAYNE_E_INVALID;
  if (!c->has_model) return fail(c, PAYNE_E_INVALID, "context has no spectral model");
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (prev != c->device) (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (void* p : c->cont_owned) (void)hipFree(p);
  c->cont_owned.clear();
  c->has_cont = false;
  auto done = [&](int rc) { if (prev != c->device) (void)hipSetDevice(prev); return rc; };
  if (!cont) return done(PAYNE_OK);
  if (cont->n_layers < 3 || cont->n_layers > PAYNE_MAX_LAYERS) return done(fail(c, PAYNE_E_INVALID, "continuum.n_layers must be 3..8"));
  if (cont->n_labels != c->n_labels) return done(fail(c, PAYNE_E_INVALID, "continuum.n_labels != model.n_labels"));
  if (!cont->xmin || !cont->xmax || !cont->wavelength || cont->npix < 2 || cont->npix > 8192)
    return done(fail(c, PAYNE_E_INVALID, "continuum.xmin/xmax/wavelength missing or npix outside 2..8192"));
  if (cont->layers[0].n_in != cont->n_labels || cont->layers[cont->n_layers - 1].n_out != cont->npix)
    return done(fail(c, PAYNE_E_INVALID, "continuum layer shapes do not match n_labels / npix"));
  int rc = PAYNE_OK, maxh = 0;
  for (int l = 0; l < cont->n_layers; ++l) {
    const payne_layer& L = cont->layers[l];
    if (!L.w || !L.b || L.n_in <= 0 || L.n_out <= 0) return done(fail(c, PAYNE_E_INVALID, "continuum layer with null weights or bad shape"));
    if (l > 0 && L.n_in != cont->layers[l - 1].n_out) return done(fail(c, PAYNE_E_INVALID, "continuum layer shapes do not chain"));
    c->clayers[l] = L;
    if (l > 0 && (L.n_in & 3)) {            // float4 tile loads need K % 4 == 0: zero-padded copy
      const int Kp = (L.n_in + 3) & ~3;
      float* wp = nullptr;
      if ((rc = dev_alloc(c, (size_t)L.n_out * Kp, &wp, c->cont_owned))) return done(rc);
      hipError_t he = hipMemcpy2D(wp, (size_t)Kp * 4, L.w, (size_t)L.n_in * 4, (size_t)L.n_in * 4, L.n_out, hipMemcpyDeviceToDevice);
      if (he != hipSuccess) return done(fail(c, PAYNE_E_HIP, std::string("hipMemcpy2D: ") + hipGetErrorString(he)));
      c->clayers[l].w = wp; c->clayers[l].n_in = Kp;
    }
    if (l + 1 < cont->n_layers) maxh = std::max(maxh, L.n_out);
  }
  c->cn_layers = cont->n_layers; c->cn_npix = cont->npix; c->cn_ld_hid = (maxh + 31) & ~31;
  for (int d = 0; d < cont->n_labels; ++d) { c->cxmin[d] = cont->xmin[d]; c->cxden[d] = cont->xmax[d] - cont->xmin[d]; }
  if ((rc = dev_alloc(c, (size_t)c->opts.b_max * c->cn_ld_hid, &c->chid[0], c->cont_owned))) return done(rc);
  if ((rc = dev_alloc(c, (size_t)c->opts.b_max * c->cn_ld_hid, &c->chid[1], c->cont_owned))) return done(rc);
  if ((rc = dev_alloc(c, (size_t)c->opts.b_max * cont->npix, &c->cont_raw, c->cont_owned, false))) return done(rc);
  // host tables: F_nu -> F_lambda factor (constants cancel in the median normalisation) and the np.interp map
  const int npc = cont->npix, npix = c->T.npix;
  std::vector<double> scale(npc), frac(npix);
  std::vector<int> idx(npix);
  const double* wc = cont->wavelength;
  for (int i = 1; i < npc; ++i)
    if (!(wc[i] > wc[i - 1])) return done(fail(c, PAYNE_E_INVALID, "continuum.wavelength must be strictly increasing"));
  const double lref = wc[npc / 2];
  for (int i = 0; i < npc; ++i) { const double r = lref / wc[i]; scale[i] = r * r; }
  for (int i = 0; i < npix; ++i) {
    const double x = c->H.lam[i];
    if (x < wc[0] || x > wc[npc - 1]) { idx[i] = -1; frac[i] = 0.0; continue; }   // left = right = NaN
    int k = (int)(std::upper_bound(wc, wc + npc, x) - wc) - 1;
    if (k > npc - 2) k = npc - 2;
    idx[i] = k;
    frac[i] = (x - wc[k]) / (wc[k + 1] - wc[k]);
  }
  if ((rc = upload(c, scale, &c->cont_scale, c->cont_owned))) return done(rc);
  if ((rc = upload(c, idx, &c->cont_idx, c->cont_owned))) return done(rc);
  if ((rc = upload(c, frac, &c->cont_frac, c->cont_owned))) return done(rc);
  c->has_cont = true;
  return done(PAYNE_OK);
}

// LSF vector for the instrumental broadening: `lsf[n]` = Gaussian dispersion (AA) at each pixel of the bound
// observed grid (getspec(inst_R=array, outwave=...), ystpred.py:248-269).  While set, theta's Inst_R column
// is ignored.  NULL removes it; re-binding the observed grid removes it too.
extern "C" int payne_ctx_set_lsf(payne_ctx* c, const double* lsf, int n) {
  if (!c) return PAYNE_E_INVALID;
  if (!c->has_model) return fail(c, PAYNE_E_INVALID, "context has no spectral model");
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (prev != c->device) (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (void* p : c->lsf_owned) (void)hipFree(p);
  c->lsf_owned.clear();
  c->has_lsf = false;
  auto done = [&](int rc) { if (prev != c->device) (void)hipSetDevice(prev); return rc; };
  if (!lsf) return done(PAYNE_OK);
  if (!c->obs_bound) return done(fail(c, PAYNE_E_INVALID, "bind the observed grid before its LSF vector"));
  if (n != c->T.nobs) return done(fail(c, PAYNE_E_INVALID, "the LSF vector must have one entry per observed pixel"));
  if (c->T.n1 > 8192) return done(fail(c, PAYNE_E_UNSUPPORTED, "LSF broadening is built for spectra up to 8192 pixels"));
  for (int i = 0; i < n; ++i)
    if (!(lsf[i] > 0.0)) return done(fail(c, PAYNE_E_INVALID, "LSF dispersions must be positive"));
  std::vector<double> v(lsf, lsf + n);
  int rc;
  if ((rc = upload(c, v, &c->lsf, c->lsf_owned))) return done(rc);
  if ((rc = dev_alloc(c, (size_t)c->opts.b_max * c->T.npix, &c->lsf_spec, c->lsf_owned, false))) return done(rc);
  if ((rc = dev_alloc(c, (size_t)c->opts.b_max * (2 * (size_t)c->T.npix + c->T.n1), &c->lsf_ws, c->lsf_owned, false))) return done(rc);
  const size_t lds = (size_t)c->T.n1 * 8 + 2 * (size_t)fft_buf_floats(c->T.n1) * 4 + (256 + 8) * 8;
  hipError_t he = hipFuncSetAttribute(reinterpret_cast<const void*>(payne_lsf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (he != hipSuccess) return done(fail(c, PAYNE_E_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(he)));
  c->has_lsf = true;
  return done(PAYNE_OK);
}

// ---- launches --------------------------------------------------------------
template <int BM, int BN, int BK, bool FUSE>
static void launch_dense(DenseParams& p, hipStream_t s) {
  p.grid_m = (p.B + BM - 1) / BM;
  p.grid_n = (p.N + BN - 1) / BN;
  constexpr size_t lds = dense_lds_bytes<BM, BN, BK>();
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(payne_dense_kernel<BM, BN, BK, FUSE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
#ifdef PAYNE_STAMPS
  p.stamps = FUSE ? nullptr : g_dense_stamps;
#endif
  PAYNE_LAUNCH((payne_dense_kernel<BM, BN, BK, FUSE>), dim3(p.grid_m * p.grid_n), dim3(256), lds, s, p);
}

// Timing experiments only (results are invalid): PAYNE_SKIP bit 0 = hidden layers, 1 = output layer, 2 = post kernel.
static int skip_mask() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("PAYNE_SKIP"); v = e ? atoi(e) : 0; }
  return v;
}

static int out_tile_choice() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("PAYNE_OUT_TILE"); v = e ? atoi(e) : 8; }   // 8: fp32 LDS-DMA (default); 9: 3 x bf16 split on the LDS-DMA ring (both need zero-padded operands, else 0); 0: register-staged streaming; 6: K-resident; 7: register-staged bf16x3
  return v;
}

static void launch_out_resident(DenseParams& p, hipStream_t s) {
  p.grid_m = (p.B + 63) / 64;
  const int ntiles = (p.N + 31) / 32;
  int groups = 256 / (p.grid_m > 0 ? p.grid_m : 1);             // ~ one workgroup per CU
  if (groups < 1) groups = 1;
  if (groups > ntiles) groups = ntiles;
  const int tiles_per_wg = (ntiles + groups - 1) / groups;
  p.grid_n = (ntiles + tiles_per_wg - 1) / tiles_per_wg;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(payne_dense_out_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)OK_LDS_BYTES);
    attr_set = true;
  }
  PAYNE_LAUNCH(payne_dense_out_kernel, dim3(p.grid_m * p.grid_n), dim3(256), OK_LDS_BYTES, s, p, tiles_per_wg);
}

template <int WN, int BK>
static void launch_out_dma_t(payne_ctx* c, DenseParams& p, hipStream_t s) {
  p.W = c->w_out_pad; p.K = c->w_out_kp;                   // padded pitch; X's pitch (ld_hid) is a multiple of 32 too
  p.grid_m = (p.B + 63) / 64;
  p.grid_n = (p.N + 32 * WN - 1) / (32 * WN);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(payne_dense_dma_kernel<WN, BK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dm_lds_bytes<WN, BK>());
    attr_set = true;
  }
#ifdef PAYNE_STAMPS
  p.stamps = g_dense_stamps;
#endif
  constexpr size_t lds = dm_lds_bytes<WN, BK>();
  PAYNE_LAUNCH((payne_dense_dma_kernel<WN, BK>), dim3(p.grid_m * p.grid_n), dim3(128 * WN), lds, s, p);
}
static void launch_out_dma(payne_ctx* c, DenseParams& p, hipStream_t s) {
  static int wide = -1, deep = -1;                         // PAYNE_DMA_WIDE=0: 64 x 64 tiles (default 64 x 128); PAYNE_DMA_BK=64: 64-deep stages
  if (wide < 0) { const char* e = getenv("PAYNE_DMA_WIDE"); wide = e ? atoi(e) : 1; }
  if (deep < 0) { const char* e = getenv("PAYNE_DMA_BK"); deep = e ? atoi(e) : 32; }
  if (wide && deep == 64 && (c->w_out_kp % 64) == 0) launch_out_dma_t<4, 64>(c, p, s);
  else if (wide) launch_out_dma_t<4, 32>(c, p, s);
  else launch_out_dma_t<2, 32>(c, p, s);
}

static void launch_out_bx3dma(payne_ctx* c, DenseParams& p, hipStream_t s) {
  p.grid_m = (p.B + 63) / 64;
  p.grid_n = (p.N + 63) / 64;
  static int dbg = -1;
  if (dbg < 0) { const char* e = getenv("PAYNE_BD_DBG"); dbg = e ? atoi(e) : 0; }
  Bd3Params q{p, c->a_planes, c->ld_hid, (size_t)c->opts.b_max * c->ld_hid, c->w_planes, c->wp_Kp, c->wp_Npad, dbg};
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(payne_dense_bx3dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BD_LDS_BYTES);
    attr_set = true;
  }
  PAYNE_LAUNCH(payne_dense_bx3dma_kernel, dim3(p.grid_m * p.grid_n), dim3(256), BD_LDS_BYTES, s, q);
}

static void launch_out_bf16x3(payne_ctx* c, DenseParams& p, hipStream_t s) {
  p.grid_m = (p.B + BX_BM - 1) / BX_BM;
  p.grid_n = (p.N + BX_BN - 1) / BX_BN;
  static int dbg = -1;
  if (dbg < 0) { const char* e = getenv("PAYNE_BX_DBG"); dbg = e ? atoi(e) : 0; }
  Bx3Params q{p, c->w_planes, c->wp_Kp, c->wp_Npad, dbg};
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(payne_dense_bf16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BX_LDS_BYTES);
    attr_set = true;
  }
  PAYNE_LAUNCH(payne_dense_bf16x3_kernel, dim3(p.grid_m * p.grid_n), dim3(256), BX_LDS_BYTES, s, q);
}

static int hidden_kernel_choice() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("PAYNE_HIDDEN_KERNEL"); v = e ? atoi(e) : 1; }   // 1: workgroup form, 0: wave-per-tile
  return v;
}

template <bool FUSE>
static void launch_hidden(DenseParams& p, PrepArgs& pa, hipStream_t s) {
  p.grid_m = (p.B + 31) / 32;
  p.grid_n = (p.N + 31) / 32;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(payne_dense_hidden_kernel<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)HK_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(payne_dense_hidden_kernel<true, PAYNE_MAX_LABELS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)HK_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(payne_dense_hidden_kernel<false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)HK_LDS_BYTES);
    attr_set = true;
  }
  pa.n_gemm = p.grid_m * p.grid_n;
  if (!FUSE) pa.out = nullptr;
#ifdef PAYNE_STAMPS
  p.stamps = FUSE ? g_hidden_stamps : nullptr;
#endif
  const dim3 grid(pa.n_gemm + (pa.out ? (p.B + 255) / 256 : 0)), block(256);
  if (!FUSE) PAYNE_LAUNCH((payne_dense_hidden_kernel<false, 4>), grid, block, HK_LDS_BYTES, s, p, pa);
  else if (p.n_labels <= 4) PAYNE_LAUNCH((payne_dense_hidden_kernel<true, 4>), grid, block, HK_LDS_BYTES, s, p, pa);
  else PAYNE_LAUNCH((payne_dense_hidden_kernel<true, PAYNE_MAX_LABELS>), grid, block, HK_LDS_BYTES, s, p, pa);
}

template <bool FUSE>
static void launch_small(DenseParams& p, PrepArgs& pa, hipStream_t s) {
  if (hidden_kernel_choice() == 1) { launch_hidden<FUSE>(p, pa, s); return; }
  pa.out = nullptr;
  p.grid_m = (p.B + 15) / 16;
  p.grid_n = (p.N + 15) / 16;
  PAYNE_LAUNCH((payne_dense_small_kernel<FUSE>), dim3(p.grid_m * p.grid_n), dim3(64), 0, s, p);
}

// ANN forward for the batch -> c->raw [B][npix] (shifted by -1)
static bool prep_enabled() {
  static int v = -1;
  if (v < 0) v = getenv("PAYNE_NO_PREP") ? 0 : 1;
  return v == 1;
}

// One network of the context: the spectral emulator or the continuum network.
struct NetRef {
  const payne_layer* layers; int n_layers; int n_labels; const double* xmin; const double* xden;
  float* const* hid; int ld_hid; float* out; int ld_out; float out_shift;
  bool spectral;                      // the spectral net owns the DMA / bf16x3 operand copies and the prep records
};
static int run_net(payne_ctx* c, const NetRef& N, const double* theta, int B, double instr_factor, hipStream_t s) {
  const int n = N.n_layers;
  for (int l = 1; l < n; ++l) {
    DenseParams p{};
    const payne_layer& L = N.layers[l];
    const bool last = (l == n - 1);
    p.W = L.w; p.K = L.n_in; p.bias = L.b; p.N = L.n_out; p.B = B; p.act = L.act;
    p.bias_shift = last ? N.out_shift : 0.f;
    p.Y = last ? N.out : N.hid[(l - 1) & 1];
    p.ldy = last ? N.ld_out : N.ld_hid;
    if (N.spectral && l == n - 2 && c->a_planes && out_tile_choice() == 9) {   // feeds the output layer: also as bf16 planes
      p.Yp = c->a_planes; p.ldyp = c->ld_hid; p.yp_plane = (size_t)c->opts.b_max * c->ld_hid;
    }
    if (N.spectral && (skip_mask() & (last ? 2 : 1))) continue;
    ProfScope ps(c, s, last ? 0 : 3);
    if (l == 1) {
      const payne_layer& L0 = N.layers[0];
      p.theta = theta; p.ld_theta = c->ncols;
      p.W0 = L0.w; p.b0 = L0.b; p.n_labels = N.n_labels; p.act0 = L0.act; p.K0 = L0.n_out;
      for (int d = 0; d < N.n_labels; ++d) { p.xmin[d] = N.xmin[d]; p.xden[d] = N.xden[d]; }
      PrepArgs pa{};
      pa.T = c->T; pa.instr_factor = instr_factor;
      pa.out = (N.spectral && c->prep && c->obs_bound && prep_enabled()) ? c->prep : nullptr;
      if (last) launch_dense<64, 64, 32, true>(p, s);
      else { launch_small<true>(p, pa, s); if (N.spectral) c->prep_valid = pa.out != nullptr; }
    } else {
      p.X = N.hid[(l - 2) & 1]; p.ldx = N.ld_hid;
      PrepArgs pa{};
      if (!last) launch_small<false>(p, pa, s);
      else if (!N.spectral) launch_dense<64, 64, 32, false>(p, s);
      else if (out_tile_choice() == 9 && c->dma_ok && c->a_planes && c->w_planes && c->ld_hid >= c->wp_Kp) launch_out_bx3dma(c, p, s);
      else if ((out_tile_choice() == 8 || out_tile_choice() == 9) && c->dma_ok && c->ld_hid >= c->w_out_kp) launch_out_dma(c, p, s);
      else if (out_tile_choice() == 7 && c->w_planes) launch_out_bf16x3(c, p, s);
      else if (out_tile_choice() == 6 && p.K <= OK_KMAX) launch_out_resident(p, s);
      else switch (out_tile_choice()) {
        case 1: launch_dense<128, 64, 32, false>(p, s); break;
        case 2: launch_dense<64, 128, 32, false>(p, s); break;
        case 3: launch_dense<64, 64, 64, false>(p, s); break;
        case 4: launch_dense<128, 64, 64, false>(p, s); break;
        default: launch_dense<64, 64, 32, false>(p, s); break;    // 5 (or K too large for the resident form)
      }
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, PAYNE_E_HIP, std::string("dense launch: ") + hipGetErrorString(e));
  return PAYNE_OK;
}

// ---- continuum network ------------------------------------------------------------------
// ystpred.py:191-209 for one candidate per workgroup: F_nu -> F_lambda, normalise by the NaN-ignoring
// median, interpolate onto the spectral ANN grid (NaN outside), multiply into the spectrum.
// The median is the middle of a bitonic sort in LDS (fp64 keys, NaN -> +inf); the spectrum row is stored
// shifted by -1, so (m C) - 1 = (m - 1) C + (C - 1).
__global__ void __launch_bounds__(256) payne_cont_kernel(const float* __restrict__ cont, int npc, int npc2,
                                                          const double* __restrict__ scale, const int* __restrict__ idx,
                                                          const double* __restrict__ frac, float* raw, int npix) {
  extern __shared__ __attribute__((aligned(16))) double cs[];       // [npc2] sort buffer
  __shared__ double med_s;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* row = cont + (size_t)b * npc;
  int nv = 0;
  for (int i = tid; i < npc2; i += 256) {
    double q = INFINITY;
    if (i < npc) { q = (double)row[i] * scale[i]; if (q != q) q = INFINITY; else ++nv; }
    cs[i] = q;
  }
  // number of non-NaN values: wave ballot sums through LDS would do; npc is small, use an LDS atomic
  __shared__ int nvalid;
  if (tid == 0) nvalid = 0;
  __syncthreads();
  atomicAdd(&nvalid, nv);
  for (int k = 2; k <= npc2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int i = tid; i < npc2; i += 256) {
        const int p = i ^ j;
        if (p > i) {
          const double a = cs[i], c2 = cs[p];
          const bool up = (i & k) == 0;
          if ((a > c2) == up) { cs[i] = c2; cs[p] = a; }
        }
      }
    }
  __syncthreads();
  if (tid == 0) {
    const int n = nvalid;                                            // np.nanmedian: mean of the two middle values
    med_s = n == 0 ? __builtin_nan("") : ((n & 1) ? cs[n >> 1] : 0.5 * (cs[(n >> 1) - 1] + cs[n >> 1]));
  }
  __syncthreads();
  const double med = med_s;
  float* r = raw + (size_t)b * npix;
  for (int i = tid; i < npix; i += 256) {
    const int k = idx[i];
    double C = __builtin_nan("");
    if (k >= 0) {
      const double q0 = (double)row[k] * scale[k], q1 = (double)row[k + 1] * scale[k + 1];
      C = (q0 + frac[i] * (q1 - q0)) / med;
    }
    const float m1 = r[i];
    r[i] = (float)((double)m1 * C + (C - 1.0));
  }
}

// `instr_factor`: what Inst_R is multiplied by (2.355 in the likelihood / genspec, 1 in getspec): the
// first-layer launch also writes the post kernel's per-candidate records (c->prep) for that factor.
static int run_ann(payne_ctx* c, const double* theta, int B, double instr_factor, hipStream_t s, bool with_cont = true) {
  c->prep_valid = false;
  NetRef N{c->layers, c->n_layers, c->n_labels, c->xmin, c->xden, c->hid, c->ld_hid, c->raw, c->T.npix, kBase, true};
  int rc = run_net(c, N, theta, B, instr_factor, s);
  if (rc || !c->has_cont || !with_cont) return rc;
  NetRef C{c->clayers, c->cn_layers, c->n_labels, c->cxmin, c->cxden, c->chid, c->cn_ld_hid, c->cont_raw, c->cn_npix, 0.f, false};
  if ((rc = run_net(c, C, theta, B, instr_factor, s))) return rc;
  int npc2 = 1;
  while (npc2 < c->cn_npix) npc2 <<= 1;
  PAYNE_LAUNCH(payne_cont_kernel, dim3(B), dim3(256), (size_t)npc2 * 8, s, c->cont_raw, c->cn_npix, npc2, c->cont_scale,
                     c->cont_idx, c->cont_frac, c->raw, c->T.npix);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, PAYNE_E_HIP, std::string("continuum launch: ") + hipGetErrorString(e));
  return PAYNE_OK;
}

static int check_call(payne_ctx* c, const void* in, int B, const void* out) {
  if (!c) return PAYNE_E_INVALID;
  if (!in || !out) return fail(c, PAYNE_E_INVALID, "NULL input/output pointer");
  if (B <= 0) return fail(c, PAYNE_E_INVALID, "B must be > 0");
  if (B > c->opts.b_max) return fail(c, PAYNE_E_BATCH, "B exceeds opts.b_max");
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev != c->device) {
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return fail(c, PAYNE_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  }
  return PAYNE_OK;
}

static int run_sed(payne_ctx* c, const double* in, int ld, int mode, int B, double* mags, hipStream_t s) {
  {
    ProfScope ps(c, s, 2);
    PAYNE_LAUNCH(payne_sed_kernel, dim3(c->P.F, B), dim3(64), (size_t)c->P.H * 16, s, c->P, in, ld, mode,
                       8 + c->opts.npoly, c->opts.photscale, mags);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, PAYNE_E_HIP, std::string("sed launch: ") + hipGetErrorString(e));
  return PAYNE_OK;
}

static int run_post_lsf(payne_ctx* c, const double* theta, int B, int stage, float* out, int ld_out, double* lnl,
                        bool with_phot, hipStream_t s);

static int run_post(payne_ctx* c, const double* theta, int B, double instr_factor, int stage, float* out, int ld_out,
                    double* lnl, bool with_phot, hipStream_t s) {
  if (c->has_lsf && (stage < 0 || stage == 2 || stage == 3)) return run_post_lsf(c, theta, B, stage, out, ld_out, lnl, with_phot, s);
  PostArgs a{};
  a.theta = theta; a.ld_theta = c->ncols; a.instr_factor = instr_factor;
  a.raw = c->raw; a.ld_raw = c->T.npix;
  a.out = out; a.ld_out = ld_out; a.out_stage = stage; a.lnl = lnl;
  if (with_phot) { a.mags = c->mags_ws; a.n_filters = c->P.F; a.obs_mag = c->obs_mag; a.obs_err = c->obs_err; }
  a.prep = c->prep_valid ? c->prep : nullptr;
  if (skip_mask() & 4) return PAYNE_OK;
  {
    ProfScope ps(c, s, 1);
    if (c->big_ws) {
      const int grid = B < c->big_grid ? B : c->big_grid;
      PAYNE_LAUNCH(payne_post_big_kernel, dim3(grid), dim3(kBigThreads), 0, s, c->T, a, c->big_ws, B);
    } else {
      PAYNE_LAUNCH((stage < 0 && !out && a.prep) ? c->post_fn_lean : c->post_fn, dim3(B), dim3(kPostThreads), c->post_lds, s, c->T, a);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, PAYNE_E_HIP, std::string("post launch: ") + hipGetErrorString(e));
  return PAYNE_OK;
}

// LSF path: spectra after rotational broadening (post kernel, stage 5) -> payne_lsf_kernel
static int run_post_lsf(payne_ctx* c, const double* theta, int B, int stage, float* out, int ld_out, double* lnl,
                        bool with_phot, hipStream_t s) {
  if (c->big_ws) return fail(c, PAYNE_E_UNSUPPORTED, "LSF broadening is built for spectra up to 8192 pixels");
  const bool had = c->has_lsf;
  c->has_lsf = false;                                      // (the stage-5 pass below goes through run_post)
  int rc = run_post(c, theta, B, 1.0, 5, c->lsf_spec, c->T.npix, nullptr, false, s);
  c->has_lsf = had;
  if (rc) return rc;
  LsfArgs a{};
  a.theta = theta; a.ld_theta = c->ncols; a.spec = c->lsf_spec; a.ld_spec = c->T.npix;
  a.obs_wave = c->d_obs_wave; a.lsf = c->lsf;
  a.ws = c->lsf_ws; a.ws_stride = 2 * (size_t)c->T.npix + c->T.n1;
  a.out = out; a.ld_out = ld_out; a.out_stage = stage; a.lnl = lnl;
  if (with_phot) { a.mags = c->mags_ws; a.n_filters = c->P.F; a.obs_mag = c->obs_mag; a.obs_err = c->obs_err; }
  const size_t lds = (size_t)c->T.n1 * 8 + 2 * (size_t)fft_buf_floats(c->T.n1) * 4 + (256 + 8) * 8;
  {
    ProfScope ps(c, s, 1);
    PAYNE_LAUNCH(payne_lsf_kernel, dim3(B), dim3(256), lds, s, c->T, a);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, PAYNE_E_HIP, std::string("lsf launch: ") + hipGetErrorString(e));
  return PAYNE_OK;
}

extern "C" int payne_lnlike_batch(payne_ctx* c, const double* theta, int B, double* lnl, void* stream) {
  int rc = check_call(c, theta, B, lnl);
  if (rc) return rc;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (c->has_phot && !c->has_obs_phot) return fail(c, PAYNE_E_INVALID, "photometric model without observed magnitudes");
  if (c->has_model) {
    if (!c->obs_bound || !c->T.obs_f1) return fail(c, PAYNE_E_INVALID, "no observed spectrum (flux, eflux) bound");
    if ((rc = run_ann(c, theta, B, 2.355, s))) return rc;
  }
  if (c->has_phot && (rc = run_sed(c, theta, c->ncols, 1, B, c->mags_ws, s))) return rc;
  if (c->has_model) return run_post(c, theta, B, 2.355, -1, nullptr, 0, lnl, c->has_phot, s);
  hipLaunchKernelGGL(payne_photonly_kernel, dim3((B + 127) / 128), dim3(128), 0, s, c->mags_ws, c->obs_mag, c->obs_err, c->P.F, B, lnl);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, PAYNE_E_HIP, std::string("photonly launch: ") + hipGetErrorString(e));
  return PAYNE_OK;
}

extern "C" int payne_predict_batch(payne_ctx* c, const double* theta, int B, int stage, unsigned flags, float* out,
                                   int ld_out, void* stream) {
  int rc = check_call(c, theta, B, out);
  if (rc) return rc;
  if (!c->has_model) return fail(c, PAYNE_E_INVALID, "context has no spectral model");
  if (stage < 0 || stage > 4) return fail(c, PAYNE_E_INVALID, "stage must be 0..4");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (stage == PAYNE_STAGE_CONT) {                       // predictcont: the continuum network's own output
    if (!c->has_cont) return fail(c, PAYNE_E_INVALID, "no continuum network bound");
    if (ld_out < c->cn_npix) return fail(c, PAYNE_E_INVALID, "ld_out too small");
    NetRef C{c->clayers, c->cn_layers, c->n_labels, c->cxmin, c->cxden, c->chid, c->cn_ld_hid, c->cont_raw, c->cn_npix, 0.f, false};
    if ((rc = run_net(c, C, theta, B, 1.0, s))) return rc;
    hipError_t he = hipMemcpy2DAsync(out, (size_t)ld_out * 4, c->cont_raw, (size_t)c->cn_npix * 4, (size_t)c->cn_npix * 4, B,
                                     hipMemcpyDeviceToDevice, s);
    if (he != hipSuccess) return fail(c, PAYNE_E_HIP, std::string("hipMemcpy2DAsync: ") + hipGetErrorString(he));
    return PAYNE_OK;
  }
  if (stage >= 2 && !c->obs_bound) return fail(c, PAYNE_E_INVALID, "no observed grid bound");
  if (ld_out < (stage >= 2 ? c->T.nobs : c->T.npix)) return fail(c, PAYNE_E_INVALID, "ld_out too small");
  if ((rc = run_ann(c, theta, B, (flags & PAYNE_F_FWHM_R) ? 2.355 : 1.0, s, stage != 0))) return rc;   // stage 0 = predictspec: no continuum
  return run_post(c, theta, B, (flags & PAYNE_F_FWHM_R) ? 2.355 : 1.0, stage, out, ld_out, nullptr, false, s);
}

// smoothspec on caller-supplied spectra (PayneSpecPredict.smoothspec, ystpred.py:279-281 -> utils.smoothing.smoothspec):
// the same stages as payne_predict_batch with the ANN forward pass replaced by `spectra` (full flux on the context's
// model grid).  stage 1: rotational broadening on the model grid; stage 2/3: ... and instrumental broadening onto
// the bound observed grid.
__global__ void payne_shift_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ out, int npix, int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)B * npix) { const size_t b = i / npix, p = i - b * npix; out[i] = in[b * ld_in + p] - kBase; }
}
extern "C" int payne_smooth_batch(payne_ctx* c, const float* spectra, int ld_spec, const double* theta, int B, int stage,
                                  unsigned flags, float* out, int ld_out, void* stream) {
  int rc = check_call(c, theta, B, out);
  if (rc) return rc;
  if (!spectra) return fail(c, PAYNE_E_INVALID, "spectra is NULL");
  if (!c->has_model) return fail(c, PAYNE_E_INVALID, "context has no spectral model");
  if (stage < 1 || stage > 3) return fail(c, PAYNE_E_INVALID, "stage must be 1..3");
  if (ld_spec < c->T.npix) return fail(c, PAYNE_E_INVALID, "ld_spec too small");
  if (stage >= 2 && !c->obs_bound) return fail(c, PAYNE_E_INVALID, "no observed grid bound");
  if (ld_out < (stage >= 2 ? c->T.nobs : c->T.npix)) return fail(c, PAYNE_E_INVALID, "ld_out too small");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t n = (size_t)B * c->T.npix;
  hipLaunchKernelGGL(payne_shift_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, spectra, ld_spec, c->raw, c->T.npix, B);
  c->prep_valid = false;                                   // no dense launch wrote records for these rows
  // stage 1 here is smoothspec('vsini') itself: getspec's edge rule (ystpred.py:223-224) is not part of it
  return run_post(c, theta, B, (flags & PAYNE_F_FWHM_R) ? 2.355 : 1.0, stage == 1 ? 6 : stage, out, ld_out, nullptr, false, s);
}

extern "C" int payne_sed_batch(payne_ctx* c, const double* pars, int B, double* mags, void* stream) {
  int rc = check_call(c, pars, B, mags);
  if (rc) return rc;
  if (!c->has_phot) return fail(c, PAYNE_E_INVALID, "context has no photometric model");
  return run_sed(c, pars, 9, 0, B, mags, reinterpret_cast<hipStream_t>(stream));
}


// ============================================================================
// device-side sampler step: prior transform, ln-prior, theta rows, random-walk proposals
// ============================================================================
struct SamplerDev {
  int ndim, ncols, nfixed;
  payne_prior_dim dims[PAYNE_MAX_DIM];
  int fixed_col[PAYNE_MAX_FIXED];
  double fixed_val[PAYNE_MAX_FIXED];
};

// unit cube -> parameter (Payne/fitting/prior.py:151-178, scipy.stats ppf's restated)
__device__ double prior_ppf(const payne_prior_dim& d, double u) {
  switch (d.kind) {
    case PAYNE_PRIOR_UNIFORM: {
      const double lo = fmin(d.p[0], d.p[1]), hi = fmax(d.p[0], d.p[1]);
      return (hi - lo) * u + lo;
    }
    case PAYNE_PRIOR_GAUSSIAN: return d.p[0] + d.p[1] * normcdfinv(u);
    case PAYNE_PRIOR_TGAUSSIAN: {
      const double a = (d.p[0] - d.p[2]) / d.p[3], b = (d.p[1] - d.p[2]) / d.p[3];
      double x;
      if (a > 0.0) {                    // both limits in the upper tail: work with survival functions
        const double sa = normcdf(-a), sb = normcdf(-b);
        x = -normcdfinv(sa - u * (sa - sb));
      } else {
        const double ca = normcdf(a), cb = normcdf(b);
        x = normcdfinv(ca + u * (cb - ca));
      }
      double v = d.p[2] + d.p[3] * x;
      if (!(v <= d.p[1])) v = (v != v) ? v : d.p[1];           // +inf (u = 1) -> hi, prior.py:165-166
      return v;
    }
    case PAYNE_PRIOR_EXP: return d.p[0] - d.p[1] * log1p(-u);
    case PAYNE_PRIOR_TEXP: {
      const double b = (d.p[1] - d.p[0]) / d.p[2];
      double v = d.p[0] - d.p[2] * log1p(u * expm1(-b));        // truncexpon.ppf
      if (!(v <= d.p[1])) v = (v != v) ? v : d.p[1];
      return v;
    }
    case PAYNE_PRIOR_LOGUNIFORM: return exp(log(d.p[0]) + u * (log(d.p[1]) - log(d.p[0])));
    default: return u;
  }
}
__device__ double prior_ln(const payne_prior_dim& d, double v) {
  double lp = 0.0;
  if (d.has_gauss) { const double z = v - d.g_mu; lp += -0.5 * ((z * z) / (d.g_sigma * d.g_sigma)); }
  if (d.has_box && ((v < d.box_lo) || (v > d.box_hi))) lp = -INFINITY;
  return lp;
}
// one theta row: NaN = absent, fixed values, then the sampled dimensions
__device__ void write_theta_row(const SamplerDev& sd, const double* v, double* row) {
  for (int c = 0; c < sd.ncols; ++c) row[c] = __builtin_nan("");
  for (int i = 0; i < sd.nfixed; ++i) row[sd.fixed_col[i]] = sd.fixed_val[i];
  for (int d = 0; d < sd.ndim; ++d) if (sd.dims[d].theta_col >= 0) row[sd.dims[d].theta_col] = v[d];
}

// counter-based generator: splitmix64 of (seed, chain, step, draw)
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ double u01(unsigned long long seed, unsigned chain, unsigned step, unsigned draw) {
  const unsigned long long x = mix64(mix64(seed ^ ((unsigned long long)chain << 32 | step)) + draw);
  return ((double)(x >> 11) + 0.5) * (1.0 / 9007199254740992.0);      // (0,1)
}

// transform only (mode 0) or transform + ln-prior + theta row (mode 1)
__global__ void payne_prior_kernel(SamplerDev sd, const double* u, int K, double* v, double* lnprior, double* rows, int mode) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= K) return;
  double vv[PAYNE_MAX_DIM];
  double lp = 0.0;
  for (int d = 0; d < sd.ndim; ++d) {
    vv[d] = prior_ppf(sd.dims[d], u[(size_t)c * sd.ndim + d]);
    v[(size_t)c * sd.ndim + d] = vv[d];
    lp += prior_ln(sd.dims[d], vv[d]);
  }
  if (mode) { lnprior[c] = lp; write_theta_row(sd, vv, rows + (size_t)c * sd.ncols); }
}
// lnprob = lnprior + lnlike (-inf prior wins; NaN likelihood stays NaN)
__global__ void payne_lnprob_kernel(const double* lnprior, const double* lnl, int K, double* out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < K) out[c] = (lnprior[c] == -INFINITY) ? -INFINITY : lnprior[c] + lnl[c];
}

// One random-walk step for every chain: first settle the previous proposal (accept iff inside the
// cube and lnprob > loglstar), then draw the next one.  `propose` = 0 on the closing call.
// ONE WAVE PER CHAIN, lane d = sampled dimension d: the inverse CDFs (the expensive part: normcdf /
// normcdfinv chains in fp64) of the dimensions run side by side, the ellipsoid step is a shuffle
// matvec, sums are wave reductions.  (One thread per chain spent 14 us per step in a ~3000-instruction
// dependent fp64 chain; the step sits between two likelihood batches, nothing overlaps it.)
__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}
__global__ void __launch_bounds__(256) payne_rwalk_kernel(SamplerDev sd, int K, double* u, double* v, double* lnprob, int* nacc, int* ncall,
                                   double* u_prop, double* v_prop, double* lnprior_prop, int* inside,
                                   const double* lnl_prop, double* rows, const double* axes, const int* ell, double scale,
                                   double loglstar, unsigned long long seed, int step, int settle, int propose) {
  const int lane = threadIdx.x & 63;
  const int c = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (c >= K) return;                                           // the whole wave leaves together
  const int nd = sd.ndim;
  const bool act = lane < nd;
  const int dl = act ? lane : 0;
  const size_t off = (size_t)c * nd + dl;
  double uc = u[off];
  if (settle && inside[c]) {
    const double lpr = lnprior_prop[c];
    const double lp = (lpr == -INFINITY) ? -INFINITY : lpr + lnl_prop[c];
    const bool accept = lp > loglstar;                          // false for NaN
    if (accept && act) { uc = u_prop[off]; u[off] = uc; v[off] = v_prop[off]; }
    if (lane == 0) {
      ncall[c] += 1;
      if (accept) { lnprob[c] = lp; nacc[c] += 1; }
    }
  }
  if (!propose) return;
  // z uniform in the unit ball: normal direction (one Box-Muller cosine per lane), radius U^(1/n)
  double z = 0.0;
  if (act) {
    const double a = u01(seed, c, step, 2 * lane), b = u01(seed, c, step, 2 * lane + 1);
    z = sqrt(-2.0 * log(a)) * cos(6.283185307179586 * b);
  }
  const double n2 = wave_sum(z * z);
  const double rad = pow(u01(seed, c, step, 128), 1.0 / (double)nd) / sqrt(n2);
  const double* ax = axes + (ell ? (size_t)ell[c] * nd * nd : 0);   // this chain's ellipsoid (bound='multi')
  double sdot = 0.0;
  for (int e = 0; e < nd; ++e) {
    const double ze = __shfl(z, e);
    sdot = fma(ax[dl * nd + e], ze, sdot);
  }
  const double up = uc + scale * rad * sdot;
  const bool in = __ballot(act && !((up > 0.0) && (up < 1.0))) == 0ull;
  const payne_prior_dim dim = sd.dims[dl];
  const double vp = in ? prior_ppf(dim, up) : v[off];           // outside: a harmless valid row
  const double lp = wave_sum(act ? prior_ln(dim, vp) : 0.0);
  if (act) { u_prop[off] = up; v_prop[off] = vp; }
  if (lane == 0) { inside[c] = in ? 1 : 0; lnprior_prop[c] = lp; }
  // theta row, lane = column: NaN = absent, fixed values, then the sampled dimensions
  double val = __builtin_nan("");
  for (int i = 0; i < sd.nfixed; ++i) val = (sd.fixed_col[i] == lane) ? sd.fixed_val[i] : val;
  for (int d = 0; d < nd; ++d) {
    const double vd = __shfl(vp, d);
    val = (sd.dims[d].theta_col == lane) ? vd : val;
  }
  if (lane < sd.ncols) rows[(size_t)c * sd.ncols + lane] = val;
}

struct payne_sampler {
  payne_ctx* ctx = nullptr;
  SamplerDev sd{};
  int k_max = 0;
  double *u_prop = nullptr, *v_prop = nullptr, *lnprior = nullptr, *lnl = nullptr, *rows = nullptr, *axes = nullptr;
  int* inside = nullptr;
  int* ell = nullptr;                     // per-chain ellipsoid index of the walk in progress
  std::vector<void*> owned;
  // a walk in progress (payne_rwalk_begin / payne_rwalk_step)
  struct { double *u, *v, *lnprob; int K, walks; double scale, loglstar; unsigned long long seed; int *nacc, *ncall; void* stream; bool open; bool multi; } run{};
};

extern "C" void payne_sampler_destroy(payne_sampler* s) {
  if (!s) return;
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(s->ctx->device);
  for (void* p : s->owned) (void)hipFree(p);
  (void)hipSetDevice(prev);
  delete s;
}

extern "C" int payne_sampler_create(payne_ctx* c, const payne_sampler_desc* d, int k_max, payne_sampler** out) {
  if (!c || !out) return PAYNE_E_INVALID;
  *out = nullptr;
  if (!d || d->ndim <= 0 || d->ndim > PAYNE_MAX_DIM) return fail(c, PAYNE_E_INVALID, "sampler.ndim out of range");
  if (d->n_fixed < 0 || d->n_fixed > PAYNE_MAX_FIXED) return fail(c, PAYNE_E_INVALID, "sampler.n_fixed out of range");
  if (k_max <= 0 || k_max > c->opts.b_max) return fail(c, PAYNE_E_INVALID, "sampler.k_max must be in 1..opts.b_max");
  for (int i = 0; i < d->ndim; ++i)
    if (d->dims[i].theta_col >= c->ncols || d->dims[i].kind < 0 || d->dims[i].kind > PAYNE_PRIOR_LOGUNIFORM)
      return fail(c, PAYNE_E_INVALID, "sampler dimension with bad theta_col / kind");
  for (int i = 0; i < d->n_fixed; ++i)
    if (d->fixed_col[i] < 0 || d->fixed_col[i] >= c->ncols) return fail(c, PAYNE_E_INVALID, "fixed parameter with bad column");
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev != c->device) (void)hipSetDevice(c->device);
  payne_sampler* s = new payne_sampler();
  s->ctx = c; s->k_max = k_max;
  s->sd.ndim = d->ndim; s->sd.ncols = c->ncols; s->sd.nfixed = d->n_fixed;
  for (int i = 0; i < d->ndim; ++i) s->sd.dims[i] = d->dims[i];
  for (int i = 0; i < d->n_fixed; ++i) { s->sd.fixed_col[i] = d->fixed_col[i]; s->sd.fixed_val[i] = d->fixed_val[i]; }
  auto alloc = [&](size_t bytes, void** p) -> int {
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return fail(c, PAYNE_E_HIP, std::string("hipMalloc(sampler): ") + hipGetErrorString(e));
    s->owned.push_back(*p);
    return PAYNE_OK;
  };
  const size_t K = (size_t)k_max, nd = (size_t)d->ndim;
  int rc;
  if ((rc = alloc(K * nd * 8, (void**)&s->u_prop)) || (rc = alloc(K * nd * 8, (void**)&s->v_prop)) ||
      (rc = alloc(K * 8, (void**)&s->lnprior)) || (rc = alloc(K * 8, (void**)&s->lnl)) ||
      (rc = alloc(K * c->ncols * 8, (void**)&s->rows)) || (rc = alloc((size_t)PAYNE_MAX_ELL * nd * nd * 8, (void**)&s->axes)) ||
      (rc = alloc(K * 4, (void**)&s->inside)) || (rc = alloc(K * 4, (void**)&s->ell))) {
    payne_sampler_destroy(s);
    return rc;
  }
  (void)hipMemset(s->inside, 0, K * 4);
  *out = s;
  return PAYNE_OK;
}

static int sampler_check(payne_sampler* s, const void* a, int K, const void* b) {
  if (!s) return PAYNE_E_INVALID;
  payne_ctx* c = s->ctx;
  if (!a || !b) return fail(c, PAYNE_E_INVALID, "NULL input/output pointer");
  if (K <= 0 || K > s->k_max) return fail(c, PAYNE_E_BATCH, "K exceeds the sampler's k_max");
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev != c->device) (void)hipSetDevice(c->device);
  return PAYNE_OK;
}

extern "C" int payne_prior_transform_batch(payne_sampler* s, const double* u, int K, double* v, void* stream) {
  int rc = sampler_check(s, u, K, v);
  if (rc) return rc;
  hipLaunchKernelGGL(payne_prior_kernel, dim3((K + 127) / 128), dim3(128), 0, reinterpret_cast<hipStream_t>(stream), s->sd, u, K,
                     v, (double*)nullptr, (double*)nullptr, 0);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(s->ctx, PAYNE_E_HIP, std::string("prior launch: ") + hipGetErrorString(e));
  return PAYNE_OK;
}

extern "C" int payne_lnprob_u_batch(payne_sampler* s, const double* u, int K, double* v, double* lnprob, void* stream) {
  int rc = sampler_check(s, u, K, v);
  if (rc) return rc;
  if (!lnprob) return fail(s->ctx, PAYNE_E_INVALID, "lnprob is NULL");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(payne_prior_kernel, dim3((K + 127) / 128), dim3(128), 0, st, s->sd, u, K, v, s->lnprior, s->rows, 1);
  if ((rc = payne_lnlike_batch(s->ctx, s->rows, K, s->lnl, stream))) return rc;
  hipLaunchKernelGGL(payne_lnprob_kernel, dim3((K + 127) / 128), dim3(128), 0, st, s->lnprior, s->lnl, K, lnprob);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(s->ctx, PAYNE_E_HIP, std::string("lnprob launch: ") + hipGetErrorString(e));
  return PAYNE_OK;
}

// The walk in two parts, so that a caller can interleave the steps of several samplers (one context and
// one HIP stream each) from one host thread: two independent batches in flight fill the idle time a single
// chain of dependent launches leaves (13.6 M against 10.6 M evaluations/s at 512 x 4096 pixels).
extern "C" int payne_rwalk_begin_ell(payne_sampler* s, double* u, double* v, double* lnprob, int K, const double* axes,
                                     int n_ell, const int* ell, double scale, double loglstar, int walks,
                                     unsigned long long seed, int* nacc, int* ncall, void* stream) {
  int rc = sampler_check(s, u, K, v);
  if (rc) return rc;
  if (!lnprob || !axes || !nacc || !ncall || walks <= 0) return fail(s->ctx, PAYNE_E_INVALID, "bad rwalk arguments");
  if (n_ell < 1 || n_ell > PAYNE_MAX_ELL || (n_ell > 1 && !ell)) return fail(s->ctx, PAYNE_E_INVALID, "bad ellipsoid list");
  if (ell)
    for (int i = 0; i < K; ++i)
      if (ell[i] < 0 || ell[i] >= n_ell) return fail(s->ctx, PAYNE_E_INVALID, "ellipsoid index out of range");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nd = s->sd.ndim;
  HIPCHK(s->ctx, hipMemcpyAsync(s->axes, axes, (size_t)n_ell * nd * nd * 8, hipMemcpyHostToDevice, st));
  if (ell) HIPCHK(s->ctx, hipMemcpyAsync(s->ell, ell, (size_t)K * 4, hipMemcpyHostToDevice, st));
  HIPCHK(s->ctx, hipMemsetAsync(nacc, 0, (size_t)K * 4, st));
  HIPCHK(s->ctx, hipMemsetAsync(ncall, 0, (size_t)K * 4, st));
  s->run = {u, v, lnprob, K, walks, scale, loglstar, seed, nacc, ncall, stream, true, ell != nullptr};
  return PAYNE_OK;
}
extern "C" int payne_rwalk_begin(payne_sampler* s, double* u, double* v, double* lnprob, int K, const double* axes,
                                 double scale, double loglstar, int walks, unsigned long long seed, int* nacc, int* ncall,
                                 void* stream) {
  return payne_rwalk_begin_ell(s, u, v, lnprob, K, axes, 1, nullptr, scale, loglstar, walks, seed, nacc, ncall, stream);
}
// step w = 0 .. walks: settle proposal w-1, draw proposal w and evaluate it (the last step only settles)
extern "C" int payne_rwalk_step(payne_sampler* s, int w) {
  if (!s || !s->ctx) return PAYNE_E_INVALID;
  if (!s->run.open || w < 0 || w > s->run.walks) return fail(s->ctx, PAYNE_E_INVALID, "payne_rwalk_step outside a walk");
  const auto& r = s->run;
  hipStream_t st = reinterpret_cast<hipStream_t>(r.stream);
  const dim3 grid((r.K + 3) / 4), block(256);                  // one wave per chain
  hipLaunchKernelGGL(payne_rwalk_kernel, grid, block, 0, st, s->sd, r.K, r.u, r.v, r.lnprob, r.nacc, r.ncall, s->u_prop,
                     s->v_prop, s->lnprior, s->inside, s->lnl, s->rows, s->axes, r.multi ? s->ell : (const int*)nullptr, r.scale,
                     r.loglstar, r.seed, w,
                     w > 0 ? 1 : 0, w < r.walks ? 1 : 0);
  int rc = PAYNE_OK;
  if (w < r.walks) rc = payne_lnlike_batch(s->ctx, s->rows, r.K, s->lnl, r.stream);
  else s->run.open = false;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(s->ctx, PAYNE_E_HIP, std::string("rwalk launch: ") + hipGetErrorString(e));
  return rc;
}
extern "C" int payne_rwalk_batch(payne_sampler* s, double* u, double* v, double* lnprob, int K, const double* axes,
                                 double scale, double loglstar, int walks, unsigned long long seed, int* nacc, int* ncall,
                                 void* stream) {
  int rc = payne_rwalk_begin(s, u, v, lnprob, K, axes, scale, loglstar, walks, seed, nacc, ncall, stream);
  for (int w = 0; !rc && w <= walks; ++w) rc = payne_rwalk_step(s, w);
  return rc;
}

extern "C" int payne_bc_batch(payne_ctx* c, const double* x, int B, double* bc, void* stream) {
  int rc = check_call(c, x, B, bc);
  if (rc) return rc;
  if (!c->has_phot) return fail(c, PAYNE_E_INVALID, "context has no photometric model");
  return run_sed(c, x, 6, 2, B, bc, reinterpret_cast<hipStream_t>(stream));
}

// ---- per-kernel timing ---------------------------------------------------------
extern "C" int payne_profile(payne_ctx* c, int enable) {
  if (!c) return PAYNE_E_INVALID;
  c->prof = enable != 0;
  if (enable) {
    c->prof_used = 0;
    for (int k = 0; k < 4; ++k) { c->prof_ms[k] = 0.0; c->prof_n[k] = 0; }
  }
  return PAYNE_OK;
}

extern "C" int payne_profile_read(payne_ctx* c, int kind, double* total_ms, long long* launches) {
  if (!c || kind < 0 || kind > 3) return PAYNE_E_INVALID;
  for (size_t i = 0; i < c->prof_used; ++i) {
    auto& r = c->prof_pool[i];
    float ms = 0.f;
    HIPCHK(c, hipEventSynchronize(r.e1));
    HIPCHK(c, hipEventElapsedTime(&ms, r.e0, r.e1));
    c->prof_ms[r.kind] += ms;
    c->prof_n[r.kind] += 1;
  }
  c->prof_used = 0;
  if (total_ms) *total_ms = c->prof_ms[kind];
  if (launches) *launches = c->prof_n[kind];
  return PAYNE_OK;
}

#ifdef PAYNE_STAMPS
// Diagnostic build only: cycle stamps of the first hidden-layer launch ([grid][16], slot 15 = grid size).
extern "C" int payne_diag_hidden_stamps(payne_ctx* c, const double* theta, int B, unsigned long long* host, int max_blocks) {
  int rc = check_call(c, theta, B, host);
  if (rc) return rc;
  unsigned long long* d = nullptr;
  const size_t nb_ = (size_t)(max_blocks < 0 ? -max_blocks : max_blocks);
  HIPCHK(c, hipMalloc(&d, nb_ * 16 * 8));
  HIPCHK(c, hipMemset(d, 0, nb_ * 16 * 8));
  if (max_blocks < 0) { max_blocks = -max_blocks; g_dense_stamps = d; } else g_hidden_stamps = d;   // negative: the output layer
  rc = run_ann(c, theta, B, 2.355, nullptr);
  g_hidden_stamps = nullptr; g_dense_stamps = nullptr;
  HIPCHK(c, hipDeviceSynchronize());
  HIPCHK(c, hipMemcpy(host, d, (size_t)max_blocks * 16 * 8, hipMemcpyDeviceToHost));
  (void)hipFree(d);
  return rc;
}
// Diagnostic build only: one lnlike batch with per-phase cycle stamps of the post kernel.
// stamps: host [B][64] (slot 0 = number of stamps, slots 1.. = s_memtime after each barrier).
extern "C" int payne_diag_post_stamps(payne_ctx* c, const double* theta, int B, unsigned long long* stamps_host) {
  int rc = check_call(c, theta, B, stamps_host);
  if (rc) return rc;
  unsigned long long* d = nullptr;
  double* lnl = nullptr;
  HIPCHK(c, hipMalloc(&d, (size_t)B * 64 * 8));
  HIPCHK(c, hipMalloc(&lnl, (size_t)B * 8));
  HIPCHK(c, hipMemset(d, 0, (size_t)B * 64 * 8));
  if ((rc = run_ann(c, theta, B, 2.355, nullptr))) return rc;
  PostArgs a{};
  a.theta = theta; a.ld_theta = c->ncols; a.instr_factor = 2.355; a.raw = c->raw; a.ld_raw = c->T.npix;
  a.out_stage = -1; a.lnl = lnl; a.stamps = d; a.prep = c->prep_valid ? c->prep : nullptr;
  hipLaunchKernelGGL(c->post_fn, dim3(B), dim3(kPostThreads), c->post_lds, nullptr, c->T, a);
  HIPCHK(c, hipDeviceSynchronize());
  HIPCHK(c, hipMemcpy(stamps_host, d, (size_t)B * 64 * 8, hipMemcpyDeviceToHost));
  (void)hipFree(d); (void)hipFree(lnl);
  return PAYNE_OK;
}
#endif
