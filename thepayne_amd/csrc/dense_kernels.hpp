// dense_kernels.hpp -- the ANN's dense layers on the matrix cores (device code only; included once by
// payne_hip.hip).  Net.eval / ANN.eval of the reference: Payne/predict/ystpred.py:41-58,
// Payne/train/NNmodels.py:92-168.  The output layer ([B x H] . [H x Npix], 78 % of the path's FLOPs) is
// payne_dense_dma_kernel (LDS-DMA ring + MFMA f32), the hidden layers payne_dense_hidden_kernel.
#pragma once

#include "sed_core.hpp"
#include "sampler_core.hpp"

// ============================================================================
// dense layer on the matrix cores
// ============================================================================
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));   // register-resident 16-byte value (HIP's f32x4_t struct arrays end up in scratch)

struct DenseParams {
  const float* X; int ldx;     // [B][ldx] activations (ignored with FUSE_L0)
  const float* W; int K;       // [N][K] row-major, K % 4 == 0
  // payne_dense_hidden_kernel<false>: a copy of W with rows ldwd >= HK_PITCH floats apart, zero beyond K -- with X's pitch the same
  // and its pad columns zero, both operand tiles go from global memory straight into LDS (hk_tile; null: staged through registers)
  const float* Wd; int ldwd;
  int dma_tiles;               // (hidden-layer kernel, first launch: Wd is set -- handed over as a preloaded argument's bit)
  int k_real;                  // LDS-DMA kernel: width before zero padding (0 = K): the last k-step stops there
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
  // 3 x bf16 planes (payne_dense_dma3_kernel): the hidden-layer kernel ALSO writes its output split in three (Yp, row pitch
  // ldp elements, planes plane_y elements apart); the output layer reads activations Xp and weights Wp [3][N][K] that way
  unsigned short* Yp; const unsigned short* Xp; const unsigned short* Wp;
  int ldp; size_t plane_y, plane_x, plane_w;
  double xmin[PAYNE_MAX_LABELS], xden[PAYNE_MAX_LABELS];
  // payne_dense_dma3_kernel: a second output layer the launch switches to when *sel == sel_seq (null: never) -- the pixel weights for
  // a batch in which some candidate does not rotate, where the launch was made for rows in the frequency domain of a model grid that
  // the rotation stage resamples (run_ann in payne_hip.hip; the hidden-layer launch's records set the word)
  const unsigned long long* sel; unsigned long long sel_seq;
  const unsigned short* Wp_alt; size_t plane_w_alt; const float* bias_alt; float bias_shift_alt; int N_alt, ldy_alt;
  const float* W_alt;          // payne_dense_dma3f_kernel: the alternative layer's weights as fp32 [N_alt][K]
  // payne_dense_dma2h_kernel (two fp16 planes an operand, three products): the hidden-layer kernel writes Yp as TWO fp16 planes of
  // y * yp_scale when yp_half is set; the weights' planes hold W[n][k] * 2^e[n] and rscale[n] = 2^-e[n] / yp_scale undoes both
  int yp_half; float yp_scale;
  const float* rscale; const float* rscale_alt;
  // hk_tile_h2 (first hidden-layer launch on fp16 pairs): the second layer's weights as two fp16 planes [2][N][304] (plane_wh elements
  // apart, row n scaled by 2^e[n]), rs1[n] = 2^-e[n] / a0_scale, a0_scale: the power of two the first layer's output is split with
  const unsigned short* Wh; size_t plane_wh; const float* rs1; float a0_scale; int h2_tiles;
#ifdef PAYNE_STAMPS
  unsigned long long* stamps;  // diagnostic build: [grid][16] cycle stamps of the hidden-layer kernel
#endif
};
// leading scalar parameters of payne_dense_dma3_kernel (14 dwords: all the hardware preloads) and the launch's values for them
// ... of payne_dense_hidden_kernel (14 dwords).  p0: theta (FUSE_L0) | X;  p1: W0 | Wd;  p2: b0 | -;  i2: ld_theta + n_labels << 16 | ldx + ldwd << 16
#define PAYNE_HK_LEAD_PARAMS const void* lead_p0, const float* lead_p1, const float* lead_p2, const float* lead_bias, unsigned lead_i0, unsigned lead_i1, \
                             unsigned lead_i2, int lead_B, unsigned lead_i4, int lead_N
#define PAYNE_HK_LEAD_TYPES const void*, const float*, const float*, const float*, unsigned, unsigned, unsigned, int, unsigned, int
#define PAYNE_D3_LEAD_PARAMS const unsigned long long* lead_sel, const unsigned short* lead_Xp, const unsigned short* lead_Wp, unsigned lead_plane_x, \
                             unsigned lead_plane_w, unsigned lead_grid, int lead_N, int lead_B, int lead_ldp, int lead_K, unsigned lead_sel_seq
#define PAYNE_D3_LEAD_TYPES const unsigned long long*, const unsigned short*, const unsigned short*, unsigned, unsigned, unsigned, int, int, int, int, unsigned
#define PAYNE_D3_LEAD_ARGS(p) (p).sel, (p).Xp, (p).Wp, (unsigned)(p).plane_x, (unsigned)(p).plane_w, ((unsigned)(p).grid_m | ((unsigned)(p).grid_n << 16)), \
                              (p).N, (p).B, (p).ldp, (p).K, (unsigned)(p).sel_seq
#ifdef PAYNE_STAMPS
#ifdef PAYNE_STAMPS_ENDS_ONLY   /* stamps 0 / 5 / 15 only: the phases in between keep their production shape */
#define HK_STAMP(k) do { if (((k) == 0 || (k) == 5 || (k) == 15) && p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HK_STAMP(k) do { if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
static unsigned long long* g_hidden_stamps = nullptr;
static unsigned long long* g_dense_stamps = nullptr;
#else
#define HK_STAMP(k) do {} while (0)
#endif

// Activations without per-lane branches: a `z > 0 ? .. : ..` chain compiles to exec-mask branches, and
// an unrolled epilogue then serialises on them (the fused first layer spent 11 500 of 22 500 cycles
// that way).  leaky ReLU(0.01) = max(z, 0.01 z) for every finite z, 0 and NaN (ystpred.py:57-58).
__device__ __forceinline__ float lrelu01(float z) { return fmaxf(z, 0.01f * z); }
// sigmoid(z) = 1 / (1 + exp(-z)) (LinNet, NNmodels.py:92-121) without the library's expf and the IEEE quotient: those are ~35
// instructions and a chain of exec-mask branches a value, and a lane of the first launch makes forty of them -- 7 000 of the 17 600
// cycles of a LinNet tile.  exp(t) = 2^n v_exp_f32(f), n = rint(t log2 e), f = t log2 e - n by two fmas (log2 e in two parts: f is good
// to 2^-25 whatever t), the quotient by v_rcp_f32 + one Newton step.  Within 2.5 ulp of the exact value for every finite z, exactly 1 / 0
// at +-inf (and past |z| = 88, where the exact value is a denormal), NaN kept.  Ten instructions, no branch.
__device__ __forceinline__ float sigmoid_f32(float z) {
  const float t = __builtin_amdgcn_fmed3f(-z, -88.0f, 88.0f);
  const float n = __builtin_rintf(t * 1.44269502f);
  float f = __builtin_fmaf(t, 1.44269502f, -n);
  f = __builtin_fmaf(t, 1.92596299e-8f, f);
  const float e = __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
  const float d = 1.0f + e;
  float r = __builtin_amdgcn_rcpf(d);
  r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
  return z != z ? z : r;
}
__device__ __forceinline__ float act_apply(float z, int act) {
  if (act == PAYNE_ACT_SIGMOID) return sigmoid_f32(z);                     // (uniform: a scalar branch)
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
// the operand tiles go from global memory straight into a 4-stage LDS ring with
// global_load_lds_dwordx4 (no VGPR staging, no address-clamp/select VALU work, no LDS store
// instructions), requested THREE k-steps ahead and waited for with explicit vmcnt counts.  The
// register-staged kernel cannot keep loads in flight across its barriers (measured: a k-step
// that issues loads takes 2750 cycles, one that does not 1550).
//   * A lane's 16 bytes land at (wave-uniform base) + 16*lane, so padding rows is impossible; bank
//     conflicts of the fragment reads are avoided by an XOR swizzle instead: 16-byte chunk c of
//     tile row r sits at chunk c ^ ((r >> 1) & 7) -- the lane simply FETCHES the chunk that belongs
//     in its slot.
//   * nothing can be masked on the way, so both operands must be zero-padded in k to a multiple
//     of 32 in memory (X: the hidden buffers' pitch; W: ctx->w_out_pad) and rows are clamped.
// ----------------------------------------------------------------------------
// ring stages NS (NS - 1 requested ahead): at WN = 4 four of 32 columns = 96 KB -- one workgroup per CU, for launches of at
// most one tile per CU (C2: 256 tiles) -- or three = 72 KB, two workgroups per CU, which cover each other's first loads and
// last stores when a CU works through many tiles (C5: 16 384); three of 64 columns = 144 KB.
// WN = wave columns: tile = 64 x (32 WN), 2 WN waves.  WN = 2 is the 64 x 64 / 256-thread form (two workgroups
// per CU); WN = 4 the 64 x 128 / 512-thread form (one per CU, same waves per SIMD): the activation tile is then
// fetched once per 128 columns, 24 KB instead of 2 x 16 KB per k-step and CU -- the kernel is bound by the CU's
// miss throughput, not by the matrix pipes.
// BK = k-depth of a stage: 32 (rows of 128 B, 8 chunks, swizzle by (r >> 1) & 7) or 64 (rows of 256 B = one full
// bank cycle, 16 chunks, swizzle by r & 15): half as many barrier steps for the same bytes.
template <int WN, int BK> constexpr int dm_stage_floats() { return (64 + 32 * WN) * BK; }
template <int WN, int BK, int NS> constexpr size_t dm_lds_bytes() { return (size_t)NS * dm_stage_floats<WN, BK>() * sizeof(float); }

// NK: the number of k-steps when known at compile time (10 = the 300-wide hidden layer of the usual nets at BK = 32), 0 =
// read from the launch.  With NK the loop below unrolls and the bookkeeping of the ring folds away: 15.6 us against
// 17.0 us for the rolled form of the same statements (C2).
// PIPE: the schedule above (one tile per CU).  !PIPE: request, then the fragments of the SAME step, a rolled loop, three
// stages -- what a CU that works through many tiles with two resident workgroups runs fastest (C5: 792 us against 866 us;
// the other workgroup's matrix instructions fill the wait, and the leaner loop leaves it more issue slots).
template <int WN, int BK, int NK, int NS, bool PIPE>
__global__ void __launch_bounds__(128 * WN) payne_dense_dma_kernel(DenseParams p) {
  constexpr int BN = 32 * WN, NW = 2 * WN;                 // tile columns, waves
  constexpr int DM_NS = NS, AHEAD = DM_NS - 1;             // ring stages; stages requested beyond the one being consumed
  static_assert(NS == 3 || NS == 4, "ring depth");
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

  const int nk = NK > 0 ? NK : p.K / BK;                   // padded: exact
  const int k_tail = (p.k_real > 0 ? p.k_real : p.K) - (nk - 1) * BK;
  const int kk_last = __builtin_amdgcn_readfirstlane(k_tail >= BK ? BK / 8 : (k_tail <= 0 ? 1 : (k_tail + 7) / 8));
  // (the epilogue's bias, requested before the first transfer: asked for at the end it is a memory round trip of its own)
  const int col = n0 + wn0 + (lane & 31);
  const float bv = p.bias[col < p.N ? col : p.N - 1] - p.bias_shift;
  HK_STAMP(0);
  // Three stages are requested ahead; the fragments of step it+1 are read from LDS BEFORE the matrix instructions of
  // step it are issued, so the LDS round trip of a step (8 ds_read_b128 per wave, ~300 cycles during which neither wave of
  // a SIMD had anything for the matrix pipe: both had just left the same barrier) runs under the 1024 cycles of the
  // previous step's v_mfma chain.
  auto wait_landed = [&](int younger) {                    // my pieces of a stage have landed once only `younger` stages' loads are outstanding
    static_assert(PER == 3 || PER == 4 || PER == 6 || PER == 8, "vmcnt literal");
    if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (younger == 1) {
      if (PER == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (PER == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (PER == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      if (PER == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (PER == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (PER == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
  };
  auto frags = [&](int stage, f32x4_t (&a)[BK / 8], f32x4_t (&b)[BK / 8]) {
    const float* Asb = dm_sm + stage * STAGE;
    const float* Bsb = Asb + 64 * BK;
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      const int c = 2 * kk + half;
      a[kk] = *reinterpret_cast<const f32x4_t*>(Asb + Ra * BK + 4 * (c ^ sa));
      b[kk] = *reinterpret_cast<const f32x4_t*>(Bsb + Rb * BK + 4 * (c ^ sb));
    }
  };
  if constexpr (!PIPE) {
    static_assert(PIPE || (NS == 3 && NK == 0), "the plain schedule: three stages, run-time step count");
    issue(0, 0);
    if (nk > 1) issue(1, BK);
    for (int it = 0; it < nk; ++it) {
      wait_landed(it + 1 < nk ? 1 : 0);                    // my pieces of stage `it` have landed once only stage it+1's loads are outstanding
      asm volatile("s_barrier" ::: "memory");              // everybody's pieces landed; everybody finished step it-1
      if (it < 13) HK_STAMP(1 + it);
      if (it + 2 < nk) issue((it + 2) % DM_NS, (it + 2) * BK);    // into the buffer step it-1 just released
      f32x4_t a[BK / 8], b[BK / 8];                         // all fragments first (one LDS round trip per step)
      frags(it % DM_NS, a, b);
      __builtin_amdgcn_sched_barrier(0);
      const int nkk = (it + 1 == nk) ? kk_last : BK / 8;   // (the zero-padded tail of the last step is skipped)
#pragma unroll
      for (int kk = 0; kk < BK / 8; ++kk) {
        if (kk < nkk) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].x, b[kk].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].y, b[kk].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].z, b[kk].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].w, b[kk].w, acc, 0, 0, 0);
        }
      }
    }
  } else {
  const int npro = nk < AHEAD ? nk : AHEAD;
#pragma unroll
  for (int q = 0; q < AHEAD; ++q)
    if (q < npro) issue(q, q * BK);
  wait_landed(npro - 1);
  asm volatile("s_barrier" ::: "memory");
  f32x4_t a0[BK / 8], b0[BK / 8], a1[BK / 8], b1[BK / 8];
  frags(0, a0, b0);
  // the head of step it (it + 1 < nk): stage it+1 has landed for everybody, the next stage is requested, the fragments
  // of step it+1 are on their way from LDS
  auto head = [&](int it, f32x4_t (&an)[BK / 8], f32x4_t (&bn)[BK / 8], const f32x4_t (&ac)[BK / 8], const f32x4_t (&bc)[BK / 8]) {
    // (the fragments of THIS step were requested a whole step ago: naming them here puts the compiler's LDS wait for
    //  them -- which it can only express as "everything outstanding" around the loop's back edge -- ahead of the next
    //  request instead of between that request and the matrix instructions)
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) asm volatile("" :: "v"(ac[kk]), "v"(bc[kk]));
    {                                                       // requested so far: stages .. min(nk, it + AHEAD) - 1; stage it+1 must have landed
      const int last = (it + AHEAD < nk ? it + AHEAD : nk) - 1;
      wait_landed(last - (it + 1));
    }
    asm volatile("s_barrier" ::: "memory");                // everybody's pieces of stage it+1 landed; everybody finished step it-1
#ifndef PAYNE_NO_LOOP_STAMPS
    if (it < 13) HK_STAMP(1 + it);                       // (diagnostic build; the loop is then fully unrolled)
#endif
    // ORDER MATTERS: a ds_read issued after a global_load_lds of the same wave does not issue until that transfer has
    // landed (the hardware keeps LDS-DMA writes and DS operations of a wave in order), and the matrix instructions queue
    // up behind it: with the request first a step cost one memory latency more (3 340 cycles against 2 380; stamps of
    // the diagnostic build, tools/post_stamps.py).  So: fragments of the next step, THEN the request.
    frags((it + 1) % DM_NS, an, bn);
    __builtin_amdgcn_sched_barrier(0);
    if (it + AHEAD < nk) issue((it + AHEAD) % DM_NS, (it + AHEAD) * BK);     // into the buffer whose fragments step it-1 consumed
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mfma_step = [&](const f32x4_t (&a)[BK / 8], const f32x4_t (&b)[BK / 8]) {
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].x, b[kk].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].y, b[kk].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].z, b[kk].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].w, b[kk].w, acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // the zero-padded tail of the last step (K = 300 -> 320: 20 of its 32 columns) contributes exact zeros: skip those
  // matrix instructions (groups of 8 columns; a uniform branch)
  auto mfma_last = [&](const f32x4_t (&a)[BK / 8], const f32x4_t (&b)[BK / 8]) {
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      if (kk < kk_last) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].x, b[kk].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].y, b[kk].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].z, b[kk].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].w, b[kk].w, acc, 0, 0, 0);
      }
    }
  };
  int it = 0;
#pragma unroll
  for (; it + 2 < nk; it += 2) {                           // two steps per trip: the fragment registers swap roles
    head(it, a1, b1, a0, b0);
    mfma_step(a0, b0);
    head(it + 1, a0, b0, a1, b1);
    mfma_step(a1, b1);
  }
  if (it + 1 < nk) {
    head(it, a1, b1, a0, b0);
    mfma_step(a0, b0);
    mfma_last(a1, b1);
  } else {
    mfma_last(a0, b0);
  }
  }
  // C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  if (col < p.N) {
    const bool act_none = __builtin_amdgcn_readfirstlane(p.act == PAYNE_ACT_NONE ? 1 : 0) != 0;   // (the usual output layer: one test, not sixteen)
    if (act_none) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < p.B) __builtin_nontemporal_store(acc[r] + bv, &p.Y[(size_t)row * p.ldy + col]);   // streamed: next read by other XCDs
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < p.B) __builtin_nontemporal_store(act_apply(acc[r] + bv, p.act), &p.Y[(size_t)row * p.ldy + col]);
      }
    }
  }
  HK_STAMP(15);
}

// ----------------------------------------------------------------------------
// Output layer as SIX bf16 matrix products on the same LDS-DMA ring (one tile per CU: C2).
// Every fp32 operand is the EXACT sum of three bf16 parts (8 + 8 + 8 significant bits: x = x1 + x2 + x3, each part the
// bf16 rounding of what the previous ones left), so a product a b is the sum of nine exact 16-bit products, of which the
// three smallest (a2 b3, a3 b2, a3 b3 < 2^-23 |a b| together) are dropped:
//   a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1),
// accumulated in the fp32 accumulator of v_mfma_f32_32x32x16_bf16, smallest terms first -- the error of a dot product is
// that of an fp32 fma chain (tests/test_gpu_parity.py compares both forms with an fp64 product).  The matrix pipe retires
// 16 k of a bf16 product in the time it retires 0.5 k of an fp32 one, so the six products cost 384 cycles per wave and
// 32-deep step against 1024: the k-step, which the fp32 form leaves at 2 380 cycles (2 048 of them matrix instructions), drops
// to ~1 100.  Weights are split at context creation, the activations by the hidden-layer kernel's epilogue.
// Stage: 3 planes x (64 A rows + 128 B rows) x 64 B = 36 KB, four stages; a 1-KiB DMA piece = 16 rows of one plane (36 pieces:
// waves 0-3 move five, waves 4-7 four); 16-byte chunk c of tile row r sits at chunk c ^ ((r >> 2) & 3): the sixteen lanes of a
// fragment read cover sixteen different bank groups.
// ----------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned short bf16_bits(__bf16 v) { return __builtin_bit_cast(unsigned short, v); }
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  const __bf16 b1 = (__bf16)x;
  const float r1 = x - (float)b1;                          // exact
  const __bf16 b2 = (__bf16)r1;
  const float r2 = r1 - (float)b2;                         // exact
  const __bf16 b3 = (__bf16)r2;
  h = bf16_bits(b1); m = bf16_bits(b2); l = bf16_bits(b3);
}
// [rows][pitch] fp32 -> three bf16 planes of the same shape (context creation: the output layer's padded weights)
__global__ void payne_split3_kernel(const float* __restrict__ src, size_t n, unsigned short* __restrict__ dst, size_t plane);
#ifdef PAYNE_TU_DENSE
__global__ void __launch_bounds__(256) payne_split3_kernel(const float* __restrict__ src, size_t n, unsigned short* __restrict__ dst, size_t plane) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned short h, m, l;
  split3(src[i], h, m, l);
  dst[i] = h; dst[plane + i] = m; dst[2 * plane + i] = l;
}
#endif

// ----------------------------------------------------------------------------
// Two fp16 planes an operand, three products (payne_dense_dma2h_kernel).  X = x * s (s a power of two) = h1 + h2 + rest, h1 = fp16(X),
// h2 = fp16(X - h1): 22 significant bits of X wherever h2 is a normal number (|X| >= 2^-3), an absolute 2^-25 below.  A product
// a b = a1 b1 + (a1 b2 + a2 b1) + a2 b2 [dropped: <= 2^-22 |a b|]: three exact 22-bit products in the fp32 accumulator of
// v_mfma_f32_32x32x16_f16 instead of six bf16 ones -- half the matrix instructions, two thirds of the operand bytes and of the
// fragment reads.  On the C2 network the rows come out with the rms error of the fp32 chain (1.0e-9 against 0.93e-9, rows of rms
// 3e-3; frequency rows 4.5e-8 against 4.2e-8 on 0.135): the accumulator's own roundings dominate both (tests/test_gpu_parity.py
// compares all forms with an fp64 product).  fp16 has five exponent bits: the weights' planes are scaled ROW BY ROW to |X| < 2^15
// (payne_ctx_create; a row's scale comes back in the epilogue), the activations by one power of two per context, calibrated at
// payne_ctx_create on the label box (corners, centre, 64 points) with a factor of 8 to spare (what lies beyond saturates; NaN stays NaN).
// ----------------------------------------------------------------------------
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned short f16_bits(_Float16 v) { return __builtin_bit_cast(unsigned short, v); }
__device__ __forceinline__ void split2h(float x, float scale, unsigned short& h1, unsigned short& h2) {
  float X = x * scale;                                     // (a power of two: exact)
  X = (__builtin_fabsf(X) > 65504.0f) ? __builtin_copysignf(65504.0f, X) : X;      // saturate; NaN compares false and stays
  const _Float16 a = (_Float16)X;
  const float r = X - (float)a;                            // exact
  h1 = f16_bits(a); h2 = f16_bits((_Float16)r);
}
// Two values at once for the kernels' epilogues: v_cvt_pkrtz_f16_f32 rounds TOWARD ZERO -- h1 + h2 then hold 21 significant bits
// instead of 22 (still below the accumulator's own roundings), a value beyond fp16's range comes out as the largest finite half by
// itself (no clamp), NaN stays NaN -- and leaves both halves packed as the planes want them: eight vector instructions a pair.
typedef __fp16 pkh2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2h_pair(float x0, float x1, float scale, unsigned& p1, unsigned& p2) {
  const float X0 = x0 * scale, X1 = x1 * scale;
  const pkh2_t a = __builtin_amdgcn_cvt_pkrtz(X0, X1);
  const float r0 = X0 - (float)a[0], r1 = X1 - (float)a[1];          // exact
  const pkh2_t b = __builtin_amdgcn_cvt_pkrtz(r0, r1);
  p1 = __builtin_bit_cast(unsigned, a); p2 = __builtin_bit_cast(unsigned, b);
}
// [rows][pitch] fp32 -> two fp16 planes of the same shape, row r scaled by scale[r] (context creation: the output layer's padded weights)
__global__ void payne_split2h_kernel(const float* __restrict__ src, int rows, int pitch, const float* __restrict__ scale,
                                     unsigned short* __restrict__ dst, size_t plane);
#ifdef PAYNE_TU_DENSE
__global__ void __launch_bounds__(256) payne_split2h_kernel(const float* __restrict__ src, int rows, int pitch, const float* __restrict__ scale,
                                                            unsigned short* __restrict__ dst, size_t plane) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)rows * pitch) return;
  unsigned short h1, h2;
  split2h(src[i], scale[i / pitch], h1, h2);
  dst[i] = h1; dst[plane + i] = h2;
}
#endif
// Stage: two planes x (64 A rows + 128 B rows) x 2 KD bytes (KD = 32: 24 KB, four stages; KD = 64: 48 KB, three stages -- half as many
// barrier steps for the same matrix instructions: a step's fixed cost (wait, barrier, fragment reads, requests: 300-500 cycles) is
// as long as its six matrix instructions per wave at KD = 32).  A 1-KiB piece = 1024 / (2 KD) rows of one plane; 16-byte chunk c of
// tile row r sits at chunk c ^ sw(r), sw = (r >> 2) & 3 for 64-byte rows, (r >> 1) & 7 for 128-byte rows: the sixteen lanes of
// every lane group of a fragment read (ds_read_b128) cover sixteen different 16-byte bank groups.
// The rows' stores: streamed (nt) -- the next reader is the post kernel, on other XCDs.  (PAYNE_EXP_ST: timing twins of other cache
// policies, tools/exp/store_policy.py: 1 plain, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0 nt.)
__device__ __forceinline__ void d2_store_row(float* q, float v) {
#if defined(PAYNE_EXP_ST) && PAYNE_EXP_ST == 1
  *q = v;
#elif defined(PAYNE_EXP_ST) && PAYNE_EXP_ST == 2
  asm volatile("global_store_dword %0, %1, off sc1" :: "v"(q), "v"(v) : "memory");
#elif defined(PAYNE_EXP_ST) && PAYNE_EXP_ST == 3
  asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(q), "v"(v) : "memory");
#elif defined(PAYNE_EXP_ST) && PAYNE_EXP_ST == 4
  asm volatile("global_store_dword %0, %1, off sc1 nt" :: "v"(q), "v"(v) : "memory");
#elif defined(PAYNE_EXP_ST) && PAYNE_EXP_ST == 5
  asm volatile("global_store_dword %0, %1, off sc0 nt" :: "v"(q), "v"(v) : "memory");
#else
  __builtin_nontemporal_store(v, q);
#endif
}
template <int KD> constexpr int d2_stage() { return 2 * (64 + 128) * 2 * KD; }
template <int KD> constexpr int d2_ns() { return KD == 64 ? 3 : 4; }
template <int KD> constexpr size_t d2_lds_bytes() { return (size_t)d2_ns<KD>() * d2_stage<KD>(); }
// The schedule of payne_dense_dma3_kernel<NK, 4, true>; NK: k-steps (of KD) fixed at compile time, or 0.
template <int NK, int KD>
__global__ void __launch_bounds__(512) payne_dense_dma2h_kernel(PAYNE_D3_LEAD_PARAMS, DenseParams p_) {
  DenseParams p = p_;
  p.sel = lead_sel; p.Xp = lead_Xp; p.Wp = lead_Wp; p.plane_x = lead_plane_x; p.plane_w = lead_plane_w;
  p.grid_m = (int)(lead_grid & 0xffffu); p.grid_n = (int)(lead_grid >> 16); p.N = lead_N; p.B = lead_B; p.ldp = lead_ldp; p.K = lead_K;
  constexpr int NS = d2_ns<KD>(), AHEAD = NS - 1, STAGE = d2_stage<KD>();
  constexpr int ROWB = 2 * KD, CPR = ROWB / 16, RPP = 1024 / ROWB;          // bytes a row, chunks a row, rows a piece
  constexpr int A_PLANE = 64 * ROWB, B_PLANE = 128 * ROWB, NPA = 64 / RPP, NPB = 128 / RPP, PW = 2 * (NPA + NPB) / 8;   // pieces a wave: 3 | 6
  constexpr int KS = KD / 16;                              // 16-deep matrix steps a stage
  auto sw = [](int row) { return KD == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
  extern __shared__ __attribute__((aligned(16))) unsigned char d2_sm[];
  if (p.sel != nullptr && (unsigned)*p.sel == lead_sel_seq) {
    p.Wp = p.Wp_alt; p.plane_w = p.plane_w_alt; p.bias = p.bias_alt; p.bias_shift = p.bias_shift_alt; p.N = p.N_alt; p.ldy = p.ldy_alt;
    p.rscale = p.rscale_alt;
  }
  const int ntiles = p.grid_m * p.grid_n;
  int t = blockIdx.x;
  if ((ntiles & 7) == 0) t = (t & 7) * (ntiles >> 3) + (t >> 3);      // XCD-aware order (see payne_dense_kernel)
  const int m0 = (t % p.grid_m) * 64, n0 = (t / p.grid_m) * 128;
  if (n0 >= p.N) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 2) * 32, wn0 = (wave & 3) * 32;
  // A piece's source = a base every lane shares (operand, plane: scalar registers) + the lane's 32-bit byte offset (row, chunk), the
  // k-step an immediate: no 64-bit vector arithmetic per address -- as first written the set-up in front of the first request was 370
  // instructions a wave (launch_out_dma2h checks that the offsets fit).
  const unsigned char* sbase[PW];
  unsigned voff[PW];
  int dst[PW];
  const int lrow = lane / CPR, lchunk = lane % CPR;
  const unsigned pitch_a = 2u * (unsigned)p.ldp, pitch_b = 2u * (unsigned)p.K;
  // Which pieces a wave moves is fixed by the piece's INDEX j, not worked out from the wave's number (as first written: a chain of scalar
  // branches a piece): KD = 64: activations (plane j, block w), j = 0, 1; weights (plane (j - 2) / 2, block 2 w + (j - 2) % 2), j = 2 .. 5;
  // KD = 32: activations (plane w / 4, block w % 4), j = 0; weights (plane j - 1, block w), j = 1, 2.  Where a piece lands in LDS is as before.
  constexpr int PWA = 2 * NPA / 8;
  static_assert((KD == 64 && NPA == 8 && NPB == 16 && PWA == 2) || (KD == 32 && NPA == 4 && NPB == 8 && PWA == 1), "the piece map below");
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const bool isA = j < PWA;
    const int jb = j - PWA;
    const int pl = isA ? (KD == 64 ? j : wave >> 2) : (KD == 64 ? jb >> 1 : jb);
    const int blk = isA ? (KD == 64 ? wave : wave & 3) : (KD == 64 ? 2 * wave + (jb & 1) : wave);
    const int row = RPP * blk + lrow;
    const int c = lchunk ^ sw(row);                        // which 16-byte chunk of the row belongs in this lane's slot
    const int top = isA ? p.B - 1 - m0 : p.N - 1 - n0;     // (scalar) last real row of the tile
    const int r = (isA ? m0 : n0) + (row < top ? row : top);
    sbase[j] = isA ? reinterpret_cast<const unsigned char*>(p.Xp) + 2 * (size_t)pl * p.plane_x
                   : reinterpret_cast<const unsigned char*>(p.Wp) + 2 * (size_t)pl * p.plane_w;
    voff[j] = (unsigned)r * (isA ? pitch_a : pitch_b) + 16u * (unsigned)c;
    dst[j] = isA ? pl * A_PLANE + blk * 1024 : 2 * A_PLANE + pl * B_PLANE + blk * 1024;
#if defined(PAYNE_EXP_D2H) && (PAYNE_EXP_D2H & 1)       /* timing twin (tools/exp/d2h_ablate_time.py): every piece comes from ONE kilobyte (no L2 traffic to speak of) */
    sbase[j] = reinterpret_cast<const unsigned char*>(p.Xp); voff[j] = 16u * (unsigned)lane;
#endif
  }
  auto issue = [&](int stage, int k0) {                    // k0 in elements (2 bytes each)
#pragma unroll
    for (int j = 0; j < PW; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbase[j] + voff[j] + 2 * k0),
                                       (__attribute__((address_space(3))) void*)(d2_sm + stage * STAGE + dst[j]), 16, 0, 0);
  };
  auto wait_landed = [&](int younger) {                    // my pieces of a stage have landed once only `younger` stages' loads are outstanding
    if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PW) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PW) : "memory");
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int Ra = wm0 + (lane & 31), Rb = wn0 + (lane & 31), h = lane >> 5;
  const int sa = sw(Ra), sb = sw(Rb);
  struct Frag { f16x8_t a[KS][2], b[KS][2]; };
  auto frags = [&](int stage, Frag& f) {
    const unsigned char* As = d2_sm + stage * STAGE;
    const unsigned char* Bs = As + 2 * A_PLANE;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = 2 * ks + h;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        f.a[ks][pl] = *reinterpret_cast<const f16x8_t*>(As + pl * A_PLANE + Ra * ROWB + 16 * (c ^ sa));
        f.b[ks][pl] = *reinterpret_cast<const f16x8_t*>(Bs + pl * B_PLANE + Rb * ROWB + 16 * (c ^ sb));
      }
    }
  };
  auto products = [&](const Frag& f, int ks) {             // smallest partial products first
#if defined(PAYNE_EXP_D2H) && (PAYNE_EXP_D2H & 2)       /* timing twin: one of the three products */
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[ks][0] + f.a[ks][1], f.b[ks][0] + f.b[ks][1], acc, 0, 0, 0);
#else
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[ks][1], f.b[ks][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[ks][0], f.b[ks][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[ks][0], f.b[ks][0], acc, 0, 0, 0);
#endif
  };
  const int nk = NK > 0 ? NK : p.K / KD;                   // padded: exact
  const int k_tail = (p.k_real > 0 ? p.k_real : p.K) - (nk - 1) * KD;
  const int ks_last = __builtin_amdgcn_readfirstlane((k_tail + 15) >> 4);   // the zero-padded 16-deep steps of the last stage are skipped
  auto all_products = [&](const Frag& f) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) products(f, ks);
  };
  auto last_products = [&](const Frag& f) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      if (ks < ks_last) products(f, ks);
  };
  // (the epilogue's bias and row scale, requested before the first transfer)
  const int col = n0 + wn0 + (lane & 31);
  const float bv = p.bias[col < p.N ? col : p.N - 1] - p.bias_shift;
  const float rs = p.rscale[col < p.N ? col : p.N - 1];
  HK_STAMP(0);
  const int npro = nk < AHEAD ? nk : AHEAD;
#pragma unroll
  for (int q = 0; q < AHEAD; ++q)
    if (q < npro) issue(q, q * KD);
  wait_landed(npro - 1);
  asm volatile("s_barrier" ::: "memory");
  Frag f0, f1;
  frags(0, f0);
  auto head = [&](int it, Frag& fn, const Frag& fc) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) asm volatile("" :: "v"(fc.a[ks][pl]), "v"(fc.b[ks][pl]));
    {
      const int last = (it + AHEAD < nk ? it + AHEAD : nk) - 1;
      wait_landed(last - (it + 1));
    }
    asm volatile("s_barrier" ::: "memory");
#ifndef PAYNE_NO_LOOP_STAMPS
    if (it < 13) HK_STAMP(1 + it);
#endif
    frags((it + 1) % NS, fn);
    __builtin_amdgcn_sched_barrier(0);
    if (it + AHEAD < nk) issue((it + AHEAD) % NS, (it + AHEAD) * KD);
    __builtin_amdgcn_sched_barrier(0);
  };
  // C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  auto epilogue = [&]() {
  if (col < p.N) {
    const bool act_none = __builtin_amdgcn_readfirstlane(p.act == PAYNE_ACT_NONE ? 1 : 0) != 0;
    if (act_none && m0 + 64 <= p.B) {
      // every row of the tile is a candidate's (the usual batch): a value's address = a base every lane shares (its row's start: scalar
      // registers, one scalar add a row) + the lane's 32-bit offset (column, upper half) -- as first written a row test, a 64-bit multiply
      // (quarter rate) and a 64-bit add per value: nine instructions a store
      const unsigned voff = 4u * ((unsigned)(4 * (lane >> 5)) * (unsigned)p.ldy + (unsigned)col);
      const unsigned char* yb = reinterpret_cast<const unsigned char*>(p.Y + (size_t)(m0 + wm0) * p.ldy);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned char* rowp = yb + (size_t)((r & 3) + 8 * (r >> 2)) * p.ldy * 4;
#if defined(PAYNE_EXP_D2H) && (PAYNE_EXP_D2H & 4)       /* timing twin: (practically) no stores */
        if (!(acc[r] == 12345.678f)) continue;
#endif
        d2_store_row(reinterpret_cast<float*>(const_cast<unsigned char*>(rowp + voff)), __builtin_fmaf(acc[r], rs, bv));
      }
    } else if (act_none) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < p.B) d2_store_row(&p.Y[(size_t)row * p.ldy + col], __builtin_fmaf(acc[r], rs, bv));
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < p.B) __builtin_nontemporal_store(act_apply(__builtin_fmaf(acc[r], rs, bv), p.act), &p.Y[(size_t)row * p.ldy + col]);
      }
    }
  }
  };
  int it = 0;
#pragma unroll
  for (; it + 2 < nk; it += 2) {
    head(it, f1, f0);
    all_products(f0);
    __builtin_amdgcn_sched_barrier(0);
    head(it + 1, f0, f1);
    all_products(f1);
    __builtin_amdgcn_sched_barrier(0);
#if defined(PAYNE_EXP_D2H) && (PAYNE_EXP_D2H & 24)      /* timing twins: half of the waves store in the middle of the k-loop and leave (what a tile finished in two
                                                           halves would do; their later operand pieces are not requested: an upper bound) -- 8: after two stages of five, 16: after four */
    if (it == ((PAYNE_EXP_D2H & 8) ? 0 : 2) && wave < 4) { epilogue(); return; }
#endif
  }
  if (it + 1 < nk) {
    head(it, f1, f0);
    all_products(f0);
    last_products(f1);
  } else {
    last_products(f0);
  }
  epilogue();
  HK_STAMP(15);
}

// ----------------------------------------------------------------------------
// payne_dense_dma2hh_kernel<NK>: payne_dense_dma2h_kernel<NK, 64>'s tile (64 candidates x 128 outputs, one a CU) FINISHED IN TWO HALVES.
// The rows' stores are 2.6 of that kernel's 9 us (tools/exp/d2h_ablate_time.py): every tile ends its k-loop at the same moment and 8.4 MB
// leave at once.  Here the activations' NK stages stay resident in LDS (NK x 16 KB) while the weights stream through a ring of three
// 16-KB slots one HALF of the tile's 128 output rows after the other: pass 1 = stages (A_q, BL_q), waves 0-3 multiply the left 64 x 64
// (one wave a SIMD), store it and LEAVE -- stores share the counter the operand waits are counted on, a wave with stores in flight cannot
// wait for later loads precisely --; pass 2 = stages (BR_q), waves 4-7 multiply the right half under the left half's stores.  Same operand
// bytes (246 KB a tile), same products in the same order: the rows are payne_dense_dma2h_kernel's to the bit (tests/test_gpu_parity.py).
// Waves 4-7 request their quarter of a pass-1 stage and all of a pass-2 stage (four 1-KB pieces either way), waves 0-3 their quarter of
// pass 1 only; requests run two stages ahead across the passes' boundary.
// ----------------------------------------------------------------------------
template <int NK> constexpr size_t d2hh_lds_bytes() { return (size_t)NK * 16384 + 3 * 16384; }
template <int NK>
__global__ void __launch_bounds__(512) payne_dense_dma2hh_kernel(PAYNE_D3_LEAD_PARAMS, DenseParams p_) {
  DenseParams p = p_;
  p.sel = lead_sel; p.Xp = lead_Xp; p.Wp = lead_Wp; p.plane_x = lead_plane_x; p.plane_w = lead_plane_w;
  p.grid_m = (int)(lead_grid & 0xffffu); p.grid_n = (int)(lead_grid >> 16); p.N = lead_N; p.B = lead_B; p.ldp = lead_ldp; p.K = lead_K;
  constexpr int KD = 64, ROWB = 2 * KD, CPR = ROWB / 16, RPP = 1024 / ROWB;      // bytes a row, chunks a row (8), rows a piece (8)
  constexpr int PLANE = 64 * ROWB, STAGE = 2 * PLANE;                            // a 64-row plane (8 KB); two planes: an A stage, a B-half stage
  constexpr int NSB = 3, AHEAD = 2, KS = KD / 16, NT = 2 * NK;                   // ring slots, stages requested ahead, matrix steps a stage, stages
  constexpr int RING = NK * STAGE;
  auto sw = [](int row) { return (row >> 1) & 7; };
  extern __shared__ __attribute__((aligned(16))) unsigned char d2_sm[];
  if (p.sel != nullptr && (unsigned)*p.sel == lead_sel_seq) {
    p.Wp = p.Wp_alt; p.plane_w = p.plane_w_alt; p.bias = p.bias_alt; p.bias_shift = p.bias_shift_alt; p.N = p.N_alt; p.ldy = p.ldy_alt;
    p.rscale = p.rscale_alt;
  }
  const int ntiles = p.grid_m * p.grid_n;
  int t = blockIdx.x;
  if ((ntiles & 7) == 0) t = (t & 7) * (ntiles >> 3) + (t >> 3);      // XCD-aware order (see payne_dense_kernel)
  const int m0 = (t % p.grid_m) * 64, n0 = (t / p.grid_m) * 128;
  if (n0 >= p.N) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, w4 = wave & 3;                          // which half of the tile's outputs this wave multiplies
  const int wm0 = (w4 >> 1) * 32, wn0 = 64 * half + (w4 & 1) * 32;
  // pieces: 8 rows x 128 bytes of one plane.  Pass 1 (every wave): activations (plane j, block wave), j = 0, 1; left weights (plane j - 2,
  // block wave), j = 2, 3.  Pass 2 (waves 4-7): right weights (plane j / 2, block 2 w4 + j % 2), j = 0 .. 3.
  const int lrow = lane / CPR, lchunk = lane % CPR;
  const unsigned pitch_a = 2u * (unsigned)p.ldp, pitch_b = 2u * (unsigned)p.K;
  const unsigned char* sb1[4]; unsigned vo1[4]; int ds1[4];
  const unsigned char* sb2[4]; unsigned vo2[4]; int ds2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    {
      const bool isA = j < 2;
      const int pl = j & 1, blk = wave;
      const int row = RPP * blk + lrow, c = lchunk ^ sw(row);
      const int top = isA ? p.B - 1 - m0 : p.N - 1 - n0;
      const int r = (isA ? m0 : n0) + (row < top ? row : top);
      sb1[j] = isA ? reinterpret_cast<const unsigned char*>(p.Xp) + 2 * (size_t)pl * p.plane_x
                   : reinterpret_cast<const unsigned char*>(p.Wp) + 2 * (size_t)pl * p.plane_w;
      vo1[j] = (unsigned)r * (isA ? pitch_a : pitch_b) + 16u * (unsigned)c;
      ds1[j] = pl * PLANE + blk * 1024;                                // (inside the stage's A block, or inside its ring slot)
    }
    {
      const int pl = j >> 1, blk = 2 * w4 + (j & 1);
      const int row = RPP * blk + lrow, c = lchunk ^ sw(row);
      const int top = p.N - 1 - n0;
      const int r = n0 + (64 + row < top ? 64 + row : top);
      sb2[j] = reinterpret_cast<const unsigned char*>(p.Wp) + 2 * (size_t)pl * p.plane_w;
      vo2[j] = (unsigned)r * pitch_b + 16u * (unsigned)c;
      ds2[j] = pl * PLANE + blk * 1024;
    }
  }
  // stage q of the sequence: q < NK: (A_q -> its resident block, BL_q -> ring slot q % 3); q >= NK: BR_(q - NK) -> ring slot q % 3
  auto issue = [&](int q) {
    if (q < NK) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb1[j] + vo1[j] + 2 * KD * q),
                                         (__attribute__((address_space(3))) void*)(d2_sm + (j < 2 ? q * STAGE : RING + (q % NSB) * STAGE) + ds1[j]), 16, 0, 0);
    } else if (half) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb2[j] + vo2[j] + 2 * KD * (q - NK)),
                                         (__attribute__((address_space(3))) void*)(d2_sm + RING + (q % NSB) * STAGE + ds2[j]), 16, 0, 0);
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int Ra = wm0 + (lane & 31), Rb = (w4 & 1) * 32 + (lane & 31), h = lane >> 5;      // rows inside the stage's A block / B-half slot
  const int sa = sw(Ra), sb = sw(Rb);
  struct Frag { f16x8_t a[KS][2], b[KS][2]; };
  auto frags = [&](int q, Frag& f) {                          // stage q's operands of this wave's 32 x 32 block
    const unsigned char* As = d2_sm + (q < NK ? q : q - NK) * STAGE;
    const unsigned char* Bs = d2_sm + RING + (q % NSB) * STAGE;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = 2 * ks + h;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        f.a[ks][pl] = *reinterpret_cast<const f16x8_t*>(As + pl * PLANE + Ra * ROWB + 16 * (c ^ sa));
        f.b[ks][pl] = *reinterpret_cast<const f16x8_t*>(Bs + pl * PLANE + Rb * ROWB + 16 * (c ^ sb));
      }
    }
  };
  auto products = [&](const Frag& f, int ks) {             // smallest partial products first (payne_dense_dma2h_kernel's order)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[ks][1], f.b[ks][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[ks][0], f.b[ks][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[ks][0], f.b[ks][0], acc, 0, 0, 0);
  };
  const int k_tail = (p.k_real > 0 ? p.k_real : p.K) - (NK - 1) * KD;
  const int ks_last = __builtin_amdgcn_readfirstlane((k_tail + 15) >> 4);   // the zero-padded 16-deep steps of a pass's last stage are skipped
  auto stage_products = [&](const Frag& f, bool last) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      if (!last || ks < ks_last) products(f, ks);
  };
  const int col = n0 + wn0 + (lane & 31);
  const float bv = p.bias[col < p.N ? col : p.N - 1] - p.bias_shift;
  const float rs = p.rscale[col < p.N ? col : p.N - 1];
  auto epilogue = [&]() {                                     // C/D map of the 32x32 block: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (col >= p.N) return;
    const bool act_none = __builtin_amdgcn_readfirstlane(p.act == PAYNE_ACT_NONE ? 1 : 0) != 0;
    if (act_none && m0 + 64 <= p.B) {
      const unsigned voff = 4u * ((unsigned)(4 * (lane >> 5)) * (unsigned)p.ldy + (unsigned)col);
      const unsigned char* yb = reinterpret_cast<const unsigned char*>(p.Y + (size_t)(m0 + wm0) * p.ldy);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned char* rowp = yb + (size_t)((r & 3) + 8 * (r >> 2)) * p.ldy * 4;
        d2_store_row(reinterpret_cast<float*>(const_cast<unsigned char*>(rowp + voff)), __builtin_fmaf(acc[r], rs, bv));
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < p.B) {
          const float y = __builtin_fmaf(acc[r], rs, bv);
          __builtin_nontemporal_store(act_none ? y : act_apply(y, p.act), &p.Y[(size_t)row * p.ldy + col]);
        }
      }
    }
  };
  HK_STAMP(0);
  Frag f0, f1;
  // The two halves' waves run two different programs over the same barriers (NK + 1 of them while both are there).
  if (half == 0) {
    // ---- waves 0-3: pass 1 -- request my quarter of its stages, multiply the left 64 x 64, store it, leave ----
    issue(0); issue(1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");         // stage 0 (stage 1's four pieces behind it)
    asm volatile("s_barrier" ::: "memory");
    frags(0, f0);
#pragma unroll
    for (int q = 0; q + 1 < NK; ++q) {
      Frag& fc = (q & 1) ? f1 : f0;
      Frag& fn = (q & 1) ? f0 : f1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // stage q + 1 (the next request follows the barrier)
      asm volatile("s_barrier" ::: "memory");
      HK_STAMP(1 + q);                                       // (diagnostic build: the left half's life, by its first wave)
      frags(q + 1, fn);
      __builtin_amdgcn_sched_barrier(0);
      if (q + AHEAD < NK) issue(q + AHEAD);
      __builtin_amdgcn_sched_barrier(0);
      stage_products(fc, false);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_barrier" ::: "memory");                  // (the barrier the other half counts on at the passes' boundary)
    stage_products(((NK - 1) & 1) ? f1 : f0, true);
    HK_STAMP(NK);
    epilogue();
    HK_STAMP(15);
    return;
  }
  // ---- waves 4-7: request my quarter of pass 1's stages and all of pass 2's; multiply the right 64 x 64 in pass 2 ----
  issue(0); issue(1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
#pragma unroll
  for (int q = 0; q + 1 < NK; ++q) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    issue(q + AHEAD);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // stage NK: the right half's first
  asm volatile("s_barrier" ::: "memory");
  frags(NK, f0);
  __builtin_amdgcn_sched_barrier(0);
  if (NK + 1 < NT) issue(NK + 1);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = NK; q < NT; ++q) {
    Frag& fc = ((q - NK) & 1) ? f1 : f0;
    Frag& fn = ((q - NK) & 1) ? f0 : f1;
    if (q + 1 < NT) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");                // (the left half's waves have left, or are about to: they are not waited for once gone)
      frags(q + 1, fn);
      __builtin_amdgcn_sched_barrier(0);
      if (q + AHEAD < NT) issue(q + AHEAD);
      __builtin_amdgcn_sched_barrier(0);
    }
    stage_products(fc, q == NT - 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  epilogue();
  HK_STAMP(15);
}

constexpr int D3_STAGE = 3 * (64 + 128) * 64;              // bytes per stage
// NS = 4, PIPE: one tile per CU (C2).  NS = 2, !PIPE: many tiles per CU (C5) -- two stages = 72 KB, two workgroups per CU, and per
// step: wait, barrier, this step's fragments, the request for the next stage, the products (the other workgroup's products fill
// the LDS round trip; the loop is bound by the 36 KB a step brings in either way).
template <int NS> constexpr size_t d3_lds_bytes() { return (size_t)NS * D3_STAGE; }
template <int NK, int NS, bool PIPE>
__global__ void __launch_bounds__(512) payne_dense_dma3_kernel(PAYNE_D3_LEAD_PARAMS, DenseParams p_) {
  // (what the prologue's addresses hang off arrives in registers at wave start: the leading scalar parameters, preloaded --
  //  -mllvm -amdgpu-kernarg-preload-count; read from the kernarg segment they are 400-700 cycles in front of the first request)
  DenseParams p = p_;
  p.sel = lead_sel; p.Xp = lead_Xp; p.Wp = lead_Wp; p.plane_x = lead_plane_x; p.plane_w = lead_plane_w;
  p.grid_m = (int)(lead_grid & 0xffffu); p.grid_n = (int)(lead_grid >> 16); p.N = lead_N; p.B = lead_B; p.ldp = lead_ldp; p.K = lead_K;
  constexpr int D3_NS = NS, AHEAD = PIPE ? D3_NS - 1 : 1;
  static_assert((PIPE && NS >= 3) || (!PIPE && NS == 2 && NK == 0), "ring depth / schedule");
  extern __shared__ __attribute__((aligned(16))) unsigned char d3_sm[];
  if (p.sel != nullptr && (unsigned)*p.sel == lead_sel_seq) {   // (a scalar load and a uniform branch; launches without the word skip both)
    p.Wp = p.Wp_alt; p.plane_w = p.plane_w_alt; p.bias = p.bias_alt; p.bias_shift = p.bias_shift_alt; p.N = p.N_alt; p.ldy = p.ldy_alt;
  }
  const int ntiles = p.grid_m * p.grid_n;
  int t = blockIdx.x;
  if ((ntiles & 7) == 0) t = (t & 7) * (ntiles >> 3) + (t >> 3);      // XCD-aware order (see payne_dense_kernel)
  const int m0 = (t % p.grid_m) * 64, n0 = (t / p.grid_m) * 128;
  if (n0 >= p.N) return;                                     // (the grid was sized for the wider of the two output layers)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 2) * 32, wn0 = (wave & 3) * 32;
  const bool five = wave < 4;                              // pieces this wave moves per stage: 5 (waves 0-3) or 4
  const unsigned char* src[5];
  int dst[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    int q = five ? wave * 5 + j : 20 + (wave - 4) * 4 + j;
    if (q > 35) q = 35;                                    // (slot 4 of the four-piece waves: never issued)
    const bool isA = q < 12;
    const int pl = isA ? q >> 2 : (q - 12) >> 3, blk = isA ? (q & 3) : ((q - 12) & 7);
    const int row = 16 * blk + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);           // which 16-byte chunk of the row belongs in this lane's slot
    if (isA) {
      const int r = (m0 + row < p.B) ? m0 + row : p.B - 1;
      src[j] = reinterpret_cast<const unsigned char*>(p.Xp + (size_t)pl * p.plane_x + (size_t)r * p.ldp) + 16 * c;
      dst[j] = pl * 4096 + blk * 1024;
    } else {
      const int r = (n0 + row < p.N) ? n0 + row : p.N - 1;
      src[j] = reinterpret_cast<const unsigned char*>(p.Wp + (size_t)pl * p.plane_w + (size_t)r * p.K) + 16 * c;
      dst[j] = 3 * 4096 + pl * 8192 + blk * 1024;
    }
  }
  auto issue = [&](int stage, int k0) {                    // k0 in elements (2 bytes each)
#pragma unroll
    for (int j = 0; j < 5; ++j)
      if (j < 4 || five)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + 2 * k0),
                                         (__attribute__((address_space(3))) void*)(d3_sm + stage * D3_STAGE + dst[j]), 16, 0, 0);
  };
  auto wait_landed = [&](int younger) {                    // my pieces of a stage have landed once only `younger` stages' loads are outstanding
    if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (five) { if (younger == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }
    else { if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int Ra = wm0 + (lane & 31), Rb = wn0 + (lane & 31), h = lane >> 5;
  const int sa = (Ra >> 2) & 3, sb = (Rb >> 2) & 3;
  struct Frag { bf16x8_t a[2][3], b[2][3]; };
  auto frags = [&](int stage, Frag& f) {
    const unsigned char* As = d3_sm + stage * D3_STAGE;
    const unsigned char* Bs = As + 3 * 4096;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {                       // two 16-deep matrix steps per 32-deep stage
      const int c = 2 * ks + h;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        f.a[ks][pl] = *reinterpret_cast<const bf16x8_t*>(As + pl * 4096 + Ra * 64 + 16 * (c ^ sa));
        f.b[ks][pl] = *reinterpret_cast<const bf16x8_t*>(Bs + pl * 8192 + Rb * 64 + 16 * (c ^ sb));
      }
    }
  };
  auto products = [&](const Frag& f, int ks) {             // smallest partial products first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][2], f.b[ks][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][1], f.b[ks][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][0], f.b[ks][2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][1], f.b[ks][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][0], f.b[ks][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][0], f.b[ks][0], acc, 0, 0, 0);
  };
  const int nk = NK > 0 ? NK : p.K / 32;                   // padded: exact
  const int k_tail = (p.k_real > 0 ? p.k_real : p.K) - (nk - 1) * 32;
  const bool last_both = __builtin_amdgcn_readfirstlane(k_tail > 16 ? 1 : 0) != 0;   // the zero-padded half of the last step is skipped
  // (the epilogue's bias, requested before the first transfer: asked for at the end it is a memory round trip of its own)
  const int col = n0 + wn0 + (lane & 31);
  const float bv = p.bias[col < p.N ? col : p.N - 1] - p.bias_shift;
  HK_STAMP(0);
  if constexpr (!PIPE) {
    issue(0, 0);
    for (int it = 0; it < nk; ++it) {
      wait_landed(0);                                      // stage `it` (the only one outstanding)
      asm volatile("s_barrier" ::: "memory");              // everybody's pieces landed; everybody finished step it-1
      Frag f;
      frags(it % D3_NS, f);
      __builtin_amdgcn_sched_barrier(0);
      if (it + 1 < nk) issue((it + 1) % D3_NS, (it + 1) * 32);   // into the buffer step it-1 consumed
      __builtin_amdgcn_sched_barrier(0);
      products(f, 0);
      if (it + 1 < nk || last_both) products(f, 1);
    }
  } else {
  const int npro = nk < AHEAD ? nk : AHEAD;
#pragma unroll
  for (int q = 0; q < AHEAD; ++q)
    if (q < npro) issue(q, q * 32);
  wait_landed(npro - 1);
  asm volatile("s_barrier" ::: "memory");
  Frag f0, f1;
  frags(0, f0);
  // (the schedule of payne_dense_dma_kernel<.., PIPE>: fragments of the next step, THEN the request, then this step's products)
  auto head = [&](int it, Frag& fn, const Frag& fc) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) asm volatile("" :: "v"(fc.a[ks][pl]), "v"(fc.b[ks][pl]));
    {
      const int last = (it + AHEAD < nk ? it + AHEAD : nk) - 1;
      wait_landed(last - (it + 1));
    }
    asm volatile("s_barrier" ::: "memory");
#ifndef PAYNE_NO_LOOP_STAMPS
    if (it < 13) HK_STAMP(1 + it);
#endif
    frags((it + 1) % D3_NS, fn);
    __builtin_amdgcn_sched_barrier(0);
    if (it + AHEAD < nk) issue((it + AHEAD) % D3_NS, (it + AHEAD) * 32);
    __builtin_amdgcn_sched_barrier(0);
  };
  int it = 0;
#pragma unroll
  for (; it + 2 < nk; it += 2) {
    head(it, f1, f0);
    products(f0, 0); products(f0, 1);
    __builtin_amdgcn_sched_barrier(0);
    head(it + 1, f0, f1);
    products(f1, 0); products(f1, 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (it + 1 < nk) {
    head(it, f1, f0);
    products(f0, 0); products(f0, 1);
    products(f1, 0);
    if (last_both) products(f1, 1);
  } else {
    products(f0, 0);
    if (last_both) products(f0, 1);
  }
  }
  // C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  if (col < p.N) {
    const bool act_none = __builtin_amdgcn_readfirstlane(p.act == PAYNE_ACT_NONE ? 1 : 0) != 0;   // (the usual output layer: one test, not sixteen)
    if (act_none) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < p.B) __builtin_nontemporal_store(acc[r] + bv, &p.Y[(size_t)row * p.ldy + col]);   // streamed: next read by other XCDs
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < p.B) __builtin_nontemporal_store(act_apply(acc[r] + bv, p.act), &p.Y[(size_t)row * p.ldy + col]);
      }
    }
  }
  HK_STAMP(15);
}

// ----------------------------------------------------------------------------
// payne_dense_dma3f_kernel<NK>: the six-product output layer of payne_dense_dma3_kernel<NK, 4, true> with the WEIGHTS AS fp32
// THROUGH THE PORT.  That kernel is bound by operand bytes: 36 KB of bf16 planes per 32-deep step through an L2-to-CU port that
// delivers ~22 B/clk (1 676 of a step's ~1 700 cycles) -- and three bf16 planes are 6 bytes for what 4 bytes say.  Here the
// weight tile of a step arrives as fp32 in REGISTERS (one 32-byte row chunk per thread, two steps ahead, three register sets in
// rotation), is split into the same three planes by the thread that fetched it (split3: the values payne_split3_kernel writes,
// bit for bit) and stored where the transfers of the planes used to land (same swizzle): 28 KB per step.  The ~45 vector
// instructions and three ds_write_b128 a thread spends on it per step sit between the step's twelve matrix instructions
// (sched_group_barrier).  Activations keep their planes (written once by the hidden-layer kernel) and their LDS transfers: twelve
// pieces a stage, waves 0-3 two each, waves 4-7 one.
// Per step `it` (before its products): wait for A of step it+1 and the registers of B of step it+2 (one vmcnt: both were requested
// by step it-2, only step it-1's requests may be outstanding) -> barrier -> fragments of step it+1 -> request A of step it+3 and B
// of step it+4 -> split B of step it+2 into its stage (free since step it-2's fragments were read) among the products of step it.
// Weights: [N][K] fp32, K = 32 NK, zero beyond the layer's width (payne_ctx_create keeps that copy anyway).
// ----------------------------------------------------------------------------
template <int N> __device__ __forceinline__ void d3f_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
template <int NK>
__global__ void __launch_bounds__(512) payne_dense_dma3f_kernel(PAYNE_D3_LEAD_PARAMS, DenseParams p_) {
  DenseParams p = p_;
  p.sel = lead_sel; p.Xp = lead_Xp; p.plane_x = lead_plane_x;
  const float* Wf = reinterpret_cast<const float*>(lead_Wp);
  p.grid_m = (int)(lead_grid & 0xffffu); p.grid_n = (int)(lead_grid >> 16); p.N = lead_N; p.B = lead_B; p.ldp = lead_ldp; p.K = lead_K;
  static_assert(NK >= 4, "prologue requests four steps of weights");
  constexpr int NS = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char d3_sm[];
  if (p.sel != nullptr && (unsigned)*p.sel == lead_sel_seq) {
    Wf = p.W_alt; p.bias = p.bias_alt; p.bias_shift = p.bias_shift_alt; p.N = p.N_alt; p.ldy = p.ldy_alt;
  }
  const int ntiles = p.grid_m * p.grid_n;
  int t = blockIdx.x;
  if ((ntiles & 7) == 0) t = (t & 7) * (ntiles >> 3) + (t >> 3);      // XCD-aware order (see payne_dense_kernel)
  const int m0 = (t % p.grid_m) * 64, n0 = (t / p.grid_m) * 128;
  if (n0 >= p.N) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 2) * 32, wn0 = (wave & 3) * 32;
  const bool two = wave < 4;                               // A pieces this wave moves per stage: 2 (waves 0-3) or 1
  const unsigned char* srcA[2];
  int dstA[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int q = two ? wave * 2 + j : 8 + (wave - 4);           // 12 pieces: 3 planes x 4 blocks of 16 rows
    const int pl = q >> 2, blk = q & 3;
    const int row = 16 * blk + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    const int r = (m0 + row < p.B) ? m0 + row : p.B - 1;
    srcA[j] = reinterpret_cast<const unsigned char*>(p.Xp + (size_t)pl * p.plane_x + (size_t)r * p.ldp) + 16 * c;
    dstA[j] = pl * 4096 + blk * 1024;
  }
  // the weight values of this thread: floats 4 (tid % 8) .. + 3 of the step's 32 in tile rows tid / 8 and tid / 8 + 64 -- a wave's
  // load instruction reads eight whole 128-byte row segments (one 32-byte chunk per thread, i.e. two instructions that each touch
  // half of every line, moved no fewer bytes through the port than the planes had)
  const int brow = tid >> 3, bq = tid & 7;
  const float* srcB0;
  const float* srcB1;
  {
    const int r0 = (n0 + brow < p.N) ? n0 + brow : p.N - 1, r1 = (n0 + brow + 64 < p.N) ? n0 + brow + 64 : p.N - 1;
    srcB0 = Wf + (size_t)r0 * p.K + 4 * bq;
    srcB1 = Wf + (size_t)r1 * p.K + 4 * bq;
  }
  const int dstB0 = 3 * 4096 + brow * 64 + 16 * ((bq >> 1) ^ ((brow >> 2) & 3)) + 8 * (bq & 1);
  const int dstB1 = dstB0 + 64 * 64;                        // (row + 64: the same swizzle, 64 rows of 64 bytes further)
  auto issueA = [&](int stage, int k0) {                   // k0 in elements
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[0] + 2 * k0),
                                     (__attribute__((address_space(3))) void*)(d3_sm + stage * D3_STAGE + dstA[0]), 16, 0, 0);
    if (two)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[1] + 2 * k0),
                                       (__attribute__((address_space(3))) void*)(d3_sm + stage * D3_STAGE + dstA[1]), 16, 0, 0);
  };
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  struct BRegs { f32x4 lo, hi; };
  // (the loads as statements the compiler cannot see into: with LDS transfers and register loads pending on one counter its
  //  wait-count pass assumes they may return out of order and drains the counter -- `s_waitcnt vmcnt(0)` right behind the requests
  //  just made, a memory round trip inside every third step.  They return in order; the waits below are counted by hand, and
  //  `landed` ties the registers' first use to the wait in front of it.)
  auto issueB = [&](BRegs& b, auto K0) {
    constexpr int off = decltype(K0)::value * 4;
    const float* s0_ = srcB0; const float* s1_ = srcB1;
#if defined(PAYNE_EXP_D3F) && (PAYNE_EXP_D3F & 4)       /* timing twin: the weights' loads hit one line (no port traffic to speak of) */
    const float* w_ = p_.bias;
    asm volatile("global_load_dwordx4 %0, %2, off nt\n\tglobal_load_dwordx4 %1, %2, off nt"
                 : "=&v"(b.lo), "=&v"(b.hi) : "v"(w_), "v"(s1_), "n"(off) : "memory");
#else
    asm volatile("global_load_dwordx4 %0, %2, off offset:%4 nt\n\tglobal_load_dwordx4 %1, %3, off offset:%4 nt"
                 : "=&v"(b.lo), "=&v"(b.hi) : "v"(s0_), "v"(s1_), "n"(off) : "memory");
#endif
  };
  auto landed = [&](BRegs& b) { asm volatile("" : "+v"(b.lo), "+v"(b.hi)); };
  // (pairs: one v_cvt_pk_bf16_f32 rounds two values and leaves them packed as the plane wants them; what it rounded away, exactly:
  //  x - float(part), the parts widened by a shift / a mask -- 11 vector instructions a pair, the values of split3 bit for bit)
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  auto split_pair = [](float x0, float x1, unsigned& ph, unsigned& pm, unsigned& pl) {
    f32x2_t x = {x0, x1};
    ph = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2_t));
    f32x2_t r1 = {x0 - __builtin_bit_cast(float, ph << 16), x1 - __builtin_bit_cast(float, ph & 0xffff0000u)};
    pm = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2_t));
    f32x2_t r2 = {r1[0] - __builtin_bit_cast(float, pm << 16), r1[1] - __builtin_bit_cast(float, pm & 0xffff0000u)};
    pl = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2_t));
  };
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  auto splitB = [&](const BRegs& b, int stage) {
    unsigned ph[4], pm[4], pl[4];
#if defined(PAYNE_EXP_D3F) && (PAYNE_EXP_D3F & 1)       /* timing twin (tools/exp/d3f_ablate.py): no split arithmetic */
    for (int j = 0; j < 4; ++j) { ph[j] = __builtin_bit_cast(unsigned, j < 2 ? b.lo[j] : b.hi[j]); pm[j] = ph[j]; pl[j] = ph[j]; }
#else
    split_pair(b.lo[0], b.lo[1], ph[0], pm[0], pl[0]); split_pair(b.lo[2], b.lo[3], ph[1], pm[1], pl[1]);
    split_pair(b.hi[0], b.hi[1], ph[2], pm[2], pl[2]); split_pair(b.hi[2], b.hi[3], ph[3], pm[3], pl[3]);
#endif
    unsigned char* B0 = d3_sm + stage * D3_STAGE + dstB0;
    unsigned char* B1 = d3_sm + stage * D3_STAGE + dstB1;
    *reinterpret_cast<u32x2*>(B0) = u32x2{ph[0], ph[1]}; *reinterpret_cast<u32x2*>(B0 + 8192) = u32x2{pm[0], pm[1]};
    *reinterpret_cast<u32x2*>(B0 + 2 * 8192) = u32x2{pl[0], pl[1]};
    *reinterpret_cast<u32x2*>(B1) = u32x2{ph[2], ph[3]}; *reinterpret_cast<u32x2*>(B1 + 8192) = u32x2{pm[2], pm[3]};
    *reinterpret_cast<u32x2*>(B1 + 2 * 8192) = u32x2{pl[2], pl[3]};
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int Ra = wm0 + (lane & 31), Rb = wn0 + (lane & 31), h = lane >> 5;
  const int sa = (Ra >> 2) & 3, sb = (Rb >> 2) & 3;
  struct Frag { bf16x8_t a[2][3], b[2][3]; };
  auto frags = [&](int stage, Frag& f) {
    const unsigned char* As = d3_sm + stage * D3_STAGE;
    const unsigned char* Bs = As + 3 * 4096;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = 2 * ks + h;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        f.a[ks][pl] = *reinterpret_cast<const bf16x8_t*>(As + pl * 4096 + Ra * 64 + 16 * (c ^ sa));
        f.b[ks][pl] = *reinterpret_cast<const bf16x8_t*>(Bs + pl * 8192 + Rb * 64 + 16 * (c ^ sb));
      }
    }
  };
  auto products = [&](const Frag& f, int ks) {             // smallest partial products first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][2], f.b[ks][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][1], f.b[ks][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][0], f.b[ks][2], acc, 0, 0, 0);
#if !(defined(PAYNE_EXP_D3F) && (PAYNE_EXP_D3F & 2))    /* timing twin: three of the six products */
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][1], f.b[ks][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][0], f.b[ks][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][0], f.b[ks][0], acc, 0, 0, 0);
#endif
  };
  constexpr int nk = NK;
  const int k_tail = (p.k_real > 0 ? p.k_real : p.K) - (nk - 1) * 32;
  const bool last_both = __builtin_amdgcn_readfirstlane(k_tail > 16 ? 1 : 0) != 0;
  const int col = n0 + wn0 + (lane & 31);
  const float bv = p.bias[col < p.N ? col : p.N - 1] - p.bias_shift;
  HK_STAMP(0);
  // ---- prologue: the request order of the steady state (step j requests A of j + 3, then B of j + 4), B of step 0 first
  BRegs breg[3];
  using std::integral_constant;
  issueB(breg[0], integral_constant<int, 0>{});
  issueA(0, 0); issueB(breg[1], integral_constant<int, 32>{});
  issueA(1, 32); issueB(breg[2], integral_constant<int, 64>{});
  issueA(2, 64);
  if (two) d3f_wait_vm<3 * 2 + 4>(); else d3f_wait_vm<3 * 1 + 4>();      // B of step 0 in: A0 B1 A1 B2 A2 may be outstanding
  landed(breg[0]);
  splitB(breg[0], 0);
  issueB(breg[0], integral_constant<int, 96>{});
  if (two) d3f_wait_vm<2 * 2 + 4>(); else d3f_wait_vm<2 * 1 + 4>();      // A of step 0 and B of step 1 in: A1 B2 A2 B3 may be outstanding
  landed(breg[1]);
  splitB(breg[1], 1);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  Frag f0, f1;
  frags(0, f0);
  auto head = [&](auto IT, Frag& fn, const Frag& fc) {
    constexpr int it = decltype(IT)::value;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) asm volatile("" :: "v"(fc.a[ks][pl]), "v"(fc.b[ks][pl]));
    // outstanding at most: what step it - 1 requested (A of it + 2, B of it + 3)
    constexpr int youngA = (it + 2 < nk) ? 1 : 0, youngB = (it + 3 < nk) ? 2 : 0;
    if (two) d3f_wait_vm<2 * youngA + youngB>(); else d3f_wait_vm<youngA + youngB>();
    if constexpr (it + 2 < nk) landed(breg[(it + 2) % 3]);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (lgkmcnt: my stores of B of step it + 1)
#ifndef PAYNE_NO_LOOP_STAMPS
    if (it < 13) HK_STAMP(1 + it);
#endif
    if constexpr (it + 1 < nk) frags((it + 1) % NS, fn);
    __builtin_amdgcn_sched_barrier(0);
  };
  // A wave's LDS operations queue behind its own LDS transfers (a ds_write issued after a global_load_lds does not issue until that
  // transfer has LANDED, and the matrix instructions behind it wait with it): the split's stores go BEFORE the step's requests --
  // first half of the products with the split among them, the stores, the requests, second half of the products.
  auto interleave = [&]() {                                // one matrix instruction, then some of the split's vector instructions
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x200, 6, 0);
  };
  auto step = [&](auto IT, Frag& fn, Frag& fc) {
    constexpr int it = decltype(IT)::value;
    head(IT, fn, fc);
    products(fc, 0);
    if constexpr (it + 2 < nk) { splitB(breg[(it + 2) % 3], (it + 2) % NS); interleave(); }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (it + 3 < nk) issueA((it + 3) % NS, (it + 3) * 32);
    if constexpr (it + 4 < nk) issueB(breg[(it + 4) % 3], std::integral_constant<int, (it + 4) * 32>{});
    __builtin_amdgcn_sched_barrier(0);
    if (it + 1 < nk || last_both) products(fc, 1);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto run = [&](auto self, auto IT) -> void {
    constexpr int it = decltype(IT)::value;
    if constexpr (it < nk) {
      if constexpr ((it & 1) == 0) step(IT, f1, f0); else step(IT, f0, f1);
      self(self, std::integral_constant<int, it + 1>{});
    }
  };
  run(run, std::integral_constant<int, 0>{});
  // C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  if (col < p.N) {
    const bool act_none = __builtin_amdgcn_readfirstlane(p.act == PAYNE_ACT_NONE ? 1 : 0) != 0;
    if (act_none) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < p.B) __builtin_nontemporal_store(acc[r] + bv, &p.Y[(size_t)row * p.ldy + col]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < p.B) __builtin_nontemporal_store(act_apply(acc[r] + bv, p.act), &p.Y[(size_t)row * p.ldy + col]);
      }
    }
  }
  HK_STAMP(15);
}

// ----------------------------------------------------------------------------
// The same six-product output layer for MANY tiles per CU (C5: 2048 x 65 536 outputs): 128 x 256 tiles, persistent workgroups.
// payne_dense_dma3_kernel<0, 2, false> brings in 36 KB of operand planes per 32-deep step for 64 x 128 outputs and runs at what the
// memory side delivers to a CU (13-14 B per clock from the Infinity Cache, 22 from L2: tools/exp/stage_rate.hip -- by LDS-DMA and
// through registers alike); its matrix instructions are busy 29 % of the time.  A 128 x 256 tile brings in 36 KB per 16-deep step
// for FOUR times the outputs per k -- half the bytes per product.  One 512-thread workgroup per CU walks its share of the tiles;
// eight waves as 2 x 4, a wave owns 64 x 64 outputs (four 32 x 32 accumulators).
// The 16-deep steps of ALL its tiles form one sequence through a FOUR-stage ring with three stages in flight: the requests never
// drain at a barrier, the first stages of the next tile are requested during the last steps of this one (ahead of the epilogue's
// stores), so neither a tile's first load nor its stores stop the stream.
// Tile order: see tile_at below (blocked: an XCD keeps 8 row tiles in its L2 while it sweeps its column tiles four at a time).
// Whole tiles only (B % 128 == 0, N % 256 == 0: the launch falls back to payne_dense_dma3_kernel otherwise).
// Stage: 3 planes x (128 A rows + 256 B rows) x 32 B = 36 KB (four stages: 144 KB); a 1-KiB DMA piece = 32 rows of one plane
// (two lanes per row), 36 pieces: waves 0-3 move five, waves 4-7 four; the two 16-byte chunks of tile row r are swapped where
// bit 4 of r is set: the sixteen lanes a ds_read_b128 serves together then cover sixteen different bank groups.
// ----------------------------------------------------------------------------
constexpr int B3_TM = 128, B3_TN = 256, B3_NS = 4;
constexpr int B3_A_PLANE = B3_TM * 32, B3_B_PLANE = B3_TN * 32;            // bytes per plane and stage
constexpr int b3_stage(bool h2) { return (h2 ? 2 : 3) * (B3_A_PLANE + B3_B_PLANE); }
constexpr size_t b3_lds_bytes(bool h2 = false) { return (size_t)B3_NS * b3_stage(h2); }
template <bool H2>
__global__ void __launch_bounds__(512) payne_dense_big3_kernel(DenseParams p) {
  // H2: operands as two fp16 planes, three products a block (see payne_dense_dma2h_kernel); else three bf16 planes, six products
  constexpr int NPL = H2 ? 2 : 3, B3_STAGE = b3_stage(H2);
  typedef typename std::conditional<H2, f16x8_t, bf16x8_t>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char b3_sm[];
  const int ntiles = p.grid_m * p.grid_n;
  // workgroup b sits on XCD b & 7 (round-robin dispatch); XCD x takes tiles [x per, (x + 1) per), its workgroups stride through them
  const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, nx = (gridDim.x + 7 - xcd) >> 3;     // my index among the XCD's nx workgroups
  const int per = (ntiles + 7) >> 3, t_lo = xcd * per, t_hi = (t_lo + per < ntiles) ? t_lo + per : ntiles;
  const int my_n = (t_lo + jx < t_hi) ? (t_hi - t_lo - jx + nx - 1) / nx : 0;
  if (my_n <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 2) * 64, wn0 = (wave & 3) * 64;
  const bool five = !H2 && wave < 4;                       // pieces this wave moves per stage: 12 NPL of them -- 5 (waves 0-3) or 4; 3 each with two planes
  // the pieces this wave moves per stage: plane, 32-row block, and where they land
  int prow[5], pdst[5], ppl[5];
  bool pA[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    int q = H2 ? wave * 3 + j : (five ? wave * 5 + j : 20 + (wave - 4) * 4 + j);
    if (q > 12 * NPL - 1) q = 12 * NPL - 1;                // (slots past a wave's count: never issued)
    pA[j] = q < 4 * NPL;
    ppl[j] = pA[j] ? q >> 2 : (q - 4 * NPL) >> 3;
    const int blk = pA[j] ? (q & 3) : ((q - 4 * NPL) & 7);
    prow[j] = 32 * blk + (lane >> 1);
    pdst[j] = pA[j] ? ppl[j] * B3_A_PLANE + blk * 1024 : NPL * B3_A_PLANE + ppl[j] * B3_B_PLANE + blk * 1024;
  }
  const unsigned char* src[5];
  auto set_src = [&](int t) {
    const int m0 = (t % p.grid_m) * B3_TM, n0 = (t / p.grid_m) * B3_TN;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int row = prow[j];
      const int c = (lane & 1) ^ ((row >> 4) & 1);         // which 16-byte chunk of the row belongs in this lane's slot
      if (pA[j]) {
        const int r = (m0 + row < p.B) ? m0 + row : p.B - 1;
        src[j] = reinterpret_cast<const unsigned char*>(p.Xp + (size_t)ppl[j] * p.plane_x + (size_t)r * p.ldp) + 16 * c;
      } else {
        const int r = (n0 + row < p.N) ? n0 + row : p.N - 1;
        src[j] = reinterpret_cast<const unsigned char*>(p.Wp + (size_t)ppl[j] * p.plane_w + (size_t)r * p.K) + 16 * c;
      }
    }
  };
  auto issue = [&](int stage, int k0) {                    // k0 in elements (2 bytes each)
#pragma unroll
    for (int j = 0; j < 5; ++j)
      if (H2 ? j < 3 : (j < 4 || five))
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + 2 * k0),
                                         (__attribute__((address_space(3))) void*)(b3_sm + stage * B3_STAGE + pdst[j]), 16, 0, 0);
  };
  f32x16 acc[2][2];
  const int h = lane >> 5;
  int oa[2], ob[2];                                        // byte offsets of this lane's fragment rows inside a plane
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int Ra = wm0 + 32 * i + (lane & 31), Rb = wn0 + 32 * i + (lane & 31);
    oa[i] = Ra * 32 + 16 * (h ^ ((Ra >> 4) & 1));
    ob[i] = Rb * 32 + 16 * (h ^ ((Rb >> 4) & 1));
  }
  struct Frag { frag_t a[2][NPL], b[2][NPL]; };             // one 16-deep matrix step: two row blocks, two column blocks, NPL planes
  auto frags = [&](int stage, Frag& f) {
    const unsigned char* As = b3_sm + stage * B3_STAGE;
    const unsigned char* Bs = As + NPL * B3_A_PLANE;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        f.a[i][pl] = *reinterpret_cast<const frag_t*>(As + pl * B3_A_PLANE + oa[i]);
        f.b[i][pl] = *reinterpret_cast<const frag_t*>(Bs + pl * B3_B_PLANE + ob[i]);
      }
  };
  auto products = [&](const Frag& f) {                     // smallest partial products first, block by block
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 a = acc[i][j];
        if constexpr (H2) {
          a = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i][1], f.b[j][0], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i][0], f.b[j][1], a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[i][0], f.b[j][0], a, 0, 0, 0);
        } else {
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][2], f.b[j][0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][1], f.b[j][1], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][2], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][1], f.b[j][0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][1], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][0], f.b[j][0], a, 0, 0, 0);
        }
        acc[i][j] = a;
      }
  };
  // 16-deep steps that hold real columns (the zero-padded tail of the padded width is skipped)
  const int nk = ((p.k_real > 0 ? p.k_real : p.K) + 15) / 16;
  const bool act_none = __builtin_amdgcn_readfirstlane(p.act == PAYNE_ACT_NONE ? 1 : 0) != 0;
  // The n-th tile of this workgroup.  Blocked order (whole XCD shares: 32 workgroups per XCD, row tiles a multiple of 8, the XCD's
  // column tiles a multiple of 4): at any time the XCD works on 8 row tiles x 4 column tiles -- 3.9 MB of operand planes, the least
  // for 32 tiles --, and keeps the SAME 8 row tiles while it sweeps its column tiles four at a time, so their activation planes
  // (2 MB) stay in its L2 and only the weights stream through.  Otherwise: the XCD's tiles in order, row tiles fastest.
  const int xcols = p.grid_n >> 3;                          // column tiles per XCD (blocked order only)
  const bool blocked = nx == 32 && (p.grid_n & 7) == 0 && (p.grid_m & 7) == 0 && (xcols & 3) == 0 && per * 8 == ntiles;
  auto tile_at = [&](int n) {
    if (!blocked) return t_lo + jx + n * nx;
    const int sweeps = xcols >> 2;                          // steps of four column tiles per row group
    const int g = n / sweeps, c4 = n - g * sweeps;
    const int row = g * 8 + (jx & 7), colt = xcd * xcols + c4 * 4 + (jx >> 3);
    return colt * p.grid_m + row;
  };
  // the request stream: step q of the whole sequence = (tile q / nk, k-step q % nk); three steps ahead of the step being multiplied
  const int total = my_n * nk;
  int rq = 0, rq_n = 0, rq_it = 0;                          // next step to request: its number, tile ordinal, k-step
  set_src(tile_at(0));
  auto request_next = [&]() {
    if (rq >= total) return;
    issue(rq & (B3_NS - 1), rq_it * 16);
    ++rq;
    if (++rq_it == nk) { rq_it = 0; ++rq_n; if (rq_n < my_n) set_src(tile_at(rq_n)); }
  };
  request_next(); request_next(); request_next();
  int s = 0;                                               // running step count: stage = s & 3
  int since_stores = 3;                                    // steps since a tile's 64 stores were issued (>= 3: none among the waited-for)
  for (int n = 0; n < my_n; ++n) {
    const int t = tile_at(n);
    const int m0 = (t % p.grid_m) * B3_TM, n0 = (t / p.grid_m) * B3_TN;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bv[2], rs[2] = {1.f, 1.f};
    for (int it = 0; it < nk; ++it, ++s) {
      // This step's stage has landed once only what was issued after it is outstanding: the requests of (up to) two younger
      // stages -- and, for three steps after an epilogue, its 64 stores, which sit among them in issue order (the counter holds 63
      // at most: "all but the youngest 63" then covers this stage's pieces, at the price of waiting for a few of the stores).
      const int younger = (rq - s - 1);                    // stages requested after this one: 2, fewer at the very end
      if (since_stores < 3) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
      else if (younger >= 2) { if (H2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else if (five) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
      else if (younger == 1) { if (H2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else if (five) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ++since_stores;
      asm volatile("s_barrier" ::: "memory");              // everybody's pieces landed; everybody finished the step before
      Frag f;
      frags(s & (B3_NS - 1), f);
      __builtin_amdgcn_sched_barrier(0);
      request_next();                                      // into the buffer the step before consumed
      if (it == 0) {                                       // the epilogue's bias (younger than the request: long landed by then)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = n0 + wn0 + 32 * j + (lane & 31);
          bv[j] = p.bias[col < p.N ? col : p.N - 1] - p.bias_shift;
          if constexpr (H2) rs[j] = p.rscale[col < p.N ? col : p.N - 1];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      products(f);
    }
    // C/D map of a 32x32 block: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn0 + 32 * j + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          // (unconditional: the launch guarantees whole tiles -- B % 128 == 0, N % 256 == 0 --, so every wave issues exactly 64
          //  stores here, which is what the vmcnt(63) above counts on)
          const float v = H2 ? __builtin_fmaf(acc[i][j][r], rs[j], bv[j]) : acc[i][j][r] + bv[j];
          __builtin_nontemporal_store(act_none ? v : act_apply(v, p.act), &p.Y[(size_t)row * p.ldy + col]);
        }
      }
    since_stores = 0;
  }
}


// ----------------------------------------------------------------------------
// Hidden layers, workgroup form: one 256-thread group per 32x32 output tile, the whole K
// extent (<= 320 per chunk) of both operands staged in LDS by coalesced f32x4_t loads issued
// together (one L2 latency), then the four waves split K between them (v_mfma_f32_16x16x4_f32,
// 2x2 tiles each) and their partial tiles are summed through LDS.  With FUSE_L0 the A tile is
// produced in place from theta (label encoding + first layer + activation).
// ----------------------------------------------------------------------------
constexpr int HK_KC = 304;          // K chunk (a multiple of 16; with the pitch below two workgroups' tiles are 79.9 KB: two fit a CU)
constexpr int HK_PITCH = 312;       // 8*odd floats: conflict-free ds_read_b128 for the 16-row x 4-offset lane map
constexpr size_t HK_LDS_BYTES = (size_t)(2 * 32 * HK_PITCH + 32 * PAYNE_MAX_LABELS) * sizeof(float);
static_assert(sed_tile_lds_bytes() <= HK_LDS_BYTES, "the photometric tile runs in the hidden-layer launch's LDS");

// Workgroups past the GEMM tiles (first-layer launch only) compute the per-candidate records of
// the post kernel (prep_candidate: Doppler / rotation / instrument scalars, mask counts, R-stage
// window), one thread per candidate, on compute units the 160 GEMM tiles leave idle.
// ... and, for joint spectrum + photometry likelihoods, workgroups past those run the photometric nets (sed_tile, sed_core.hpp:
// one (filter, 64 candidates) tile each): the magnitudes the post kernel's chi^2_SED needs, without a launch of their own.
struct PrepArgs {
  PostTables T;
  CandState* out;            // [B] (null: no records from this launch)
  double instr_factor;
  int n_gemm;                // workgroups that are GEMM tiles
  int n_prep;                // ... that write per-candidate records (256 candidates each)
  PhotTables P;              // photometric nets (sed_mags != null)
  double* sed_mags;          // [B][F] magnitudes of this batch (null: no photometry in this launch)
  int sed_off, sed_photscale;   // theta column of the photometric block; the log(A) parametrisation
  int sed_cb;                // candidates per photometric tile (<= kSedCandsMax, chosen so that the tiles fit the idle compute units)
  // the sampler's walk: workgroups past those make the NEXT chain step's proposal for both outcomes of the one this batch evaluates
  // (rwalk_spec_wave, sampler_core.hpp; eight chains per workgroup)
  const WalkTail* spec_walk; WalkState spec_w; int spec_step, n_spec, n_sed;
  // rows in the frequency domain of a RESAMPLED model grid (freq_rows with maps): a candidate that does not rotate needs the pixels
  // themselves -- the thread that writes its record says so (*rot_flag = rot_seq) and the output layer and the post kernel of this
  // batch, which read the word at their start, run on pixels (null: not such a launch)
  unsigned long long* rot_flag; unsigned long long rot_seq;
};

// The epilogue of a hidden-layer tile: the wave's 16 x 16 quadrant (C/D map: column r, rows 4 g + q) stored as the fp32 tile or as the
// output layer's operand planes.  Neighbouring lanes hold neighbouring columns: lane pairs swap half of their rows (DPP), so that every
// lane ends up with two adjacent columns of two rows -- 8-byte stores of the fp32 tile, 4-byte stores of each plane, without LDS.
__device__ __forceinline__ void hk_store_quadrant(const DenseParams& p, const float (&y)[4], int m0, int n0, int qi, int qj, int r, int g) {
  {
    const bool pairs_ok = ((p.ldy | n0) & 1) == 0 && (!p.Yp || (p.ldp & 1) == 0);
    const int colq = n0 + 16 * qj + r, row0 = m0 + 16 * qi + 4 * g;
    if (pairs_ok) {
      const bool oddl = (r & 1) != 0;
      // even lane keeps rows 0, 1 and gets the odd lane's; odd lane keeps rows 2, 3 and gets the even lane's
      const float s0 = oddl ? y[0] : y[2], s1 = oddl ? y[1] : y[3];
      const float t0 = __shfl_xor(s0, 1), t1 = __shfl_xor(s1, 1);
      float pa_[2], pb_[2];                                          // (column, column + 1) of the lane's two rows
      pa_[0] = oddl ? t0 : y[0]; pb_[0] = oddl ? y[2] : t0;
      pa_[1] = oddl ? t1 : y[1]; pb_[1] = oddl ? y[3] : t1;
      const int col = colq & ~1;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int row = row0 + (oddl ? 2 : 0) + e;
        if (row < p.B && col < p.N) {
          const bool two = col + 1 < p.N;
          if (!p.Yp) {
            float* yo = &p.Y[(size_t)row * p.ldy + col];
            if (two) *reinterpret_cast<float2*>(yo) = make_float2(pa_[e], pb_[e]); else yo[0] = pa_[e];
          } else if (p.yp_half) {                                    // two fp16 parts for payne_dense_dma2h_kernel
            unsigned q1, q2;
            split2h_pair(pa_[e], pb_[e], p.yp_scale, q1, q2);
            const size_t o = (size_t)row * p.ldp + col;
            if (two) {
              __builtin_nontemporal_store(q1, reinterpret_cast<unsigned*>(&p.Yp[o]));
              __builtin_nontemporal_store(q2, reinterpret_cast<unsigned*>(&p.Yp[p.plane_y + o]));
            } else { p.Yp[o] = (unsigned short)(q1 & 0xffffu); p.Yp[p.plane_y + o] = (unsigned short)(q2 & 0xffffu); }
          } else {                                                   // three bf16 parts for payne_dense_dma3_kernel (its only reader)
            unsigned short h3[2], m3[2], l3[2];
            split3(pa_[e], h3[0], m3[0], l3[0]);
            split3(pb_[e], h3[1], m3[1], l3[1]);
            const size_t o = (size_t)row * p.ldp + col;
            if (two) {                                               // (streamed: the next reader is another XCD)
              __builtin_nontemporal_store((unsigned)h3[0] | ((unsigned)h3[1] << 16), reinterpret_cast<unsigned*>(&p.Yp[o]));
              __builtin_nontemporal_store((unsigned)m3[0] | ((unsigned)m3[1] << 16), reinterpret_cast<unsigned*>(&p.Yp[p.plane_y + o]));
              __builtin_nontemporal_store((unsigned)l3[0] | ((unsigned)l3[1] << 16), reinterpret_cast<unsigned*>(&p.Yp[2 * p.plane_y + o]));
            } else { p.Yp[o] = h3[0]; p.Yp[p.plane_y + o] = m3[0]; p.Yp[2 * p.plane_y + o] = l3[0]; }
          }
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = row0 + q;
        if (row < p.B && colq < p.N) {
          if (!p.Yp) p.Y[(size_t)row * p.ldy + colq] = y[q];
          else if (p.yp_half) {
            unsigned q1, q2;
            split2h_pair(y[q], 0.f, p.yp_scale, q1, q2);
            const size_t o = (size_t)row * p.ldp + colq;
            p.Yp[o] = (unsigned short)(q1 & 0xffffu); p.Yp[p.plane_y + o] = (unsigned short)(q2 & 0xffffu);
          } else {
            unsigned short h3, m3, l3;
            split3(y[q], h3, m3, l3);
            const size_t o = (size_t)row * p.ldp + colq;
            p.Yp[o] = h3; p.Yp[p.plane_y + o] = m3; p.Yp[2 * p.plane_y + o] = l3;
          }
        }
      }
    }
  }
}

// ----------------------------------------------------------------------------
// hk_tile_h2: the first launch of a net (label encoding + first layer + second layer) with the second layer's products on fp16 PAIRS
// (split2h above: three v_mfma_f32_16x16x32_f16 a 32-deep step instead of eight v_mfma_f32_16x16x4_f32 -- 30 matrix instructions a
// wave for K = 300 instead of 76).  Weight tile: two fp16 planes [32][304] straight into LDS (38 transfers of 1 KB,
// the planes split and row-scaled at payne_ctx_create).  First layer TRANSPOSED on the matrix cores (operands swapped: D[unit][candidate]),
// so that a lane ends up with FOUR CONSECUTIVE units of one candidate row -- split in registers, one 8-byte LDS store a plane.  Rows of
// a plane are 608 bytes = 38 chunks of 16 bytes apart: 38 = 6 (mod 16) puts the sixteen lanes of every lane group of a fragment read
// (ds_read_b128: lane (r, g) reads chunk 4 s + g of row r) on sixteen different 16-byte bank groups.
// ----------------------------------------------------------------------------
constexpr int HK2_PB = 608, HK2_PLANE = 32 * HK2_PB, HK2_K = 304;
static_assert(4 * HK2_PLANE <= (int)HK_LDS_BYTES, "four planes in the hidden-layer launch's LDS");
// The matrix phase and the epilogue of a tile whose operands sit in LDS as fp16 planes (hk_tile_h2, hk_tile_h2x): the wave's 16 x 16
// quadrant over ten 32-deep steps (the tenth holds k = 288 .. 303 only), three products each, smallest first.
__device__ __forceinline__ void hk_h2_quadrant(const DenseParams& p, const unsigned char* As, const unsigned char* Bs, float bias1, float rs1,
                                               int m0, int n0, int qi, int qj, int r, int g) {
  // ---- the wave's quadrant: nine 32-deep steps and one 16-deep, three products each, smallest first ---------------------------
  f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const unsigned char* ar = As + (16 * qi + r) * HK2_PB + 16 * g;
  const unsigned char* br = Bs + (16 * qj + r) * HK2_PB + 16 * g;
  constexpr int NS32 = HK2_K / 32;                                     // 9
  f16x8_t fa[NS32][2], fb[NS32][2];
#pragma unroll
  for (int sx = 0; sx < NS32; ++sx)
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      fa[sx][pl] = *reinterpret_cast<const f16x8_t*>(ar + pl * HK2_PLANE + 64 * sx);
      fb[sx][pl] = *reinterpret_cast<const f16x8_t*>(br + pl * HK2_PLANE + 64 * sx);
    }
  // the tenth step holds k = 288 .. 303 only: lanes g = 0, 1 read their chunks, lanes g = 2, 3 (k = 304 .. 319: past the row) hold zeros
  // (the 16-deep v_mfma_f32_16x16x16f16 in its place gave wrong accumulator halves now and then: read before its last pass had landed)
  f16x8_t ta[2], tb[2];
  {
    const int gc = g < 2 ? g : 0;
    const f16x8_t zero8 = (f16x8_t)(_Float16)0;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const f16x8_t va = *reinterpret_cast<const f16x8_t*>(As + pl * HK2_PLANE + (16 * qi + r) * HK2_PB + 64 * NS32 + 16 * gc);
      const f16x8_t vb = *reinterpret_cast<const f16x8_t*>(Bs + pl * HK2_PLANE + (16 * qj + r) * HK2_PB + 64 * NS32 + 16 * gc);
      ta[pl] = g < 2 ? va : zero8; tb[pl] = g < 2 ? vb : zero8;
    }
  }
#pragma unroll
  for (int sx = 0; sx < NS32; ++sx) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[sx][1], fb[sx][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[sx][0], fb[sx][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[sx][0], fb[sx][0], acc, 0, 0, 0);
  }
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ta[1], tb[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ta[0], tb[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ta[0], tb[0], acc, 0, 0, 0);
  HK_STAMP(4);
  {
    float y[4];
    const float bsh = bias1 - p.bias_shift;
#pragma unroll
    for (int q = 0; q < 4; ++q) y[q] = act_apply(__builtin_fmaf(acc[q], rs1, bsh), p.act);
    hk_store_quadrant(p, y, m0, n0, qi, qj, r, g);
  }
}

// Register loads the compiler cannot see into (hk_tile_h2): with LDS transfers and register loads pending on one counter its wait-count
// pass drains the counter before the first loaded value is used -- the first layer then starts when the weight planes' 39 KB have
// landed, not when its own few values have.  Loads return in order: the wait is counted by hand (`hk2_small_landed`).
// (`base`: the same for every lane, `off`: the lane's byte offset -- the scalar-base form, no 64-bit vector arithmetic per address)
// `s_nop 4`: a vector-memory instruction that reads a scalar register a VECTOR instruction has just written (v_readlane / v_readfirstlane:
// how the compiler brings back a scalar it parked in a vector register's lanes when scalars run short) needs five wait states in between, and
// the compiler's hazard recognizer does not look inside an asm statement.  Found with rocgdb's precise memory faults in the stamped
// diagnostic twin (more scalars live: `v_readlane_b32 s11, v150, 3` right in front of the labels' load, which then went through a stale
// base); the production build happened to have its bases in preloaded registers.
__device__ __forceinline__ void gload(float& v, const float* base, unsigned off) { asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory"); }
__device__ __forceinline__ void gload(double& v, const double* base, unsigned off) { asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory"); }
__device__ __forceinline__ void gload(f32x4_t& v, const float* base, unsigned off) { asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory"); }

// NW: the workgroup's waves (4 | 8).  Eight waves share the requests, the first layer and its LDS stores (three unit blocks a wave instead of
// five: the phase is instruction-bound); the matrix phase and the epilogue are the first four waves' (one 16 x 16 quadrant each: LDS-read-bound,
// more waves would read the same bytes), the others leave after the barrier.
template <int NL, int NW>
__device__ __forceinline__ void hk_tile_h2(DenseParams& p, int tile, float* hk_sm, const int tid) {
  HK_STAMP(0);
  unsigned char* As = reinterpret_cast<unsigned char*>(hk_sm);
  unsigned char* Bs = As + 2 * HK2_PLANE;
  const int tm = tile / p.grid_n, tn = tile - tm * p.grid_n;
  const int m0 = tm * 32, n0 = tn * 32;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, g = lane >> 4;
  const int qi = wave >> 1, qj = wave & 1;
  // The tile's first requests are ~50 vector instructions, not ~480: every address is a scalar base + a 32-bit lane offset, the weight
  // transfers' (row, chunk) pairs follow from the first by a recurrence and serve both planes.
  float bias1, rs1;
  {
    const int c0 = n0 + 16 * qj + r, cc = c0 < p.N ? c0 : p.N - 1;
    gload(bias1, p.bias, 4u * (unsigned)cc); gload(rs1, p.rs1, 4u * (unsigned)cc);
  }
  static_assert(NW == 4 || NW == 8, "waves a workgroup");
  constexpr int NLG = (NL + 3) / 4, MAXT = (HK2_K / 16 + NW - 1) / NW;   // label groups of four; unit blocks of 16 a wave (5 | 3)
  constexpr int ntile = HK2_K / 16;                                   // 19
  // labels of the candidates (B operand of the transposed first layer: lane (r, g) = xhat[candidate 16 i + r][label 4 lg + g])
  double xl[2][NLG];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (m0 + 16 * i + r < p.B) ? m0 + 16 * i + r : p.B - 1;
#pragma unroll
    for (int lg = 0; lg < NLG; ++lg) { const int d = 4 * lg + g; gload(xl[i][lg], p.theta, 8u * ((unsigned)row * (unsigned)p.ld_theta + (unsigned)(d < 4 ? d : 6))); }
  }
  // first-layer weights (A operand: lane (r, g) = W0[unit 16 t + r][label 4 lg + g]) and biases (four units 16 t + 4 g + q a lane)
  float w0t[MAXT][NLG];
  f32x4_t bz4[MAXT];
#pragma unroll
  for (int tt = 0; tt < MAXT; ++tt) {
    const int tcol = wave + NW * tt, tc = tcol < ntile ? tcol : 0;
    const int k = 16 * tc + r, kq = k < p.K0 ? k : p.K0 - 1;
#pragma unroll
    for (int lg = 0; lg < NLG; ++lg) {
      const int d = 4 * lg + g;
      gload(w0t[tt][lg], p.W0, 4u * ((unsigned)kq * (unsigned)p.n_labels + (unsigned)(d < p.n_labels ? d : 0)));     // (clamped address, masked below)
    }
    {                                                                 // (units past the layer's width are masked below: any values do)
      const int u0 = 16 * tc + 4 * g, uq = (u0 + 4 <= p.K0) ? u0 : ((p.K0 - 4) & ~3);
      gload(bz4[tt], p.b0, 4u * (unsigned)(uq > 0 ? uq : 0));
    }
  }
  // the weight tile: 2 planes x 19 transfers of 64 consecutive 16-byte chunks (chunk sl = (row sl / 38, chunk sl % 38)); wave w moves
  // transfers w, w + NW, ..: a step of NW transfers is 64 NW chunks = 6 rows + 28 chunks on (four waves; 13 rows + 18 chunks: eight)
  constexpr int NCH = HK2_PB / 16, NTR = 32 * NCH / 64;                 // 38 chunks a row, 19 transfers a plane
  {
    constexpr int DR = 64 * NW / NCH, DC = 64 * NW - DR * NCH;
    static_assert(32 * NCH % 64 == 0 && NTR <= NW * MAXT && DC < NCH, "whole transfers; the recurrence's constants");
    unsigned voff[MAXT];
    {
      const int sl = 64 * wave + lane;
      int rr = sl / NCH, c = sl - rr * NCH;
#pragma unroll
      for (int q = 0; q < MAXT; ++q) {
        const int nr = (n0 + rr < p.N) ? n0 + rr : p.N - 1;
        voff[q] = (unsigned)nr * (unsigned)(2 * HK2_K) + 16u * (unsigned)c;
        c += DC; rr += DR;
        if (c >= NCH) { c -= NCH; rr += 1; }
      }
    }
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(p.Wh);
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const unsigned char* wpl = wb + (size_t)pl * p.plane_wh * 2;
#pragma unroll
      for (int q = 0; q < MAXT; ++q) {
        const int jj = wave + NW * q;
        if (q + 1 < MAXT || jj < NTR)                                   // (scalar: the last waves move one transfer a plane fewer)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wpl + voff[q]),
                                           (__attribute__((address_space(3))) void*)(Bs + pl * HK2_PLANE + 1024 * jj), 16, 0, 0);
      }
    }
  }
  // The register loads above are older than this wave's transfers (2 MAXT of them, two fewer for the last waves): they have landed once
  // only that many operations are outstanding.  Every value named behind the wait, before anything is done with it.
  if (wave + NW * (MAXT - 1) < NTR) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * MAXT) : "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * MAXT - 2) : "memory");
  asm volatile("" : "+v"(bias1), "+v"(rs1));
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int lg = 0; lg < NLG; ++lg) asm volatile("" : "+v"(xl[i][lg]));
#pragma unroll
  for (int tt = 0; tt < MAXT; ++tt) {
#pragma unroll
    for (int lg = 0; lg < NLG; ++lg) asm volatile("" : "+v"(w0t[tt][lg]));
    asm volatile("" : "+v"(bz4[tt]));
  }
#pragma unroll
  for (int tt = 0; tt < MAXT; ++tt) {
    const int tcol = wave + NW * tt, tc = tcol < ntile ? tcol : 0;
    const bool klive = 16 * tc + r < p.K0;
#pragma unroll
    for (int lg = 0; lg < NLG; ++lg) w0t[tt][lg] = (4 * lg + g < p.n_labels && klive) ? w0t[tt][lg] : 0.f;
  }
  HK_STAMP(1);
  float xa[2][NLG];
#pragma unroll
  for (int lg = 0; lg < NLG; ++lg) {
    double xm = p.xmin[4 * lg < PAYNE_MAX_LABELS ? 4 * lg : 0], xdn = p.xden[4 * lg < PAYNE_MAX_LABELS ? 4 * lg : 0];
#pragma unroll
    for (int e = 1; e < 4; ++e) {
      if (4 * lg + e < PAYNE_MAX_LABELS) { xm = (g == e) ? p.xmin[4 * lg + e] : xm; xdn = (g == e) ? p.xden[4 * lg + e] : xdn; }
    }
    const bool dl = 4 * lg + g < p.n_labels;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      xa[i][lg] = (dl && m0 + 16 * i + r < p.B) ? (float)((xl[i][lg] - xm) / xdn - 0.5) : 0.f;
  }
  HK_STAMP(2);
  // first layer, activation, split: two dwords a plane for every (unit block, candidate block) of this wave
  unsigned pk[MAXT][2][2][2];                                           // [block][candidate block][plane][dword]
  const float s0 = p.a0_scale;
  auto first_layer = [&](auto actf) {
#pragma unroll
    for (int tt = 0; tt < MAXT; ++tt) {
      const int tcol = wave + NW * tt;
      if (tcol < ntile) {                                             // (wave-uniform)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          f32x4_t z = bz4[tt];
#pragma unroll
          for (int lg = 0; lg < NLG; ++lg) z = __builtin_amdgcn_mfma_f32_16x16x4f32(w0t[tt][lg], xa[i][lg], z, 0, 0, 0);
          float yv[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) yv[q] = ((16 * tcol + 4 * g + q) < p.K0) ? actf(z[q]) : 0.f;
          split2h_pair(yv[0], yv[1], s0, pk[tt][i][0][0], pk[tt][i][1][0]);
          split2h_pair(yv[2], yv[3], s0, pk[tt][i][0][1], pk[tt][i][1][1]);
        }
      }
    }
  };
  if (p.act0 == PAYNE_ACT_LRELU) first_layer([](float z) { return lrelu01(z); });
  else if (p.act0 == PAYNE_ACT_SIGMOID) first_layer([](float z) { return sigmoid_f32(z); });
  else first_layer([](float z) { return z; });
  HK_STAMP(6);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // my pieces of the weight planes have landed
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int tt = 0; tt < MAXT; ++tt) {
    const int tcol = wave + NW * tt;
    if (tcol < ntile) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          *reinterpret_cast<u32x2_t*>(As + pl * HK2_PLANE + (16 * i + r) * HK2_PB + (16 * tcol + 4 * g) * 2) = u32x2_t{pk[tt][i][pl][0], pk[tt][i][pl][1]};
    }
  }
  __syncthreads();
  HK_STAMP(3);
  if (NW > 4 && wave >= 4) return;                                      // (no barrier below)
  hk_h2_quadrant(p, As, Bs, bias1, rs1, m0, n0, qi, qj, r, g);
  HK_STAMP(5);
}

// hk_tile_h2x: a hidden layer past the second on fp16 pairs -- both operand tiles are planes in memory (the activations written so
// by the launch before: Xp, pitch 304; the weights split at payne_ctx_create): 76 transfers of 1 KB, the matrix phase, the epilogue.
__device__ __forceinline__ void hk_tile_h2x(DenseParams& p, int tile, float* hk_sm, const int tid) {
  HK_STAMP(0);
  unsigned char* As = reinterpret_cast<unsigned char*>(hk_sm);
  unsigned char* Bs = As + 2 * HK2_PLANE;
  const int tm = tile / p.grid_n, tn = tile - tm * p.grid_n;
  const int m0 = tm * 32, n0 = tn * 32;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, g = lane >> 4;
  const int qi = wave >> 1, qj = wave & 1;
  float bias1, rs1;
  {
    const int c0 = n0 + 16 * qj + r, cc = c0 < p.N ? c0 : p.N - 1;
    bias1 = p.bias[cc]; rs1 = p.rs1[cc];
  }
  // four planes (the weights' two, the activations' two) x 19 transfers of 64 consecutive 16-byte chunks; wave w moves transfers w, w + 4, ..
  // of every plane (five, the last wave four).  A chunk's (row, column) follows from the wave's first by a recurrence -- a step of four
  // transfers is 256 chunks = 6 rows + 28 chunks on -- and serves all four planes: scalar bases + 32-bit lane offsets (as first written:
  // a division and 64-bit address arithmetic per transfer, ~480 vector instructions in front of the tile's only wait).
  constexpr int NCH = HK2_PB / 16, NTR = 32 * NCH / 64, MAXQ = (NTR + 3) / 4;      // 38 chunks a row, 19 transfers a plane
  static_assert(NCH == 38 && NTR == 19, "the recurrence's constants");
  unsigned voffw[MAXQ], voffx[MAXQ];
  {
    const int sl = 64 * wave + lane;
    int rr = sl / NCH, c = sl - rr * NCH;
    const int topw = p.N - 1 - n0, topx = p.B - 1 - m0;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
      voffw[q] = (unsigned)(n0 + (rr < topw ? rr : topw)) * (unsigned)(2 * HK2_K) + 16u * (unsigned)c;
      voffx[q] = (unsigned)(m0 + (rr < topx ? rr : topx)) * (unsigned)(2 * HK2_K) + 16u * (unsigned)c;
      c += 256 - 6 * NCH; rr += 6;
      if (c >= NCH) { c -= NCH; rr += 1; }
    }
  }
#pragma unroll
  for (int pq = 0; pq < 4; ++pq) {                                      // 0, 1: the weights' planes; 2, 3: the activations'
    const int pl = pq & 1;
    const unsigned char* base = pq < 2 ? reinterpret_cast<const unsigned char*>(p.Wh) + 2 * (size_t)pl * p.plane_wh
                                       : reinterpret_cast<const unsigned char*>(p.Xp) + 2 * (size_t)pl * p.plane_x;
    unsigned char* dstp = (pq < 2 ? Bs : As) + pl * HK2_PLANE;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
      const int jj = wave + 4 * q;
      if (q + 1 < MAXQ || jj < NTR)                                     // (scalar)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (pq < 2 ? voffw[q] : voffx[q])),
                                         (__attribute__((address_space(3))) void*)(dstp + 1024 * jj), 16, 0, 0);
    }
  }
  HK_STAMP(1);
  HK_STAMP(2);
  HK_STAMP(6);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // my pieces have landed (and the bias values are here)
  __syncthreads();
  HK_STAMP(3);
  hk_h2_quadrant(p, As, Bs, bias1, rs1, m0, n0, qi, qj, r, g);
  HK_STAMP(5);
}

// One 32 x 32 tile of a hidden layer by the first 256 threads of the workgroup (`tile`: index in the launch's grid_m x grid_n).
template <bool FUSE_L0, int NL>
__device__ __forceinline__ void hk_tile(DenseParams& p, int tile, float* hk_sm, const int tid) {
  HK_STAMP(0);
  float* As = hk_sm;
  float* Bs = As + 32 * HK_PITCH;
  float* Xh = Bs + 32 * HK_PITCH;
  const int tm = tile / p.grid_n, tn = tile - tm * p.grid_n;
  const int m0 = tm * 32, n0 = tn * 32;
  const int lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
  // Wave w owns the 16 x 16 quadrant (rows 16 (w >> 1), columns 16 (w & 1)) of the tile over the WHOLE K extent: no partial tiles
  // to sum through LDS afterwards (as first written the four waves split K: 16 ds_write + a barrier + 8 ds_read a thread at the
  // end of every workgroup's life).  Two accumulators take turns, so that consecutive matrix instructions never wait for each other.
  const int qi = wave >> 1, qj = wave & 1;
  f32x4_t acc[2];
  acc[0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; acc[1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const f32x4_t z4 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // the epilogue's bias value (lane -> column 16 qj + r of the tile): requested now -- asked for in the epilogue it is one more
  // memory round trip at the end of the workgroup's life
  float bias1;
  {
    const int c0 = n0 + 16 * qj + r;
    bias1 = p.bias[c0 < p.N ? c0 : p.N - 1];
  }

  // one K chunk; the usual single-chunk case (K <= 320) is called outside any loop: around a loop the
  // compiler's wait-count bookkeeping turns conservative (a vmcnt(0) right after the first load)
  auto chunk = [&](const int kc) {
    const int kn = (p.K - kc < HK_KC) ? (p.K - kc) : HK_KC;        // multiple of 4
    const int kn16 = (kn + 15) & ~15;
    const int nk4 = kn16 >> 2;
    // ---- both tiles straight into LDS (layers past the second; K <= HK_KC, padded operands) ------------------------------
    // The tile [32][HK_PITCH] is 39 x 1 KB: one global_load_lds_dwordx4 of a wave lands 64 consecutive 16-byte chunks, lane l of
    // transfer j bringing chunk s = 64 j + l = (row s / 78, k 4 (s % 78)) -- the SAME layout the register-staged form stores, its
    // pad columns (k >= 304) filled with zeros of the operands' own padding instead of being skipped.  No staging registers, no
    // 24 ds_write_b128 a thread; rows past the operand's end are clamped (their outputs are never stored).
    bool staged = false;
    if constexpr (!FUSE_L0) {
      if (p.Wd != nullptr) {                                        // (uniform)
        static_assert((32 * HK_PITCH / 4) % 64 == 0, "whole transfers");
        constexpr int NCH = HK_PITCH / 4, NTR = 32 * NCH / 64;       // chunks a row (78), transfers a tile (39)
#pragma unroll
        for (int j0 = 0; j0 < NTR; j0 += 4) {
          const int j = j0 + wave;
          if (j < NTR) {                                            // (wave-uniform)
            const int sl = 64 * j + lane, rr = sl / NCH, c4 = sl - rr * NCH;
            const int nr = (n0 + rr < p.N) ? n0 + rr : p.N - 1, mr = (m0 + rr < p.B) ? m0 + rr : p.B - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.Wd + (size_t)nr * p.ldwd + 4 * c4),
                                             (__attribute__((address_space(3))) void*)(Bs + 256 * j), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.X + (size_t)mr * p.ldx + 4 * c4),
                                             (__attribute__((address_space(3))) void*)(As + 256 * j), 16, 0, 0);
          }
        }
        HK_STAMP(1);
        HK_STAMP(2);
        HK_STAMP(6);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // my pieces have landed (and the bias values are here)
        staged = true;
      }
    }
    // ---- first launch of a net: the weight tile the same way, the fused first layer WITHOUT LDS until the tile has landed ---------
    // (a wave's DS operations queue behind its own global_load_lds transfers: a label exchange through LDS, as in the staged form
    //  below, would wait for the whole tile.)  Every lane fetches the labels of its own A fragment -- rows 16 i + r, labels
    // 4 lg + g -- and encodes them itself (the same fp64 expression), the layer's MFMAs and activations fill registers, and the
    // A tile is written once the transfers are in: what the staged form spends on 12 ds_write_b128 a thread after the layer
    // (1 750 of the workgroup's 12 500 cycles) is gone.
    if constexpr (FUSE_L0) {
      if (p.dma_tiles && kc == 0 && p.K <= HK_KC) {                 // (uniform)
        constexpr int NLG = (NL + 3) / 4, MAXT = (HK_KC / 16 + 3) / 4;
        constexpr int NCH = HK_PITCH / 4, NTR = 32 * NCH / 64;
        const int ntile = kn16 >> 4;
        double xl[2][NLG];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = (m0 + 16 * i + r < p.B) ? m0 + 16 * i + r : p.B - 1;
#pragma unroll
          for (int lg = 0; lg < NLG; ++lg) { const int d = 4 * lg + g; xl[i][lg] = p.theta[(size_t)row * p.ld_theta + (d < 4 ? d : 6)]; }
        }
        float w0t[MAXT][NLG], bzt[MAXT];
#pragma unroll
        for (int tt = 0; tt < MAXT; ++tt) {
          const int tcol = wave + 4 * tt;
          const int k = 16 * (tcol < ntile ? tcol : 0) + r;
          const int kq = k < p.K0 ? k : p.K0 - 1;
          bzt[tt] = p.b0[kq];
#pragma unroll
          for (int lg = 0; lg < NLG; ++lg) {
            const int d = 4 * lg + g;
            const float w = p.W0[(size_t)kq * p.n_labels + (d < p.n_labels ? d : 0)];
            w0t[tt][lg] = (d < p.n_labels) ? w : 0.f;
          }
        }
#pragma unroll
        for (int j0 = 0; j0 < NTR; j0 += 4) {
          const int j = j0 + wave;
          if (j < NTR) {
            const int sl = 64 * j + lane, rr = sl / NCH, c4 = sl - rr * NCH;
            const int nr = (n0 + rr < p.N) ? n0 + rr : p.N - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.Wd + (size_t)nr * p.ldwd + 4 * c4),
                                             (__attribute__((address_space(3))) void*)(Bs + 256 * j), 16, 0, 0);
          }
        }
        HK_STAMP(1);
        float xa[2][NLG];
#pragma unroll
        for (int lg = 0; lg < NLG; ++lg) {
          double xm = p.xmin[4 * lg < PAYNE_MAX_LABELS ? 4 * lg : 0], xdn = p.xden[4 * lg < PAYNE_MAX_LABELS ? 4 * lg : 0];
#pragma unroll
          for (int e = 1; e < 4; ++e) {
            if (4 * lg + e < PAYNE_MAX_LABELS) { xm = (g == e) ? p.xmin[4 * lg + e] : xm; xdn = (g == e) ? p.xden[4 * lg + e] : xdn; }
          }
          const bool dl = 4 * lg + g < p.n_labels;
#pragma unroll
          for (int i = 0; i < 2; ++i)
            xa[i][lg] = (dl && m0 + 16 * i + r < p.B) ? (float)((xl[i][lg] - xm) / xdn - 0.5) : 0.f;
        }
        HK_STAMP(2);
        float zr[MAXT][2][4];
        auto first_layer_regs = [&](auto actf) {
#pragma unroll
          for (int tt = 0; tt < MAXT; ++tt) {
            const int tcol = wave + 4 * tt;
            if (tcol < ntile) {             // (wave-uniform)
              const bool live = (16 * tcol + r) < p.K0;
#pragma unroll
              for (int i = 0; i < 2; ++i) {
                f32x4_t z = (f32x4_t){bzt[tt], bzt[tt], bzt[tt], bzt[tt]};
#pragma unroll
                for (int lg = 0; lg < NLG; ++lg) z = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i][lg], w0t[tt][lg], z, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) zr[tt][i][q] = live ? actf(z[q]) : 0.f;
              }
            }
          }
        };
        if (p.act0 == PAYNE_ACT_LRELU) first_layer_regs([](float z) { return lrelu01(z); });
        else if (p.act0 == PAYNE_ACT_SIGMOID) first_layer_regs([](float z) { return sigmoid_f32(z); });
        else first_layer_regs([](float z) { return z; });
        HK_STAMP(6);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // my pieces of the weight tile have landed
        float* const arow = As + (4 * g) * HK_PITCH + 16 * wave + r;
#pragma unroll
        for (int tt = 0; tt < MAXT; ++tt) {
          if (wave + 4 * tt < ntile) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int q = 0; q < 4; ++q) arow[(16 * i + q) * HK_PITCH + 64 * tt] = zr[tt][i][q];
          }
        }
        staged = true;
      }
    }
    if (!staged) {
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
    // The first layer runs on the matrix cores too: [32 rows x 4 labels] . [4 x 16 columns] is ONE v_mfma_f32_16x16x4_f32 per
    // 16 x 16 block of the A tile (bias in the accumulator), 2 x 19 of them at H = 300 shared by the four waves -- against
    // 32 x (4 fma + activation) per thread, behind 40 LDS reads, in the vector form this replaces (4 900 of the workgroup's
    // 15 000 cycles).  Wave w owns the 16-column blocks w, w + 4, ..; NLG label groups of four (the fifth label: a second one).
    constexpr int NLG = (NL + 3) / 4, MAXT = (HK_KC / 16 + 3) / 4;   // label groups; column blocks per wave (5)
    const int ntile = kn16 >> 4;
    float w0t[MAXT][NLG], bzt[MAXT];
    if (FUSE_L0) {
#pragma unroll
      for (int tt = 0; tt < MAXT; ++tt) {
        const int tcol = wave + 4 * tt;                            // column block; clamped loads, masked below
        const int k = kc + 16 * (tcol < ntile ? tcol : 0) + r;
        const int kq = k < p.K0 ? k : p.K0 - 1;
        bzt[tt] = p.b0[kq];
#pragma unroll
        for (int lg = 0; lg < NLG; ++lg) {
          const int d = 4 * lg + g;
          const float w = p.W0[(size_t)kq * p.n_labels + (d < p.n_labels ? d : 0)];
          w0t[tt][lg] = (d < p.n_labels) ? w : 0.f;
        }
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
      // A operand: lane (r, g) holds xhat[row 16 i + r][label 4 lg + g]; B operand: W0[column][label 4 lg + g]; C: column r,
      // rows 4 g + q of the block
      float xa[2][NLG];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int lg = 0; lg < NLG; ++lg) xa[i][lg] = Xh[(16 * i + r) * PAYNE_MAX_LABELS + 4 * lg + g];
      // One specialised copy of the loop per activation: chosen inside it, the choice is three scalar branches PER VALUE
      // (eighty values a lane) around an inlined sigmoid -- 3 800 of the workgroup's cycles as first written.
      float* const arow = As + (4 * g) * HK_PITCH + 16 * wave + r;   // + (16 i + q) rows, + 64 tt columns: immediates
      auto first_layer = [&](auto actf) {
#pragma unroll
        for (int tt = 0; tt < MAXT; ++tt) {
          const int tcol = wave + 4 * tt;
          if (tcol < ntile) {             // (wave-uniform)
            const bool live = (kc + 16 * tcol + r) < p.K0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              f32x4_t z = (f32x4_t){bzt[tt], bzt[tt], bzt[tt], bzt[tt]};
#pragma unroll
              for (int lg = 0; lg < NLG; ++lg) z = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i][lg], w0t[tt][lg], z, 0, 0, 0);
#pragma unroll
              for (int q = 0; q < 4; ++q) arow[(16 * i + q) * HK_PITCH + 64 * tt] = live ? actf(z[q]) : 0.f;
            }
          }
        }
      };
      if (p.act0 == PAYNE_ACT_LRELU) first_layer([](float z) { return lrelu01(z); });
      else if (p.act0 == PAYNE_ACT_SIGMOID) first_layer([](float z) { return sigmoid_f32(z); });
      else first_layer([](float z) { return z; });
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
    }   // (!staged)
    __syncthreads();
    HK_STAMP(3);
    // ---- the wave's quadrant over this chunk's K steps ------------------------------------------
    // (the fragments of ten steps requested before their first matrix instruction, twice: 80 registers -- all nineteen at once
    //  were 152 and cost the 5-label and the later-layer instantiations their second workgroup per CU)
    const int steps = kn16 >> 4;
    constexpr int MAXS = HK_KC / 16, HALF = (MAXS + 1) / 2;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      f32x4_t fa[HALF], fb[HALF];
#pragma unroll
      for (int u = 0; u < HALF; ++u) {
        const int sx = hb * HALF + u;
        const int k = (sx < steps ? sx : 0) * 16 + 4 * g;
        fa[u] = *reinterpret_cast<const f32x4_t*>(&As[(16 * qi + r) * HK_PITCH + k]);
        fb[u] = *reinterpret_cast<const f32x4_t*>(&Bs[(16 * qj + r) * HK_PITCH + k]);
      }
#pragma unroll
      for (int u = 0; u < HALF; ++u) {
        if (hb * HALF + u < steps && hb * HALF + u < MAXS) {        // (uniform)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][e], fb[u][e], acc[e & 1], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };
  if (p.K <= HK_KC) chunk(0);
  else for (int kc = 0; kc < p.K; kc += HK_KC) chunk(kc);
  HK_STAMP(4);
  // ---- epilogue straight from the accumulators (hk_store_quadrant) ---------------------------------------------------------
  {
    float y[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) y[q] = act_apply((acc[0][q] + acc[1][q]) + (bias1 - p.bias_shift), p.act);
    hk_store_quadrant(p, y, m0, n0, qi, qj, r, g);
  }
  HK_STAMP(5);
}

// NL: label slots the fused first layer loops over (4 for the usual Teff/logg/FeH/aFe nets, else PAYNE_MAX_LABELS)
// NW: waves a workgroup (4; 8: the first launch of a net whose tiles are hk_tile_h2's and with which no photometric tile rides along -- those
// count on two workgroups a CU: launch_hidden.  Everything but hk_tile_h2<NL, 8> is the first 256 threads' work).  A kernel of its own per NW:
// with both copies of the tile code in ONE kernel the stamped diagnostic twin faulted (NOTES R6.17); with one copy a kernel it runs clean.
template <bool FUSE_L0, int NL, int NW = 4>
__global__ void __launch_bounds__(64 * NW, 1) payne_dense_hidden_kernel(PAYNE_HK_LEAD_PARAMS, DenseParams p_, const PrepArgs pa) {
  static_assert(NW == 4 || (NW == 8 && FUSE_L0), "waves a workgroup");
  extern __shared__ __attribute__((aligned(16))) float hk_sm[];
  // (what a workgroup's first requests hang off arrives in registers at wave start: see payne_dense_dma3_kernel.  `pa` itself stays
  //  the kernel's argument: a modified COPY of it would have to live in scratch memory for the functions that take its tables by reference)
  DenseParams p = p_;
  const int pa_n_spec = (int)(lead_i0 & 0xffffu), pa_n_prep = (int)((lead_i0 >> 16) & 0x7fffu), pa_n_gemm = (int)(lead_i1 & 0xffffu);
  // (bit 31: the weight tile goes straight into LDS -- known HERE, not when the record's p.Wd has arrived: the first layer's requests wait for nothing)
  p.dma_tiles = (int)(lead_i0 >> 31);
  p.grid_n = (int)(lead_i1 >> 16);
  p.B = lead_B; p.N = lead_N; p.K = (int)(lead_i4 & 0xffffu); p.bias = lead_bias;
  if constexpr (FUSE_L0) {
    p.theta = static_cast<const double*>(lead_p0); p.W0 = lead_p1; p.b0 = lead_p2;
    p.ld_theta = (int)(lead_i2 & 0xffffu); p.n_labels = (int)((lead_i2 >> 16) & 0xffu); p.K0 = (int)(lead_i4 >> 16);
    p.h2_tiles = (int)(lead_i2 >> 31);                     // (the second layer on fp16 pairs: hk_tile_h2)
  } else {
    p.X = static_cast<const float*>(lead_p0); p.Wd = lead_p1;
    p.ldx = (int)(lead_i2 & 0xffffu); p.ldwd = (int)(lead_i2 >> 16);
    p.h2_tiles = (int)(lead_i4 >> 31);                     // (both operands as fp16 planes: hk_tile_h2x)
  }
  // Order of the launch's workgroups = order of dispatch: the walk's proposals made ahead (the longest-lived workgroups of the launch:
  // ~1 200 dependent fp64 instructions a wave) first, then the record writers, then the GEMM tiles, then the photometric tiles.
  const int front = pa_n_spec + pa_n_prep, bx = (int)blockIdx.x;
  if (bx < front || bx >= front + pa_n_gemm) {
    if constexpr (FUSE_L0) {
      if (NW > 4 && threadIdx.x >= 256) return;
      if (bx < pa_n_spec) {
        const int w = bx * 4 + (int)(threadIdx.x >> 6);
        if (pa.spec_walk) rwalk_spec_wave(pa.spec_walk->sd, pa.spec_w, w, (int)threadIdx.x & 63, pa.spec_step);
      } else if (bx < front) {
        const int cand = (bx - pa_n_spec) * 256 + (int)threadIdx.x;
        if (pa.out && cand < p.B) {
          // (the fall-back word from the row itself: read back from the record just written it was a dependent memory round trip at the
          //  end of the launch's longest-lived workgroups)
          const double vrot_row = p.theta[(size_t)cand * p.ld_theta + 5];      // (requested with the record's own reads, looked at after them)
          prep_candidate(pa.T, p.theta + (size_t)cand * p.ld_theta, pa.instr_factor, pa.out[cand]);
          if (pa.rot_flag && !(vrot_row != 0.0)) *pa.rot_flag = pa.rot_seq;   // ystpred.py:214 (every writer writes the same value)
        }
      } else if (pa.sed_mags) {
        const int j = bx - front - pa_n_gemm, f = j % pa.P.F, blk = j / pa.P.F;
#ifdef PAYNE_STAMPS
        unsigned long long* st = p.stamps ? p.stamps + (size_t)blockIdx.x * 16 : nullptr;
#else
        unsigned long long* st = nullptr;
#endif
        sed_tile(pa.P, p.theta, p.ld_theta, pa.sed_off, pa.sed_photscale, p.B, f, blk * pa.sed_cb, pa.sed_cb, pa.sed_mags,
                 reinterpret_cast<unsigned char*>(hk_sm), st);
      }
    }
    return;
  }
  // Nothing is outstanding on this path -- but the riders' branch above is structured into a region that flows through here, and the
  // compiler's wait-count pass carries ITS pending loads into the tile code: a drain (`s_waitcnt vmcnt(0)`) between the fifth and the
  // sixth of the tile's first requests.  A wait the pass can see (and that waits for nothing) clears its books.
  __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0)
#ifdef PAYNE_STAMPS
  if (p.stamps) p.stamps -= (size_t)front * 16;                    // (diagnostic build: row = GEMM tile)
#endif
  if constexpr (FUSE_L0) {
    if constexpr (NW > 4) { hk_tile_h2<NL, NW>(p, bx - front, hk_sm, (int)threadIdx.x); return; }         // (the eight-wave kernel has no other tile code)
    else if (p.h2_tiles) { hk_tile_h2<NL, 4>(p, bx - front, hk_sm, (int)threadIdx.x); return; }           // (uniform)
  } else {
    if (p.h2_tiles) { hk_tile_h2x(p, bx - front, hk_sm, (int)threadIdx.x); return; }
  }
  if constexpr (NW == 4) hk_tile<FUSE_L0, NL>(p, bx - front, hk_sm, (int)threadIdx.x);
}



// ----------------------------------------------------------------------------
// payne_dense_chain_kernel (PAYNE_V_HID_CHAIN; MEASURED SLOWER than a launch per layer, kept as a tested variant: LinNet's three layers
// 22.7 us against 3 x 5.7 -- a hop is an agent-scope release, ~6.5 us with a tile's planes freshly dirtied in the XCD's L2, + an acquire,
// ~1.7 us, as MI355X_MICROARCH.md prices them; 15 us a hop while every thread fenced).
// The hidden layers past the second of a deeper net (LinNet: three of them) in ONE launch.  A launch of
// such a layer is 5.7 us of launch and first-touch latency around 0.6 us of matrix instructions; its only cross-workgroup dependency
// is between the ten 32 x 32 tiles of one 32-candidate row block -- layer l + 1 of a row block needs layer l of the same block, of
// nothing else.  So: the grid of one layer (row blocks x ten column tiles, every workgroup resident), each workgroup running its
// tile of layer l (hk_tile_h2x: both operand tiles as fp16 planes straight into LDS), publishing it (planes stored, agent-scope
// release, one atomic increment of the row block's counter for this hop) and waiting for the block's ten increments before it
// reads the planes of layer l as the next layer's A operand (acquire).  Counters are never reset: every call adds `grid_n` to every
// (row block, hop) counter -- row blocks past the batch's end take part with their increments only -- and a call waits for
// `target0 + grid_n`, target0 = grid_n x (calls before this one).
// ----------------------------------------------------------------------------
constexpr int kChainMax = 6;
struct ChainParams {
  int n;                                                   // layers in the chain (hops: n - 1)
  const unsigned short* Wh[kChainMax]; const float* rs[kChainMax]; const float* bias[kChainMax]; int act[kChainMax];
  float out_scale[kChainMax];                              // the power of two layer i's output planes are written with
  unsigned short* buf[2]; size_t plane_x;                  // activation planes [2][b_max][304], two buffers taking turns
  int first_in;                                            // the buffer layer 0 of the chain reads
  unsigned short* Yp_last; size_t plane_y_last; int ldp_last; int half_last;   // where the last layer writes (the output layer's operand planes)
  unsigned long long* flags; unsigned long long target0;   // [grid_m_max][kChainMax] counters
  int B, N, grid_m, grid_m_max, grid_n;
#ifdef PAYNE_STAMPS
  unsigned long long* stamps;
#endif
};
__global__ void __launch_bounds__(256, 1) payne_dense_chain_kernel(const ChainParams cp);
#ifdef PAYNE_TU_DENSE
__global__ void __launch_bounds__(256, 1) payne_dense_chain_kernel(const ChainParams cp) {
  extern __shared__ __attribute__((aligned(16))) float hk_sm[];
  const int tid = (int)threadIdx.x, tile = (int)blockIdx.x;
  const int tm = tile / cp.grid_n;
  unsigned long long* const my = cp.flags + (size_t)tm * kChainMax;
  const bool live = tm < cp.grid_m;                        // (row blocks past the batch: counters only)
  for (int l = 0; l < cp.n; ++l) {
    if (l > 0 && live) {
      if (tid == 0) {
        const unsigned long long want = cp.target0 + (unsigned long long)cp.grid_n;
        while (__hip_atomic_load(my + (l - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (the compute unit's L1 is one: one wave's invalidate serves the workgroup)
      }
      __syncthreads();
    }
    if (live) {
      DenseParams p{};
      p.B = cp.B; p.N = cp.N; p.grid_m = cp.grid_m; p.grid_n = cp.grid_n;
      p.bias = cp.bias[l]; p.rs1 = cp.rs[l]; p.act = cp.act[l]; p.bias_shift = 0.f;
      p.Wh = cp.Wh[l]; p.plane_wh = (size_t)cp.N * HK2_K;
      p.Xp = cp.buf[(cp.first_in + l) & 1]; p.plane_x = cp.plane_x;
      if (l + 1 < cp.n) { p.Yp = cp.buf[(cp.first_in + l + 1) & 1]; p.plane_y = cp.plane_x; p.ldp = HK2_K; p.yp_half = 1; }
      else { p.Yp = cp.Yp_last; p.plane_y = cp.plane_y_last; p.ldp = cp.ldp_last; p.yp_half = cp.half_last; }
      p.yp_scale = cp.out_scale[l];
#ifdef PAYNE_STAMPS
      p.stamps = nullptr;
#endif
      hk_tile_h2x(p, tile, hk_sm, tid);
    }
    if (l + 1 < cp.n) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my stores of this layer's planes have left the compute unit ...
      __syncthreads();                                     // ... everybody's have (and everybody has read its fragments: the next layer's transfers may land)
      if (tid == 0) {                                      // ONE release for the workgroup (the write-back it implies is the cache's, not a thread's:
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); //  issued by all 256 threads it made a hop 15 us)
        __hip_atomic_fetch_add(my + l, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
#endif

// The instantiations that exist (compiled in k_dense.hip; `extern template` elsewhere).
#ifdef PAYNE_TU_DENSE
#define PAYNE_DENSE_T template
#else
#define PAYNE_DENSE_T extern template
#endif
PAYNE_DENSE_T __global__ void payne_dense_kernel<64, 64, 32, true>(DenseParams);
PAYNE_DENSE_T __global__ void payne_dense_kernel<64, 64, 32, false>(DenseParams);
PAYNE_DENSE_T __global__ void payne_dense_dma_kernel<4, 32, 0, 4, true>(DenseParams);
PAYNE_DENSE_T __global__ void payne_dense_dma_kernel<4, 32, 10, 4, true>(DenseParams);
PAYNE_DENSE_T __global__ void payne_dense_dma_kernel<4, 32, 0, 3, false>(DenseParams);
PAYNE_DENSE_T __global__ void payne_dense_dma_kernel<4, 64, 0, 3, true>(DenseParams);
PAYNE_DENSE_T __global__ void payne_dense_dma_kernel<4, 64, 5, 3, true>(DenseParams);
PAYNE_DENSE_T __global__ void payne_dense_dma3_kernel<0, 4, true>(PAYNE_D3_LEAD_TYPES, DenseParams);
PAYNE_DENSE_T __global__ void payne_dense_dma3_kernel<10, 4, true>(PAYNE_D3_LEAD_TYPES, DenseParams);
PAYNE_DENSE_T __global__ void payne_dense_dma3_kernel<0, 2, false>(PAYNE_D3_LEAD_TYPES, DenseParams);
PAYNE_DENSE_T __global__ void payne_dense_dma3f_kernel<10>(PAYNE_D3_LEAD_TYPES, DenseParams);
PAYNE_DENSE_T __global__ void payne_dense_dma2h_kernel<10, 32>(PAYNE_D3_LEAD_TYPES, DenseParams);
PAYNE_DENSE_T __global__ void payne_dense_dma2h_kernel<0, 32>(PAYNE_D3_LEAD_TYPES, DenseParams);
PAYNE_DENSE_T __global__ void payne_dense_dma2h_kernel<5, 64>(PAYNE_D3_LEAD_TYPES, DenseParams);
PAYNE_DENSE_T __global__ void payne_dense_big3_kernel<false>(DenseParams);
PAYNE_DENSE_T __global__ void payne_dense_big3_kernel<true>(DenseParams);
PAYNE_DENSE_T __global__ void payne_dense_dma2hh_kernel<5>(PAYNE_D3_LEAD_TYPES, DenseParams);
PAYNE_DENSE_T __global__ void payne_dense_hidden_kernel<true, 4>(PAYNE_HK_LEAD_TYPES, DenseParams, const PrepArgs);
PAYNE_DENSE_T __global__ void payne_dense_hidden_kernel<true, PAYNE_MAX_LABELS>(PAYNE_HK_LEAD_TYPES, DenseParams, const PrepArgs);
PAYNE_DENSE_T __global__ void payne_dense_hidden_kernel<false, 4>(PAYNE_HK_LEAD_TYPES, DenseParams, const PrepArgs);
PAYNE_DENSE_T __global__ void payne_dense_hidden_kernel<true, 4, 8>(PAYNE_HK_LEAD_TYPES, DenseParams, const PrepArgs);
PAYNE_DENSE_T __global__ void payne_dense_hidden_kernel<true, PAYNE_MAX_LABELS, 8>(PAYNE_HK_LEAD_TYPES, DenseParams, const PrepArgs);
