// dense_kernels.hpp -- the ANN's dense layers on the matrix cores (device code only; included once by
// payne_hip.hip).  Net.eval / ANN.eval of the reference: Payne/predict/ystpred.py:41-58,
// Payne/train/NNmodels.py:92-168.  The output layer ([B x H] . [H x Npix], 78 % of the path's FLOPs) is
// payne_dense_dma_kernel (LDS-DMA ring + MFMA f32), the hidden layers payne_dense_hidden_kernel.
#pragma once

// ============================================================================
// dense layer on the matrix cores
// ============================================================================
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));   // register-resident 16-byte value (HIP's f32x4_t struct arrays end up in scratch)

struct DenseParams {
  const float* X; int ldx;     // [B][ldx] activations (ignored with FUSE_L0)
  const float* W; int K;       // [N][K] row-major, K % 4 == 0
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
  const int k_tail = (p.k_real > 0 ? p.k_real : p.K) - (nk - 1) * BK;
  const int kk_last = __builtin_amdgcn_readfirstlane(k_tail >= BK ? BK / 8 : (k_tail <= 0 ? 1 : (k_tail + 7) / 8));
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
    // the zero-padded tail of the last step (K = 300 -> 320: 20 of its 32 columns) contributes exact zeros: skip
    // those matrix instructions (groups of 8 columns; a uniform branch)
    const int nkk = (it + 1 == nk) ? kk_last : BK / 8;
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
