// post_kernels.hpp -- kernels around the per-candidate spectrum pipeline whose phase code lives in
// post_core.hpp / post_seq.hpp (device code only; included once by payne_hip.hip): the LDS-resident
// payne_post_kernel, the global-workspace payne_post_big_kernel for spectra larger than LDS, and
// payne_lsf_kernel (LSF-vector broadening, Payne/utils/smoothing.py:482-586).
#pragma once
#include "select.hpp"
#include "sampler_core.hpp"
#include "post_onchip.hpp"
#include "post_onchip2.hpp"

// ============================================================================
// per-candidate spectrum pipeline
// ============================================================================
constexpr int kStampRow = 192;        // diagnostic build: stamps per candidate (the four-step transform has ~90 phases)
struct PostArgs {
  const double* theta; int ld_theta;
  double instr_factor;
  const float* raw; int ld_raw;
  float* out; int ld_out; int out_stage;
  double* lnl;                       // [B] or null
  const double* mags; int n_filters; // SED magnitudes of this batch (null: no photometry)
  const double* obs_mag; const double* obs_err;
  unsigned long long* stamps;        // diagnostic build: [B][kStampRow] cycle stamps (slot 0 = count)
  int stamp_sparse;                  // diagnostic build: only the first and the last stamp (slots 1 and kStampRow - 1)
  const CandState* prep;             // [B] per-candidate records made by the first dense launch (null: none)
  // the sampler's walk (payne_rwalk_step): the likelihood-only kernel runs chain b's NEXT step (settle this proposal, draw the
  // next) at its tail -- the workgroup of candidate b has just produced the one value that step waits for -- instead of a
  // launch of its own between two likelihood batches (null: no walk in progress)
  const WalkTail* tail; int tail_step, tail_propose;
  int tail_spec;                     // the next proposal was made ahead by this batch's hidden-layer launch (rwalk_spec_wave): the tail only settles and copies
  // rows launched in the frequency domain of a resampled model grid: *rot_flag == rot_seq says the batch held a candidate that does
  // not rotate and the output layer wrote PIXELS (row pitch ld_raw_alt) instead (dense_kernels.hpp PrepArgs; null: not such a launch)
  const unsigned long long* rot_flag; unsigned long long rot_seq; int ld_raw_alt;
};

// BUF_LDS: the two spectrum buffers are LDS (else a global workspace); TW_LDS: so is the twiddle table.
// NTHR: the workgroup's size when the kernel knows it (0: blockDim.x -- a read of the dispatch packet's implicit arguments: one more
// scalar load and its wait in front of the first phase's requests)
template <bool BUF_LDS, bool TW_LDS, int NTHR = 0>
struct DevExecT {
  static __device__ __forceinline__ int nthr_() { if constexpr (NTHR > 0) return NTHR; else return (int)blockDim.x; }
  static constexpr bool kTwLds = TW_LDS;                     // the twiddle table handed to run_candidate is the kernel's LDS copy
#ifdef __HIP_DEVICE_COMPILE__
  static __device__ __forceinline__ auto buf(c32* p) {
    if constexpr (BUF_LDS) return (PAYNE_AS_LDS f2v*)p; else return (PAYNE_AS_GLOBAL f2v*)p;
  }
  static __device__ __forceinline__ auto twid(const c32* p) {
    if constexpr (TW_LDS) return (const PAYNE_AS_LDS f2v*)p; else return (const PAYNE_AS_GLOBAL f2v*)p;
  }
  static __device__ __forceinline__ auto lds(c32* p) { return (PAYNE_AS_LDS f2v*)p; }
#else
  static c32* buf(c32* p) { return p; }
  static const c32* twid(const c32* p) { return p; }
  static c32* lds(c32* p) { return p; }
#endif
#ifdef PAYNE_STAMPS
  // diagnostic build only (libpayne_hip_diag.so): cycle stamp after every phase barrier
  unsigned long long* stamps = nullptr;
  int nst = 0;
  bool sparse = false;
#endif
  template <class F>
  __device__ __forceinline__ void par(F&& f) {
    f((int)threadIdx.x, nthr_());
    __syncthreads();
    mark(0);
  }
  template <class F>
  __device__ __forceinline__ void single(F&& f) { if (threadIdx.x == 0) f(nthr_()); }
  // diagnostic build: an extra cycle stamp inside a phase, written by thread `who`
  __device__ __forceinline__ void mark(int who) {
#ifdef PAYNE_STAMPS
    if (stamps && nst < kStampRow - 2) { ++nst; if (!sparse && (int)threadIdx.x == who) stamps[nst] = __builtin_amdgcn_s_memtime(); }
#else
    (void)who;
#endif
  }
  __device__ __forceinline__ int nthreads() const { return nthr_(); }
  // LDS tile of the four-step transform (2 x fft_tile_complex()); null: runtime-geometry passes
  c32* tile_ = nullptr;
  __device__ __forceinline__ c32* tile() const { return tile_; }
};

__device__ __forceinline__ double sed_chi2(const double* mags, const double* obs, const double* err, int F) {
  double s = 0.0;   // likelihood.py:109-112
  for (int f = 0; f < F; ++f) { const double d = mags[f] - obs[f]; s += (d * d) / (err[f] * err[f]); }
  return s;
}

// (a call, not inlined: the walk's registers must not count against the per-pixel phases' budget of 128 -- two
// workgroups per CU -- and pinning the kernel to that budget by attribute costs the FFT passes 2.5 us of scheduling freedom)
__device__ __attribute__((noinline)) static void walk_tail(const WalkTail* t, int b, int lane, double lnl, int step, int propose) {
  const WalkState W = uniform_copy(&t->w);
  rwalk_step_wave(t->sd, W, b, lane, lnl, step, 1, propose);
}

// (the walk stays behind the pointer here: by value in PostArgs it was 150 more bytes of kernel arguments whose scalar loads the
// compiler hoists to the kernel's start -- 93 spilled scalar registers against 52 --, and handed to a call by reference a stack copy)
__device__ __attribute__((noinline)) static void walk_tail_spec(const WalkTail* t, int b, int lane, double lnl) {
  const WalkState W = uniform_copy(&t->w);
  rwalk_settle_spec(W, b, lane, lnl);
}

// LEAN: the likelihood-only instantiation (out_stage == -1, no spectrum output, per-candidate records present):
// the output variants of the observed-grid loop, the stage branches and the in-kernel setup are compiled out
// (most of the 270 KB of the full kernel).
template <int LOG2N, bool TW_LDS, bool LEAN = false>
#if PAYNE_POST_THREADS > 512
#define PAYNE_POST_BOUNDS __launch_bounds__(kPostThreads) __attribute__((amdgpu_waves_per_eu(8, 8)))
#else
#define PAYNE_POST_BOUNDS __launch_bounds__(kPostThreads)
#endif
__global__ void PAYNE_POST_BOUNDS payne_post_kernel(const c32* lead_twf, const float* lead_raw, const CandState* lead_prep, const double* lead_theta, const unsigned long long* lead_rot_flag, const double* lead_mags, unsigned lead_ints, unsigned lead_rot_seq, const PostTables T_, PostArgs a_) {
  // The kernel's first loads hang off a handful of its arguments, and a wave waits 400-700 cycles for arguments it reads from the
  // kernarg segment (tools/exp/kernarg_preload.hip) -- those few are the LEADING scalar parameters, which the hardware hands over
  // in registers at wave start (-mllvm -amdgpu-kernarg-preload-count, build.py); the two records repeat them for everything else.
  // (lead_ints: ld_raw | ld_theta << 17 | n_filters << 23 | raw_freq << 31 -- post_lead_ints)
  PostTables T = T_;
  T.twf = lead_twf; T.raw_freq = (int)(lead_ints >> 31);
  PostArgs a = a_;
  a.raw = lead_raw; a.prep = lead_prep; a.theta = lead_theta; a.rot_flag = lead_rot_flag; a.mags = lead_mags;
  a.ld_raw = (int)(lead_ints & 0x1ffffu); a.ld_theta = (int)((lead_ints >> 17) & 0x3fu); a.n_filters = (int)((lead_ints >> 23) & 0xffu);
  // T by value: its pointer members then live in the kernarg segment and are known to be
  // global (a struct read through a device pointer yields generic pointers -> flat_load,
  // which also ties every table load to the LDS wait counter)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // (the fixed-geometry instantiations know their length: read from the kernel arguments it was a scalar load, a wait and two
  //  branches in front of everything else the kernel asks for)
  const int n1 = LOG2N > 0 ? (1 << LOG2N) : T.n1;
  int raw_freq = T.raw_freq, ld_raw = a.ld_raw;
  // (the batch's sequence number by its low 32 bits, a preloaded argument: the whole word sits in the record, another 400-700 cycles away)
  if (a.rot_flag != nullptr && (unsigned)*a.rot_flag == lead_rot_seq) { raw_freq = 0; ld_raw = a.ld_raw_alt; }
  float* bufA = reinterpret_cast<float*>(smem);
  float* bufB = bufA + fft_buf_floats(n1);                     // room for the padded FFT intermediates
  double* red = reinterpret_cast<double*>(bufB + fft_buf_floats(n1));          // scratch_doubles(256)
  CandState* S = reinterpret_cast<CandState*>(red + scratch_doubles(kPostThreads));
  // Start-up traffic of the kernel itself, REQUESTED here and committed inside run_candidate's first phase (`early`), after that
  // phase has requested the row, the record and theta: one memory round trip at the start of a workgroup's life, not two (as first
  // written the table's six loads were waited for and stored before the row was even asked for: ~0.8 us).
  const c32* twf = T.twf;
  typedef float f2g __attribute__((ext_vector_type(2)));
  constexpr int NTW = LOG2N > 0 ? plan_table_len((1 << (LOG2N > 0 ? LOG2N : 1)) / 2) : 1, PER = (NTW + kPostThreads - 1) / kPostThreads;
  f2g tw_tmp[PER];
  // rows handed over in the frequency domain: the first phase loads the split factors exp(-2 pi i j / 2M) slot by slot anyway
  // (slots_issue) and puts them into the LDS table itself -- here they would be 8 KB more for every workgroup to pull through an
  // L2 port that the kernel's first phase saturates (22 B/clk/CU: the row, the table and the records of two workgroups)
  const int ntw_copy = (LOG2N > 0 && TW_LDS && raw_freq) ? plan_total((1 << (LOG2N > 0 ? LOG2N : 1)) / 2) : NTW;
  if (TW_LDS) {   // the FFT's (pass-ordered) twiddles into LDS: the passes then never leave the CU
    c32* twl = reinterpret_cast<c32*>(reinterpret_cast<unsigned char*>(S) + ((sizeof(CandState) + 15) & ~(size_t)15));
    const f2g* __restrict__ g = reinterpret_cast<const f2g*>(T.twf);
    f2g* tl = reinterpret_cast<f2g*>(twl);
    if constexpr (LOG2N > 0) {
      // every load of the table in flight at once (a load -> store loop pays one L2 round trip per
      // iteration: twelve of them at 4096 points, ~3.5 us before the first phase could start)
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int i0 = (int)threadIdx.x + q * kPostThreads;
        tw_tmp[q] = g[i0 < ntw_copy ? i0 : ntw_copy - 1];
      }
    } else {
      const int nt = T.twf_n;
      for (int i = threadIdx.x; i < nt; i += kPostThreads) tl[i] = g[i];
    }
    twf = twl;                                                 // made visible by the first phase barrier
  }
  const int b = blockIdx.x;
  // joint likelihoods: the photometric terms ((m - o) / e)^2 by one lane per filter of the second wave, requested NOW -- asked for by
  // thread 0 after the spectrum's chi^2 they were a memory round trip and seven dependent fp64 divisions at the end of the
  // workgroup's life (C3: 1.3 us of the post kernel)
  __shared__ double sed_terms[64];
  const bool sed_early = a.mags != nullptr && a.n_filters <= 64;
  const bool sed_lane = sed_early && (int)threadIdx.x >= 64 && (int)threadIdx.x < 64 + a.n_filters;
  double sed_m = 0.0, sed_o = 0.0, sed_e = 1.0;
  if (sed_lane) {
    const int f = (int)threadIdx.x - 64;
    sed_m = a.mags[(size_t)b * a.n_filters + f]; sed_o = a.obs_mag[f]; sed_e = a.obs_err[f];
  }
  auto early = [&]() {
    // (what it needs beyond the loaded values is worked out again here: every scalar it captured was a scalar register the whole
    //  first phase could not use, and the kernel sits one spilled scalar away from 129 vector registers = one workgroup per CU)
    if constexpr (TW_LDS && LOG2N > 0) {
      f2g* tl = reinterpret_cast<f2g*>(reinterpret_cast<unsigned char*>(S) + ((sizeof(CandState) + 15) & ~(size_t)15));
      // (every value named before the first store: the compiler otherwise sinks the load of a slot whose store is conditional
      //  -- the last one -- into that condition, behind the wait for the others: a second round trip)
#pragma unroll
      for (int q = 0; q < PER; ++q) asm volatile("" : "+v"(tw_tmp[q]));
      const int ncopy = raw_freq ? plan_total((1 << (LOG2N > 0 ? LOG2N : 1)) / 2) : NTW;
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int i0 = (int)threadIdx.x + q * kPostThreads;
        if (i0 < ncopy) tl[i0] = tw_tmp[q];
      }
    }
    if (a.mags != nullptr && a.n_filters <= 64 && (int)threadIdx.x >= 64 && (int)threadIdx.x < 64 + a.n_filters) {
      const double d = sed_m - sed_o;
      sed_terms[(int)threadIdx.x - 64] = (d * d) / (sed_e * sed_e);   // likelihood.py:109-112 (read by thread 0 behind the phases' barriers)
    }
  };
  DevExecT<true, TW_LDS, kPostThreads> ex;
#ifdef PAYNE_STAMPS
  if (a.stamps) {
    ex.stamps = a.stamps + (size_t)b * kStampRow;
    if (threadIdx.x == 0) { ex.stamps[1] = __builtin_amdgcn_s_memtime(); }
    ex.nst = 1;
    ex.sparse = a.stamp_sparse != 0;
  }
#endif
  double* chi2 = red + scratch_doubles(kPostThreads) - 1;
  const int ostage = LEAN ? -1 : a.out_stage;
  float* outp = LEAN ? nullptr : (a.out ? a.out + (size_t)b * a.ld_out : nullptr);
  const CandState* prep = a.prep ? a.prep + b : nullptr;
  if (LEAN) { prep = a.prep + b; __builtin_assume(prep != nullptr); }     // LEAN is launched only with records
  run_candidate<LOG2N, kPostThreads>(ex, T, twf, a.theta + (size_t)b * a.ld_theta, a.instr_factor,
                                     a.raw + (size_t)b * ld_raw, bufA, bufB, *S, red, outp, ostage, chi2, prep, early, raw_freq);
  double lnl_v = 0.0;
  if (threadIdx.x == 0 && a.lnl && ostage < 0) {
    double x2 = *chi2;
    if (sed_early) { double sx = 0.0; for (int f = 0; f < a.n_filters; ++f) sx += sed_terms[f]; x2 += sx; }   // (sed_chi2's order)
    else if (a.mags) x2 += sed_chi2(a.mags + (size_t)b * a.n_filters, a.obs_mag, a.obs_err, a.n_filters);
    lnl_v = -0.5 * x2;                                          // likelihood.py:117
    a.lnl[b] = lnl_v;
  }
  if constexpr (LEAN) {
    if (a.tail && threadIdx.x < 64) {                           // (wave 0: thread 0 holds the value, the others get it by shuffle)
      const double l = __shfl(lnl_v, 0);
      if (a.tail_spec) walk_tail_spec(a.tail, b, (int)threadIdx.x, l);
      else walk_tail(a.tail, b, (int)threadIdx.x, l, a.tail_step, a.tail_propose);
    }
  }
#ifdef PAYNE_STAMPS
  if (a.stamps && threadIdx.x == 0) { ex.stamps[0] = (unsigned long long)ex.nst; ex.stamps[kStampRow - 1] = __builtin_amdgcn_s_memtime(); }
#endif
}


// Spectra that do not fit LDS (n1 > 16384, e.g. the 65k-pixel R~100k grid): the same phase code
// with the two spectrum buffers in a per-workgroup global workspace (L2 / Infinity-Cache resident
// while it is being worked on) and the runtime-geometry FFT.  Workgroups are persistent and walk
// the batch with stride gridDim.x, so the workspace is sized by the grid, not by the batch.  All
// waves of a workgroup share one CU's L1, so __syncthreads() orders the global accesses between
// phases exactly as it orders LDS.  This is the HBM/L2-bandwidth-bound regime of SURVEY.md 8(d).
// With `tile_lds` the launch carries 2 x fft_tile_complex() complex values of dynamic LDS and the transforms
// take the four-step form (fft_run_tiled): two round trips through the workspace per transform instead of five.
#ifndef PAYNE_BIG_THREADS
#define PAYNE_BIG_THREADS 512
#endif
constexpr int kBigThreads = PAYNE_BIG_THREADS;
#ifndef PAYNE_TU_BIG
__global__ void __launch_bounds__(kBigThreads) payne_post_big_kernel(const PostTables T, PostArgs a, float* ws, int B, int tile_lds);
#else
__global__ void __launch_bounds__(kBigThreads) payne_post_big_kernel(const PostTables T, PostArgs a, float* ws, int B, int tile_lds) {
  __shared__ double red[kBigThreads + kBigThreads / 2 + 2];
  __shared__ CandState S;
  extern __shared__ __attribute__((aligned(16))) unsigned char big_sm[];      // the four-step transform's tile (or nothing)
  float* bufA = ws + (size_t)blockIdx.x * 2 * T.n1;
  float* bufB = bufA + T.n1;
  DevExecT<false, false> ex;
  if (tile_lds & 1) ex.tile_ = reinterpret_cast<c32*>(big_sm);
  double* chi2 = red + scratch_doubles(kBigThreads) - 1;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
#ifdef PAYNE_STAMPS
    if (a.stamps) {                                        // diagnostic build: cycle stamps of every candidate's phases
      ex.stamps = a.stamps + (size_t)b * kStampRow;
      if (threadIdx.x == 0) ex.stamps[1] = __builtin_amdgcn_s_memtime();
      ex.nst = 1;
    }
#endif
    run_candidate<0, kBigThreads>(ex, T, T.tw, a.theta + (size_t)b * a.ld_theta, a.instr_factor,
                                  a.raw + (size_t)b * a.ld_raw, bufA, bufB, S, red,
                                  a.out ? a.out + (size_t)b * a.ld_out : nullptr, a.out_stage, chi2,
                                  a.prep ? a.prep + b : nullptr);     // records made ahead of the kernel: no mask scan here
    if (threadIdx.x == 0 && a.lnl && a.out_stage < 0) {
      double x2 = *chi2;
      if (a.mags) x2 += sed_chi2(a.mags + (size_t)b * a.n_filters, a.obs_mag, a.obs_err, a.n_filters);
      a.lnl[b] = -0.5 * x2;
    }
#ifdef PAYNE_STAMPS
    if (a.stamps && threadIdx.x == 0) ex.stamps[0] = (unsigned long long)ex.nst;
#endif
    __syncthreads();
  }
}

#endif

// The same walk of the batch with each convolution stage of a 65 536-point spectrum kept on the compute unit (post_onchip.hpp):
// 512 threads, the stage's 32 768 complex points in registers, 128 KB of LDS as the transpose buffer.  The phases around the
// two stages (mask window, resampling, observed grid, chi^2) are the global-workspace phases of payne_post_big_kernel; a
// candidate whose instrumental window needs a shorter transform takes that kernel's runtime-geometry passes for that stage.
namespace payne { template <bool TW, int N> struct ex_lds_tail<DevExecT<true, TW, N>> { static constexpr bool value = true; }; }
struct ChipExec : DevExecT<false, false> {
#ifdef __HIP_DEVICE_COMPILE__
  ChipLds L;
#endif
};
namespace payne { template <> struct ex_chip<ChipExec> { static constexpr bool value = true; }; }
#ifdef __HIP_DEVICE_COMPILE__
namespace payne {
template <bool VSINI, class Ex>
PAYNE_SEQ float* chip_conv_stage(Ex& ex, float* work, const TaperArgs& ta, bool& edge, const float* src0, const Window* rs, bool zin, bool scrub) {
  if constexpr (ex_chip<Ex>::value) {
    if (rs) {                                                // the instrumental stage: its load resamples `src0` itself
      __shared__ ChipResample R;                             // (handed to the out-of-line stage through LDS)
      if (threadIdx.x == 0) R = chip_resample_of(*rs);
      __syncthreads();
      chip_conv<VSINI>(ex.L, src0, work, ta, false, edge, (int)threadIdx.x, &R, false);
    } else {
      chip_conv<VSINI>(ex.L, src0 ? src0 : work, work, ta, src0 != nullptr && scrub, edge, (int)threadIdx.x, nullptr, zin && src0 != nullptr);
    }
    ex.mark(0);
    edge = false;
  }
  return work;
}
}  // namespace payne
#endif
#ifndef PAYNE_TU_BIG
__global__ void __launch_bounds__(kChipThreads) payne_post_chip_kernel(const PostTables T, PostArgs a, float* ws, int B);
#else
// (anything but the usual path of a candidate -- no records, a spectrum output before the instrumental stage, a shorter window, a
//  pixel row that does not rotate -- through the general sequence, OUT OF LINE: inlined into the kernel its dozens of branches
//  were what the register allocator spilled around the stages' calls, 67 vector registers and 350 bytes of scratch a thread)
__device__ __attribute__((noinline)) static void chip_general_candidate(ChipExec& ex, const PostTables& T, const PostArgs& a, int b, float* bufA,
                                                                        float* bufB, CandState& S, double* red, double* chi2) {
  run_candidate<0, kChipThreads>(ex, T, T.tw, a.theta + (size_t)b * a.ld_theta, a.instr_factor,
                                 a.raw + (size_t)b * a.ld_raw, bufA, bufB, S, red,
                                 a.out ? a.out + (size_t)b * a.ld_out : nullptr, a.out_stage, chi2,
                                 a.prep ? a.prep + b : nullptr);
}
__global__ void __launch_bounds__(kChipThreads) payne_post_chip_kernel(const PostTables T, PostArgs a, float* ws, int B) {
#ifdef __HIP_DEVICE_COMPILE__
  __shared__ double red[kChipThreads + kChipThreads / 2 + 2];
  __shared__ CandState S;
  __shared__ ChipResample Rs;
  __shared__ int obs_on_chip;
  __shared__ float obs_edge;
  extern __shared__ __attribute__((aligned(16))) unsigned char chip_sm[];
  chip_fill_tables(chip_lds(chip_sm), T.tw, (int)threadIdx.x);
  __syncthreads();
  double* chi2 = red + scratch_doubles(kChipThreads) - 1;
  const int stage = a.out_stage;
  // what every candidate of the usual path shares (uniform)
  const bool usual_launch = a.prep != nullptr && T.geo && T.rot_identity && T.nobs > 0 && T.n1 == kChipN1 &&
                            (stage == -1 || stage == 2 || stage == 3) && (a.out != nullptr || T.obs_f1 != nullptr) &&
                            row_vectorised(T.npix, a.raw) && (a.ld_raw & 3) == 0;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    // (everything a candidate needs is derived from the kernel's arguments HERE, inside the loop: nothing but the loop's own scalars
    //  is alive across the stages' calls, which use the whole register file)
    const int tid = (int)threadIdx.x;
    float* bufA = ws + (size_t)blockIdx.x * 2 * T.n1;
    float* bufB = bufA + T.n1;
    ChipExec ex;
    ex.L = chip_lds(chip_sm);
#ifdef PAYNE_STAMPS
    if (a.stamps) {                                        // diagnostic build: cycle stamps of every candidate's phases
      ex.stamps = a.stamps + (size_t)b * kStampRow;
      if (threadIdx.x == 0) ex.stamps[1] = __builtin_amdgcn_s_memtime();
      ex.nst = 1;
    }
#endif
    bool usual = usual_launch;
    if (usual) {                                           // the record into LDS (a dword copy by the first waves)
      const unsigned* src = reinterpret_cast<const unsigned*>(a.prep + b);
      unsigned* dst = reinterpret_cast<unsigned*>(&S);
      for (int i = tid; i < kPrepDwords; i += kChipThreads) dst[i] = src[i];
      __syncthreads();
      // rotation + instrumental smoothing with a 65 536-point window: both stages on the compute unit.  (A pixel row that does
      // not rotate skips the first stage: general sequence.)
      usual = S.do_smooth && S.w_ready && !S.W.bad && S.W.n2 == kChipN1 && (S.do_rot || T.raw_freq);
      // ... and every bin of the rotation stage inside the taper table (post_onchip.hpp chip_taper_pairs: the stage without its cold call)
      if (usual && S.do_rot) {
        const double c64 = (S.vs_a * T.vs_val) * (1.0 / kVsTabStep);
        usual = (double)(kChipN1 / 2) * c64 < (double)(T.vs_tab_n - 3);          // (false for NaN)
      }
    }
    if (!usual) {
      __syncthreads();
      // (copies made HERE: what is passed by reference lives in memory, and the kernel's own copies of its arguments must not)
      PostTables Tl = T;
      PostArgs al = a;
      ChipExec exl = ex;
      chip_general_candidate(exl, Tl, al, b, bufA, bufB, S, red, chi2);
#ifdef PAYNE_STAMPS
      ex.nst = exl.nst;
#endif
    } else {
      // ---- rotation stage (ystpred.py:211-224): the row itself in -- pixels (NaN -> 0) or, T.raw_freq, its transform in the order
      // the taper wants it (a candidate that does not rotate: the taper of u = 0) --, the edge rule out; into bufB
      {
        TaperArgs ta{};
        ta.vs_tab = T.vs_tab; ta.vs_tab_n = T.vs_tab_n;
        ta.vs_c = S.do_rot ? S.vs_a * T.vs_val : 0.0;
        ta.vs_c64 = ta.vs_c * (1.0 / kVsTabStep);
        const bool rot = S.do_rot != 0;
        chip_conv<true, false>(ex.L, a.raw + (size_t)b * a.ld_raw, bufB, ta, rot, rot && stage != 6 && stage != 7, tid, nullptr, T.raw_freq != 0);
        ex.mark(0);
      }
      // ---- instrumental stage: the candidate's window (mask, Doppler shift, pow-2 log grid) gathered while loading, the result
      // over its input
      {
        if (tid == 0) {
          Rs = chip_resample_of(S.W);
          // chi^2 alone, no blaze, whole blocks of records, ascending wavelengths ALL strictly inside the window (its two ends, with
          // a margin far above what the two logarithms here and the records' own can differ by): the observed grid on the compute unit
          int ok = (a.out == nullptr && stage == -1 && T.obs_f1 != nullptr && T.npoly == 0 && T.obs_sorted && obs_fast_ok(T, 16 * kChipThreads)) ? 1 : 0;
          if (ok) {
            const double t0 = fma(log(T.obs_min), S.W.obA, S.W.obB), t1 = fma(log(T.obs_max), S.W.obA, S.W.obB);
            ok = (t0 >= 1e-3 && t1 <= (double)(S.W.n2 - 1) - 1e-3) ? 1 : 0;      // (false for NaN)
          }
          obs_on_chip = ok;
        }
        __syncthreads();
        TaperArgs tg{};
        tg.g_c2 = S.W.g_c2;
        if (obs_on_chip) {
          const int npad = (T.nobs + 16 * kChipThreads - 1) / (16 * kChipThreads) * (16 * kChipThreads);
          const float acc = chip_conv_obs(ex.L, bufB, tg, tid, &Rs, T.obs_rec, npad, obs_fast_consts(S.W), (PAYNE_AS_LDS float*)&obs_edge);
          ex.mark(0);
          store_partial(tid, (double)acc, red);
          __syncthreads();
          if (tid == 0) { double s = 0.0; for (int i = 0; i < n_slots(kChipThreads); ++i) s += red[i]; *chi2 = s; }
        } else {
          chip_conv<false>(ex.L, bufB, bufB, tg, false, false, tid, &Rs, false);
          ex.mark(0);
          // ---- observed grid, blaze, chi^2
          float* outp = a.out ? a.out + (size_t)b * a.ld_out : nullptr;
          store_partial(tid, phase_obs<16>(tid, kChipThreads, T, S, S.W, bufB, outp, stage), red);
          __syncthreads();
          if (tid == 0) { double s = 0.0; for (int i = 0; i < n_slots(kChipThreads); ++i) s += red[i]; *chi2 = s; }
        }
      }
    }
    if (threadIdx.x == 0 && a.lnl && a.out_stage < 0) {
      double x2 = *chi2;
      if (a.mags) x2 += sed_chi2(a.mags + (size_t)b * a.n_filters, a.obs_mag, a.obs_err, a.n_filters);
      a.lnl[b] = -0.5 * x2;
    }
#ifdef PAYNE_STAMPS
    if (a.stamps && threadIdx.x == 0) ex.stamps[0] = (unsigned long long)ex.nst;
#endif
    __syncthreads();
  }
#endif
}
#endif

// 32 768-point spectra (16 384 < n1 <= 32 768: e.g. the reference's own 25 600-pixel demo spectrum, demo/runPayne.py:43-50, any
// R ~ 50-80k fit): the same on-chip stages, TWO candidates at a time (post_onchip2.hpp).  A workgroup walks the batch in pairs; a
// pair whose candidates both take the usual path -- rotation, instrumental smoothing, a 32 768-point instrumental window, records
// made ahead -- runs both convolution stages of both candidates on the compute unit (five transfers of each spectrum, as at
// 65 536 points); anything else (a candidate without rotation, a shorter window, spectrum output stages 0 / 1) goes candidate by
// candidate through payne_post_big_kernel's phases with the runtime-geometry transform.
constexpr size_t kChip2WsFloatsPerGroup(int n1) { return (size_t)4 * n1; }      // two buffers per candidate
#ifndef PAYNE_TU_CHIP2
__global__ void __launch_bounds__(kChipThreads) payne_post_chip2_kernel(const PostTables T, PostArgs a, float* ws, int B);
#else
__device__ __attribute__((noinline)) static void chip2_general_candidate(DevExecT<false, false>& ex, const PostTables& T, const PostArgs& a, int b,
                                                                         float* bufA, float* bufB, CandState& S, double* red, double* chi2) {
  run_candidate<0, kChipThreads>(ex, T, T.tw, a.theta + (size_t)b * a.ld_theta, a.instr_factor, a.raw + (size_t)b * a.ld_raw,
                                 bufA, bufB, S, red, a.out ? a.out + (size_t)b * a.ld_out : nullptr, a.out_stage, chi2,
                                 a.prep ? a.prep + b : nullptr);
}
__global__ void __launch_bounds__(kChipThreads) payne_post_chip2_kernel(const PostTables T, PostArgs a, float* ws, int B) {
#ifdef __HIP_DEVICE_COMPILE__
  __shared__ double red[kChipThreads + kChipThreads / 2 + 2];
  __shared__ CandState S2[2];
  __shared__ ChipResample R2[2];
  extern __shared__ __attribute__((aligned(16))) unsigned char chip_sm[];
  float* base = ws + (size_t)blockIdx.x * 4 * T.n1;
  float* buf[2][2] = {{base, base + T.n1}, {base + 2 * (size_t)T.n1, base + 3 * (size_t)T.n1}};
  DevExecT<false, false> ex;                                   // (the phases around the stages: global-workspace executor)
  const ChipLds L = chip_lds(chip_sm);
  chip2_fill_tables(L, T.tw, (int)threadIdx.x);
  __syncthreads();
  double* chi2 = red + scratch_doubles(kChipThreads) - 1;
  const int tid = (int)threadIdx.x;
  const int stage = a.out_stage;
  for (int p = blockIdx.x; 2 * p < B; p += gridDim.x) {
    const int bb[2] = {2 * p, (2 * p + 1 < B) ? 2 * p + 1 : 2 * p};      // (an odd batch: the last candidate twice, written once)
    const int ncand = bb[1] != bb[0] ? 2 : 1;
    bool pair = a.prep != nullptr && T.geo && T.nobs > 0 && (stage == -1 || stage == 2 || stage == 3) && (a.out != nullptr || T.obs_f1 != nullptr);
    if (pair) {
      // both records into LDS (a dword copy by the first two waves)
      const unsigned* s0 = reinterpret_cast<const unsigned*>(a.prep + bb[0]);
      const unsigned* s1 = reinterpret_cast<const unsigned*>(a.prep + bb[1]);
      unsigned* d0 = reinterpret_cast<unsigned*>(&S2[0]);
      unsigned* d1 = reinterpret_cast<unsigned*>(&S2[1]);
      for (int i = tid; i < kPrepDwords; i += kChipThreads) { d0[i] = s0[i]; d1[i] = s1[i]; }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 2; ++c)
        pair = pair && S2[c].do_rot && S2[c].do_smooth && S2[c].w_ready && !S2[c].W.bad && S2[c].W.n2 == kChip2N1;
    }
    if (!pair) {
      __syncthreads();
      const float* rowp[2] = {a.raw + (size_t)bb[0] * a.ld_raw, a.raw + (size_t)bb[1] * a.ld_raw};
      if (T.raw_freq) {
        // rows handed over transformed: back to pixels first (the taper of u = 0 is 1 in every bin; no scrub: a NaN row stays NaN),
        // into the second candidate's buffers, which the candidate-by-candidate phases below do not use
        TaperArgs id{};
        id.vs_tab = T.vs_tab; id.vs_tab_n = T.vs_tab_n;
        Chip2Io io;
        io.in[0] = rowp[0]; io.in[1] = rowp[1]; io.out[0] = buf[1][0]; io.out[1] = buf[1][1];
        chip2_conv<true>(L, io, id, id, false, false, tid, nullptr, true);
        rowp[0] = buf[1][0]; rowp[1] = buf[1][1];
      }
      for (int c = 0; c < ncand; ++c) {
        const int b = bb[c];
        {                                                    // (out of line, on copies made here: see payne_post_chip_kernel)
          PostTables Tl = T;
          PostArgs al = a;
          al.raw = rowp[c] - (size_t)b * a.ld_raw;           // (so that the callee's row b is this one)
          DevExecT<false, false> exl = ex;
          chip2_general_candidate(exl, Tl, al, b, buf[0][0], buf[0][1], S2[0], red, chi2);
        }
        if (tid == 0 && a.lnl && stage < 0) {
          double x2 = *chi2;
          if (a.mags) x2 += sed_chi2(a.mags + (size_t)b * a.n_filters, a.obs_mag, a.obs_err, a.n_filters);
          a.lnl[b] = -0.5 * x2;
        }
        __syncthreads();
      }
      continue;
    }
    const float* raw[2] = {a.raw + (size_t)bb[0] * a.ld_raw, a.raw + (size_t)bb[1] * a.ld_raw};
    // ---- rotation stage of both candidates (ystpred.py:211-224, smoothing.py:293-336): the raw rows in (NaN -> 0), the edge rule out
    TaperArgs ta[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      ta[c] = TaperArgs{};
      ta[c].vs_tab = T.vs_tab; ta[c].vs_tab_n = T.vs_tab_n;
      ta[c].vs_c = S2[c].vs_a * T.vs_val;
      ta[c].vs_c64 = ta[c].vs_c * (1.0 / kVsTabStep);
    }
    const bool ident = T.rot_identity != 0;
    Chip2Io io;
    if (ident) { io.in[0] = raw[0]; io.in[1] = raw[1]; }
    else {                                                   // resample_wave onto the pow-2 log grid first (static maps)
      for (int c = 0; c < 2; ++c) ex.par([&](int t, int n) { phase_rot_resample(t, n, T, raw[c], buf[c][1]); });
      io.in[0] = buf[0][1]; io.in[1] = buf[1][1];
    }
    io.out[0] = buf[0][1]; io.out[1] = buf[1][1];
    chip2_conv<true>(L, io, ta[0], ta[1], ident, ident, tid, nullptr, ident && T.raw_freq != 0);
    float* spec[2] = {buf[0][1], buf[1][1]};
    if (!ident) {                                            // back onto the model grid (NaN outside), then the edge rule
      for (int c = 0; c < 2; ++c) ex.par([&](int t, int n) { phase_rot_back(t, n, T, buf[c][1], buf[c][0]); });
      spec[0] = buf[0][0]; spec[1] = buf[1][0];
      ex.par([&](int t, int) { phase_rot_edges(t, T.npix, spec[0]); phase_rot_edges(t, T.npix, spec[1]); });
    }
    // ---- instrumental stage: each candidate's own window (mask, Doppler shift, pow-2 log grid), gathered while loading
    if (tid < 2) R2[tid] = chip_resample_of(S2[tid].W);
    __syncthreads();
    ta[0].g_c2 = S2[0].W.g_c2; ta[1].g_c2 = S2[1].W.g_c2;
    float* conv[2] = {spec[0] == buf[0][1] ? buf[0][0] : buf[0][1], spec[1] == buf[1][1] ? buf[1][0] : buf[1][1]};
    io.in[0] = spec[0]; io.in[1] = spec[1]; io.out[0] = conv[0]; io.out[1] = conv[1];
    chip2_conv<false>(L, io, ta[0], ta[1], false, false, tid, R2, false);
    // ---- observed grid, blaze, chi^2: candidate by candidate
    for (int c = 0; c < ncand; ++c) {
      const int b = bb[c];
      float* outp = a.out ? a.out + (size_t)b * a.ld_out : nullptr;
      const Window W = S2[c].W;
      ex.par([&](int t, int n) { store_partial(t, phase_obs<16>(t, n, T, S2[c], W, conv[c], outp, stage), red); });
      if (tid == 0 && a.lnl && stage < 0) {
        double x2 = 0.0;
        const int ns = n_slots(kChipThreads);
        for (int i = 0; i < ns; ++i) x2 += red[i];
        if (a.mags) x2 += sed_chi2(a.mags + (size_t)b * a.n_filters, a.obs_mag, a.obs_err, a.n_filters);
        a.lnl[b] = -0.5 * x2;
      }
      __syncthreads();
    }
  }
#endif
}
#endif

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
  const double* obs_wave;                // [nobs]
  const double* lsf_wave; const double* lsf; int n_lsf;   // the dispersion vector and its abscissa ([n_lsf]; getspec: the observed grid itself)
  double* ws; size_t ws_stride;          // per candidate: a[npix] | cdf[npix] | lam[n1]
  float* fws; size_t fws_stride;         // GLOBAL form: per candidate the two FFT buffers (2 x fft_buf_floats(n1) floats)
  float* out; int ld_out; int out_stage; // 2 / 3 / -1
  double* lnl;
  const double* mags; int n_filters; const double* obs_mag; const double* obs_err;
};
#ifdef PAYNE_TU_BIG
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
// GLOBAL: the median by radix selection (select.hpp) and the two FFT buffers in a global workspace -- no limit on the
// spectrum's length; otherwise everything in LDS (spectra up to 8192 pixels).
template <bool GLOBAL>
__global__ void __launch_bounds__(256) payne_lsf_kernel(const PostTables T, LsfArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lsm[];
  double* sortb = reinterpret_cast<double*>(lsm);                      // [n1] median sort buffer (LDS form)
  float* bufA = GLOBAL ? a.fws + (size_t)blockIdx.x * a.fws_stride : reinterpret_cast<float*>(sortb + T.n1);
  float* bufB = bufA + fft_buf_floats(T.n1);
  double* red = GLOBAL ? reinterpret_cast<double*>(lsm) : reinterpret_cast<double*>(bufB + fft_buf_floats(T.n1));    // [256 + 8]
  __shared__ int hist_s[256];
  __shared__ unsigned long long bc_s[2];
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
      const double sig = interp_np(W(i), a.lsf_wave, a.lsf, a.n_lsf);
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
    // ---- x_per_sigma = nanmedian(gradient(cdf) / r)
    auto ratio = [&](int i) {
      const double g = (i == 0) ? (wsC[1] - wsC[0]) : ((i == n - 1) ? (wsC[n - 1] - wsC[n - 2]) : (wsC[i + 1] - wsC[i - 1]) / 2.0);
      return g / wsA[i];
    };
    double xps_sel = 0.0;
    if (GLOBAL) {
      xps_sel = nanmedian_select<256>(ratio, n, hist_s, bc_s, &nvalid_s);
    } else {
    // bitonic sort of the ratios in LDS
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
    }
    if (tid == 0) {
      const int m = nvalid_s;
      const double xps = GLOBAL ? xps_sel : (m == 0 ? __builtin_nan("") : ((m & 1) ? sortb[m >> 1] : 0.5 * (sortb[(m >> 1) - 1] + sortb[m >> 1])));
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
    DevExecT<!GLOBAL, false> ex;
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

template __global__ void payne_lsf_kernel<false>(const PostTables, LsfArgs);
template __global__ void payne_lsf_kernel<true>(const PostTables, LsfArgs);
#else
template <bool GLOBAL> __global__ void payne_lsf_kernel(const PostTables T, LsfArgs a);
extern template __global__ void payne_lsf_kernel<false>(const PostTables, LsfArgs);
extern template __global__ void payne_lsf_kernel<true>(const PostTables, LsfArgs);
#endif

// The instantiations that exist (each is compiled in one of the k_post_*.hip units; everybody else sees them as
// `extern template`): X(LOG2N, TW_LDS, LEAN)
#define PAYNE_POST_LEAN_LIST(X) X(12, true, true) X(11, true, true) X(10, true, true) X(13, false, true)
#define PAYNE_POST_FULL_A_LIST(X) X(12, true, false) X(0, true, false)
#define PAYNE_POST_FULL_B_LIST(X) X(10, true, false) X(11, true, false) X(13, false, false) X(0, false, false)
#define PAYNE_POST_SIG const c32*, const float*, const CandState*, const double*, const unsigned long long*, const double*, unsigned, unsigned, const PostTables, PostArgs
// the packed leading integers: 17 + 6 + 8 + 1 bits (the LDS kernel's rows are at most 16 384 + padding floats apart; run_post checks the rest)
static inline bool post_lead_fits(int ld_raw, int ld_theta, int n_filters) { return ld_raw >= 0 && ld_raw < (1 << 17) && ld_theta >= 0 && ld_theta < 64 && n_filters >= 0 && n_filters < 256; }
static inline unsigned post_lead_ints(int ld_raw, int ld_theta, int n_filters, int raw_freq) {
  return (unsigned)ld_raw | ((unsigned)ld_theta << 17) | ((unsigned)n_filters << 23) | ((unsigned)(raw_freq != 0) << 31);
}
#define PAYNE_POST_EXTERN(L, TW, LEAN) extern template __global__ void payne_post_kernel<L, TW, LEAN>(PAYNE_POST_SIG);
#define PAYNE_POST_DEFINE(L, TW, LEAN) template __global__ void payne_post_kernel<L, TW, LEAN>(PAYNE_POST_SIG);
#ifndef PAYNE_TU_POST_LEAN
PAYNE_POST_LEAN_LIST(PAYNE_POST_EXTERN)
#endif
#ifndef PAYNE_TU_POST_FULL_A
PAYNE_POST_FULL_A_LIST(PAYNE_POST_EXTERN)
#endif
#ifndef PAYNE_TU_POST_FULL_B
PAYNE_POST_FULL_B_LIST(PAYNE_POST_EXTERN)
#endif

typedef void (*post_kernel_fn)(PAYNE_POST_SIG);
// the instantiation pick_post_kernel returns, by name (payne_last_kernel)
static const char* post_kernel_label(int n1, bool tw_lds, bool lean) {
  if (lean) {
    if (tw_lds && n1 == 4096) return "payne_post_kernel<12, true, true>";
    if (tw_lds && n1 == 2048) return "payne_post_kernel<11, true, true>";
    if (tw_lds && n1 == 1024) return "payne_post_kernel<10, true, true>";
    if (!tw_lds && n1 == 8192) return "payne_post_kernel<13, false, true>";
  }
  if (tw_lds) return n1 == 1024 ? "payne_post_kernel<10, true>" : n1 == 2048 ? "payne_post_kernel<11, true>" : n1 == 4096 ? "payne_post_kernel<12, true>" : "payne_post_kernel<0, true>";
  return n1 == 8192 ? "payne_post_kernel<13, false>" : "payne_post_kernel<0, false>";
}
// compile-time FFT geometry for the common spectrum lengths, runtime geometry otherwise
static post_kernel_fn pick_post_kernel(int n1, bool tw_lds, bool lean = false) {
  if (lean) {                                   // likelihood-only builds of the LDS-twiddle sizes that matter
    if (tw_lds && n1 == 4096) return payne_post_kernel<12, true, true>;
    if (tw_lds && n1 == 2048) return payne_post_kernel<11, true, true>;
    if (tw_lds && n1 == 1024) return payne_post_kernel<10, true, true>;
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

