// census.hip -- the phases of payne_post_kernel<12, true, true> (C2: 4096 points, rows in the frequency domain) as kernels of their
// own, with the template arguments run_candidate gives them, so that `tools/valu_census.py` can count the instructions of each phase
// in the compiler's output (loops with compile-time trip counts: the static count IS what a wave executes).  Device code only; never
// linked into the library, never run.
#include <hip/hip_runtime.h>
#include "../../thepayne_amd/csrc/post_seq.hpp"
using namespace payne;
#include "../../thepayne_amd/csrc/select.hpp"
#include "../../thepayne_amd/csrc/sampler_core.hpp"

struct DevEx {
  static constexpr bool kTwLds = true;
  static __device__ __forceinline__ auto buf(c32* p) { return (PAYNE_AS_LDS f2v*)p; }
  static __device__ __forceinline__ auto twid(const c32* p) { return (const PAYNE_AS_LDS f2v*)p; }
  static __device__ __forceinline__ auto lds(c32* p) { return (PAYNE_AS_LDS f2v*)p; }
  template <class F> __device__ __forceinline__ void par(F&& f) { f((int)threadIdx.x, (int)blockDim.x); __syncthreads(); }
  template <class F> __device__ __forceinline__ void single(F&& f) { if (threadIdx.x == 0) f((int)blockDim.x); }
  __device__ __forceinline__ void mark(int) {}
  __device__ __forceinline__ int nthreads() const { return (int)blockDim.x; }
  __device__ __forceinline__ c32* tile() const { return nullptr; }
  __device__ __forceinline__ bool fuse() const { return false; }
};
constexpr int NT = 512, MF = 2048, UX = 8;
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

// first phase: the row's slots + the record requested, the taper applied on the way to LDS
__global__ void __launch_bounds__(NT) census_first(const PostTables T, const float* raw, const CandState* prep, const double* th) {
  float* bufB = reinterpret_cast<float*>(smem);
  CandState* S = reinterpret_cast<CandState*>(bufB + 2 * fft_buf_floats(4096));
  const int t = threadIdx.x;
  SlotRegs<2> slots;
  slots_issue<2>(t, NT, MF, raw + (size_t)blockIdx.x * 4096, T.twf + plan_total(MF), slots);
  PrepRegs pr;
  phase_take_prep_issue(t, prep + blockIdx.x, pr);
  const double th5 = th[blockIdx.x * 12 + 5];
  phase_take_prep_commit(t, pr, *S);
  taper_slots_fast<2>(vsini_taper_args(T, th5), t, NT, MF, slots);            // (every bin inside the table: the usual case)
  slots_store<2>(t, NT, MF, slots, DevEx::buf((c32*)bufB), th5 != 0.0);
  __syncthreads();
}
// one transform (out of line in the product kernel as well)
__global__ void __launch_bounds__(NT) census_fft(int sel) {
  float* a = reinterpret_cast<float*>(smem);
  float* b = a + fft_buf_floats(4096);
  const c32* twf = reinterpret_cast<const c32*>(b + fft_buf_floats(4096));
  DevEx ex;
  c32* r = fft_fixed<MF, NT>(ex, (c32*)a, (c32*)b, twf, sel ? 0x80000000u : 0u, sel > 1);
  if (r == nullptr) __builtin_trap();
}
// resampling onto the candidate's window (Doppler shift + mask)
__global__ void __launch_bounds__(NT) census_resample(const PostTables T) {
  float* a = reinterpret_cast<float*>(smem);
  float* b = a + fft_buf_floats(4096);
  CandState* S = reinterpret_cast<CandState*>(b + fft_buf_floats(4096));
  const Window W = S->W;
  if (R_resample_fast<UX>((int)threadIdx.x, NT, W, a, b)) __builtin_trap();   // (a window of 4096 points, no NaN: the usual case)
  __syncthreads();
}
__global__ void __launch_bounds__(NT) census_resample_general(const PostTables T) {
  float* a = reinterpret_cast<float*>(smem);
  float* b = a + fft_buf_floats(4096);
  CandState* S = reinterpret_cast<CandState*>(b + fft_buf_floats(4096));
  const Window W = S->W;
  R_resample_loop<true, UX>((int)threadIdx.x, NT, T, *S, W, a, b);
  __syncthreads();
}
// the instrumental stage's middle step (Gaussian taper on the conjugate pairs)
__global__ void __launch_bounds__(NT) census_taper() {
  float* a = reinterpret_cast<float*>(smem);
  float* b = a + fft_buf_floats(4096);
  CandState* S = reinterpret_cast<CandState*>(b + fft_buf_floats(4096));
  const c32* twf = reinterpret_cast<const c32*>(S + 1);
  TaperArgs ta{};
  ta.g_c2 = S->W.g_c2;
  rfft_taper_phase<false, UX / 4>((int)threadIdx.x, NT, DevEx::buf((c32*)a), MF, DevEx::twid(twf + plan_total(MF)), 1, ta);
  __syncthreads();
}
// observed grid, chi^2
__global__ void __launch_bounds__(NT) census_obs(const PostTables T, double* out) {
  float* a = reinterpret_cast<float*>(smem);
  float* b = a + fft_buf_floats(4096);
  CandState* S = reinterpret_cast<CandState*>(b + fft_buf_floats(4096));
  double* red = reinterpret_cast<double*>(S + 1);
  const Window W = S->W;
  bool bad;
  const float acc = obs_loop_fast<UX>((int)threadIdx.x, NT, T, W, a, bad);        // (every pixel inside the window: the usual case)
  if (bad) __builtin_trap();
  store_partial((int)threadIdx.x, (double)acc, red);
  __syncthreads();
  if (threadIdx.x == 0) { double s = 0.0; for (int i = 0; i < 8; ++i) s += red[i]; out[blockIdx.x] = -0.5 * s; }
}
__global__ void __launch_bounds__(NT) census_obs_general(const PostTables T, double* out) {
  float* a = reinterpret_cast<float*>(smem);
  float* b = a + fft_buf_floats(4096);
  CandState* S = reinterpret_cast<CandState*>(b + fft_buf_floats(4096));
  double* red = reinterpret_cast<double*>(S + 1);
  const Window W = S->W;
  const float acc = obs_loop<0, false, true, false, UX>((int)threadIdx.x, NT, T, *S, W, a, nullptr, -1);
  store_partial((int)threadIdx.x, (double)acc, red);
  __syncthreads();
  if (threadIdx.x == 0) { double s = 0.0; for (int i = 0; i < 8; ++i) s += red[i]; out[blockIdx.x] = -0.5 * s; }
}
// grids that are not 2^k long: the rotation stage's way back onto the model grid (C2r)
__global__ void __launch_bounds__(NT) census_rot_back(const PostTables T) {
  float* a = reinterpret_cast<float*>(smem);
  float* b = a + fft_buf_floats(4096);
  phase_rot_back<UX>((int)threadIdx.x, NT, T, a, b, true);
  __syncthreads();
}
