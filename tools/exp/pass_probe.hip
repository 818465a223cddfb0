// pass_probe.hip -- what does one barrier-separated FFT pass of the post kernel cost, and which part of it?
// One 256-thread workgroup per CU runs ITER passes over a 2048-point complex buffer in LDS (the real
// fft_pass_fixed<8, 2048, 8, 256> of post_core.hpp) in several reduced forms:
//   0 full pass + barrier | 1 LDS reads + writes + barrier, no arithmetic | 2 arithmetic only (registers), no LDS, no barrier
//   3 barrier only | 4 reads only + barrier | 5 writes only + barrier | 6 full pass, two workgroups per CU
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o tools/exp/pass_probe tools/exp/pass_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../thepayne_amd/csrc/post_seq.hpp"
using namespace payne;
constexpr int M = 2048, NT = 256, ITER = 400;
#ifndef PROBE_THREADS
#define PROBE_THREADS 256
#endif
typedef float f2v_ __attribute__((ext_vector_type(2)));

struct ProbeEx {
#ifdef __HIP_DEVICE_COMPILE__
  static __device__ __forceinline__ auto buf(c32* p) { return (PAYNE_AS_LDS f2v*)p; }
  static __device__ __forceinline__ auto twid(const c32* p) { return (const PAYNE_AS_LDS f2v*)p; }
  static __device__ __forceinline__ auto lds(c32* p) { return (PAYNE_AS_LDS f2v*)p; }
#endif
  template <class F> __device__ __forceinline__ void par(F&& f) { f((int)threadIdx.x, (int)blockDim.x); __syncthreads(); }
  template <class F> __device__ __forceinline__ void single(F&& f) { if (threadIdx.x == 0) f((int)blockDim.x); }
  __device__ __forceinline__ void mark(int) {}
  __device__ __forceinline__ int nthreads() const { return (int)blockDim.x; }
  __device__ __forceinline__ c32* tile() const { return nullptr; }
  __device__ __forceinline__ bool fuse() const { return false; }
};

template <int KIND>
__global__ void __launch_bounds__(PROBE_THREADS) probe(float* out, unsigned long long* cyc, const c32* twg) {
#ifdef __HIP_DEVICE_COMPILE__
  extern __shared__ __attribute__((aligned(16))) float sm[];
  c32* A = reinterpret_cast<c32*>(sm);
  c32* Bf = A + M + M / 8;
  c32* tw = Bf + M + M / 8;
  const int tid = threadIdx.x;
  for (int i = tid; i < M + M / 8; i += (int)blockDim.x) { A[i] = {1.0f + i * 1e-6f, 0.5f}; Bf[i] = {0.f, 0.f}; }
  for (int i = tid; i < plan_table_len(M); i += (int)blockDim.x) tw[i] = twg[i];
  __syncthreads();
  auto a = (PAYNE_AS_LDS f2v*)A;
  auto b = (PAYNE_AS_LDS f2v*)Bf;
  auto t = (const PAYNE_AS_LDS f2v*)tw;
  c32 u[8];
  for (int r = 0; r < 8; ++r) u[r] = {1.0f + tid * 1e-3f + r, 0.25f * r};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
    if (KIND == 0 || KIND == 6) {
      fft_pass_fixed<8, M, 8, NT>(tid, a, b, t, 0u);
      __syncthreads();
      auto x = a; a = b; b = x;
    } else if (KIND == 7 || KIND == 8) {             // the kernel's whole 2048-point transform: 8 x 8 x 8 x 4
      fft_pass_fixed<8, M, 1, NT>(tid, a, b, t, 0u);
      __syncthreads();
      fft_pass_fixed<8, M, 8, NT>(tid, b, a, t, 0u);
      __syncthreads();
      fft_pass_fixed<8, M, 64, NT>(tid, a, b, t, 0u);
      __syncthreads();
      fft_pass_fixed<4, M, 512, NT>(tid, b, a, t, 0u);
      __syncthreads();
    } else if (KIND == 9 || KIND == 10) {            // the kernel's own out-of-line transform
      ProbeEx ex;
      c32* r = fft_fixed<M, PROBE_THREADS>(ex, A, Bf, tw, 0u, false);
      if (r != A && it == ITER - 1) u[0].x += 1.f;
    } else if (KIND == 1) {
#pragma unroll
      for (int r = 0; r < 8; ++r) u[r] = ldc(a, tid + r * 256);
#pragma unroll
      for (int r = 0; r < 8; ++r) stc(b, (tid >> 3) * 72 + (tid & 7) + r * 8, u[r]);
      __syncthreads();
      auto x = a; a = b; b = x;
    } else if (KIND == 2) {
      c32 w[8];
#pragma unroll
      for (int r = 1; r < 8; ++r) w[r] = {0.999f, 0.001f * r};
#pragma unroll
      for (int r = 1; r < 8; ++r) u[r] = cmul(u[r], w[r]);
      dft8(u);
#pragma unroll
      for (int r = 0; r < 8; ++r) asm volatile("" : "+v"(u[r].x), "+v"(u[r].y));
    } else if (KIND == 3) {
      __syncthreads();
    } else if (KIND == 4) {
#pragma unroll
      for (int r = 0; r < 8; ++r) { c32 v = ldc(a, tid + r * 256); u[r].x += v.x; u[r].y += v.y; }
      __syncthreads();
    } else if (KIND == 5) {
#pragma unroll
      for (int r = 0; r < 8; ++r) stc(b, (tid >> 3) * 72 + (tid & 7) + r * 8, u[r]);
      __syncthreads();
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0.f;
  for (int r = 0; r < 8; ++r) acc += u[r].x + u[r].y;
  acc += A[tid].x + Bf[tid].y;
  out[blockIdx.x * 512 + tid] = acc;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
#endif
}

template <int KIND>
void run(const char* name, const c32* tw, int grid) {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, (size_t)grid * 512 * 4); (void)hipMalloc(&cyc, (size_t)grid * 8);
  const size_t lds = (size_t)(2 * (M + M / 8) + plan_table_len(M)) * 8;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(KIND == 9 || KIND == 10 ? PROBE_THREADS : NT), lds, 0, out, cyc, tw);
  (void)hipEventRecord(e0, 0);
  for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(KIND == 9 || KIND == 10 ? PROBE_THREADS : NT), lds, 0, out, cyc, tw);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(grid);
  (void)hipMemcpy(h.data(), cyc, (size_t)grid * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += (double)v;
  printf("%-52s %7.0f cycles per iteration (grid %d); kernel %.1f us = %.0f cycles in-kernel -> >= %.2f GHz\n", name, s / grid / ITER, grid,
         ms * 100.0, s / grid, s / grid / (ms * 100.0) / 1e3);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  std::vector<c32> tw(plan_table_len(M));
  for (size_t i = 0; i < tw.size(); ++i) tw[i] = {cosf(0.001f * i), -sinf(0.001f * i)};
  c32* d; (void)hipMalloc(&d, tw.size() * 8); (void)hipMemcpy(d, tw.data(), tw.size() * 8, hipMemcpyHostToDevice);
  run<0>("full radix-8 pass + barrier", d, 256);
  run<6>("full radix-8 pass + barrier, 2 workgroups per CU", d, 512);
  run<7>("whole transform (4 passes)", d, 256);
  run<8>("whole transform (4 passes), 2 workgroups per CU", d, 512);
  run<9>("fft_fixed<2048,256> as the kernel calls it", d, 256);
  run<10>("fft_fixed<2048,256>, 2 workgroups per CU", d, 512);
  run<1>("LDS reads + writes + barrier (no arithmetic)", d, 256);
  run<2>("arithmetic only (7 cmul + dft8), registers", d, 256);
  run<3>("barrier only", d, 256);
  run<4>("8 reads + barrier", d, 256);
  run<5>("8 writes + barrier", d, 256);
  return 0;
}
