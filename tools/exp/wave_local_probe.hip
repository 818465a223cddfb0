// wave_local_probe.hip -- the 2048-point transform of the post kernel with its first three radix-8 passes WAVE-LOCAL.
// The Stockham plan 8 x 8 x 8 x 4 computes, in its first three passes, four independent 512-point transforms of the subsequences
// x[w + 4 n] -- interleaved over all lanes.  Re-assigned so that wave w owns subsequence w (its 512 points = 64 lanes x 8), the three
// passes touch only that wave's 576-element block of either buffer: NO s_barrier between them (a wave's LDS operations execute in order),
// the same twiddle table, the same padded layouts (those of a 512-point plan on 64 threads).  One barrier, then the radix-4 pass.
//   kinds: 0 = the kernel's four passes (barrier after each), 1 = wave-local passes + barrier + radix-4 pass + barrier (4 waves local, 4 or 8 cross),
//   each alone on a CU and two workgroups per CU; --check compares the two transforms' outputs on the same input.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o tools/exp/wave_local_probe tools/exp/wave_local_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../../thepayne_amd/csrc/post_seq.hpp"
using namespace payne;
constexpr int M = 2048, ITER = 400, BLK = 576;

#ifdef __HIP_DEVICE_COMPILE__
// one radix-8 pass of a wave's own 512-point transform: lane = butterfly (64 of them)
template <int P, int OFF, class SP, class DP, class TP>
__device__ __forceinline__ void pass_local8(int lane, SP src, DP dst, TP twf) {
  constexpr int R = 8, NB = 64;
  constexpr int PI = (P == 1) ? 0 : (P == 8 ? 1 : 8), RI = (P == 1) ? 0 : 8;
  constexpr bool last = (P == 64);
  constexpr int PO = last ? 0 : P;
  int i = lane;
  if constexpr (P == 1) i = (lane & ~31) | ((lane & 14) << 1) | ((lane >> 3) & 2) | (lane & 1);
  const int k = i & (P - 1);
  const int ib = fft_lay<PI, RI>(i);
  c32 u[R];
#pragma unroll
  for (int r = 0; r < R; ++r) u[r] = ldc1(src, ib + fft_lay<PI, RI>(r * NB));
  if constexpr (P > 1) {
    c32 w[R];
    w[1] = ldc1(twf, OFF + k); w[2] = ldc1(twf, OFF + P + k); w[4] = ldc1(twf, OFF + 2 * P + k);
    w[3] = cmul(w[1], w[2]); w[5] = cmul(w[1], w[4]); w[6] = cmul(w[2], w[4]); w[7] = cmul(w[3], w[4]);
    twiddle_all<R>(u, w);
  }
  dftR<R>(u);
  const int j = fft_lay<PO, R>((i - k) * R + k);
#pragma unroll
  for (int r = 0; r < R; ++r) stc(dst, j + r * P, u[r]);
}
// the closing radix-4 pass over the four waves' blocks (BLK apart), plain output
template <int NT, class SP, class DP, class TP>
__device__ __forceinline__ void pass_cross4(int tid, SP src, DP dst, TP twf) {
  constexpr int OFF = plan_offset(M, 512);
#pragma unroll
  for (int i0 = 0; i0 < 512; i0 += NT) {
    const int i = i0 + tid;
    c32 u[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) u[r] = ldc1(src, i + r * BLK);
    c32 w[4];
    w[1] = ldc1(twf, OFF + i); w[2] = cmul(w[1], w[1]); w[3] = cmul(w[2], w[1]);
    twiddle_all<4>(u, w);
    dftR<4>(u);
#pragma unroll
    for (int r = 0; r < 4; ++r) stc(dst, i + r * 512, u[r]);
  }
}
#endif

template <int KIND, int NT>
__global__ void __launch_bounds__(NT) probe(float* out, unsigned long long* cyc, const c32* twg, const c32* xin, c32* xout, int iters) {
#ifdef __HIP_DEVICE_COMPILE__
  extern __shared__ __attribute__((aligned(16))) float sm[];
  c32* A = reinterpret_cast<c32*>(sm);
  c32* Bf = A + M + M / 8;
  c32* tw = Bf + M + M / 8;
  const int tid = threadIdx.x;
  for (int i = tid; i < M + M / 8; i += NT) { A[i] = {0.f, 0.f}; Bf[i] = {0.f, 0.f}; }
  __syncthreads();
  for (int n = tid; n < M; n += NT) {
    if (KIND == 0) A[n] = xin[n];
    else A[BLK * (n & 3) + (n >> 2)] = xin[n];               // subsequence w = n mod 4 in wave w's block
  }
  for (int i = tid; i < plan_table_len(M); i += NT) tw[i] = twg[i];
  __syncthreads();
  auto a = (PAYNE_AS_LDS f2v*)A;
  auto b = (PAYNE_AS_LDS f2v*)Bf;
  auto t = (const PAYNE_AS_LDS f2v*)tw;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
      if (tid < 256) fft_pass_fixed<8, M, 1, 256>(tid, a, b, t, 0u);
      __syncthreads();
      if (tid < 256) fft_pass_fixed<8, M, 8, 256>(tid, b, a, t, 0u);
      __syncthreads();
      if (tid < 256) fft_pass_fixed<8, M, 64, 256>(tid, a, b, t, 0u);
      __syncthreads();
      if (tid < 256) fft_pass_fixed<4, M, 512, 256>(tid, b, a, t, 0u);
      __syncthreads();
    } else {
      if (tid < 256) {
        const int w = tid >> 6, l = tid & 63;
        auto aw = a + BLK * w; auto bw = b + BLK * w;
        pass_local8<1, 0>(l, aw, bw, t);
        asm volatile("" ::: "memory");
        pass_local8<8, plan_offset(M, 8)>(l, bw, aw, t);
        asm volatile("" ::: "memory");
        pass_local8<64, plan_offset(M, 64)>(l, aw, bw, t);
      }
      __syncthreads();
      if (KIND == 1) { if (tid < 256) pass_cross4<256>(tid, b, a, t); }
      else pass_cross4<NT>(tid, b, a, t);
      __syncthreads();
      if (iters > 1) {                                       // (timing loop: the output is the next input, whatever its order)
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (xout && blockIdx.x == 0) for (int n = tid; n < M; n += NT) xout[n] = A[n];
  out[blockIdx.x * 512 + (tid & 511)] = A[tid].x + Bf[tid].y;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
#endif
}

template <int KIND, int NT>
double run(const char* name, const c32* tw, const c32* xin, c32* xout, int grid, int iters, bool print = true) {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, (size_t)grid * 512 * 4); (void)hipMalloc(&cyc, (size_t)grid * 8);
  const size_t lds = (size_t)(2 * (M + M / 8) + plan_table_len(M)) * 8;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<KIND, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<KIND, NT>), dim3(grid), dim3(NT), lds, 0, out, cyc, tw, xin, xout, iters);
  (void)hipEventRecord(e0, 0);
  for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL((probe<KIND, NT>), dim3(grid), dim3(NT), lds, 0, out, cyc, tw, xin, xout, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(grid);
  (void)hipMemcpy(h.data(), cyc, (size_t)grid * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += (double)v;
  if (print) printf("%-64s %7.0f cycles per transform (grid %d, %d threads); kernel %.1f us\n", name, s / grid / iters, grid, NT, ms * 100.0);
  (void)hipFree(out); (void)hipFree(cyc);
  return s / grid / iters;
}

int main() {
  // the kernel's own twiddle table (host_tables.hpp builds the same): pass-ordered powers, then the split factors
  std::vector<c32> tw(plan_table_len(M));
  {
    int p = 1;
    while (p < M) {
      const int r = plan_radix(M, p);
      if (p > 1) {
        const int off = plan_offset(M, p);
        for (int j = 0; j < plan_tw_rows(r); ++j) for (int k = 0; k < p; ++k) {
          const int pw = (r == 8) ? (1 << j) : 1;
          const double ang = -2.0 * M_PI * (double)k * pw / ((double)p * r);
          tw[off + j * p + k] = {(float)cos(ang), (float)sin(ang)};
        }
      }
      p *= r;
    }
  }
  std::vector<c32> x(M);
  for (int n = 0; n < M; ++n) x[n] = {(float)sin(0.37 * n) + 0.25f * (float)cos(0.011 * n * n), (float)cos(0.73 * n) - 0.1f};
  c32 *d, *dx, *y0, *y1;
  (void)hipMalloc(&d, tw.size() * 8); (void)hipMemcpy(d, tw.data(), tw.size() * 8, hipMemcpyHostToDevice);
  (void)hipMalloc(&dx, M * 8); (void)hipMemcpy(dx, x.data(), M * 8, hipMemcpyHostToDevice);
  (void)hipMalloc(&y0, M * 8); (void)hipMalloc(&y1, M * 8);
  run<0, 512>("", d, dx, y0, 1, 1, false);
  run<2, 512>("", d, dx, y1, 1, 1, false);
  std::vector<c32> h0(M), h1(M);
  (void)hipMemcpy(h0.data(), y0, M * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(h1.data(), y1, M * 8, hipMemcpyDeviceToHost);
  // against a double-precision DFT at a few bins, and one against the other everywhere
  double worst = 0, wref = 0;
  for (int k = 0; k < M; ++k) worst = fmax(worst, fmax(fabs((double)h0[k].x - h1[k].x), fabs((double)h0[k].y - h1[k].y)));
  for (int k : {0, 1, 5, 511, 512, 1023, 1500, 2047}) {
    double re = 0, im = 0;
    for (int n = 0; n < M; ++n) { const double a = -2.0 * M_PI * (double)((long long)k * n % M) / M; re += x[n].x * cos(a) - x[n].y * sin(a); im += x[n].x * sin(a) + x[n].y * cos(a); }
    wref = fmax(wref, fmax(fabs(re - h1[k].x), fabs(im - h1[k].y)));
  }
  printf("check: wave-local vs four-pass max |d| = %.3g; wave-local vs fp64 DFT (8 bins) max |d| = %.3g\n", worst, wref);
  run<0, 512>("four passes, barrier after each (8 waves, 4 work)", d, dx, nullptr, 256, ITER);
  run<0, 512>("four passes, two workgroups per CU", d, dx, nullptr, 512, ITER);
  run<1, 512>("wave-local 8x8x8 + barrier + radix-4 on 4 waves", d, dx, nullptr, 256, ITER);
  run<1, 512>("... two workgroups per CU", d, dx, nullptr, 512, ITER);
  run<2, 512>("wave-local 8x8x8 + barrier + radix-4 on 8 waves", d, dx, nullptr, 256, ITER);
  run<2, 512>("... two workgroups per CU", d, dx, nullptr, 512, ITER);
  return 0;
}
