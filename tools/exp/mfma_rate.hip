// mfma_rate.hip -- how fast does a SIMD of gfx950 retire v_mfma_f32_32x32x2_f32 (the output layer's matrix
// instruction), as a function of waves per SIMD and of how the instructions depend on each other?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate tools/exp/mfma_rate.hip && ./mfma_rate
// KIND 0: one accumulator per wave (every instruction waits for the previous one: the output layer's chain)
// KIND 1: two accumulators, alternating        KIND 2: four accumulators
// KIND 3: one accumulator, 16x16x4 instead (32 cycles of pipe each)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NI = 16, REP = 512;

template <int KIND>
__global__ void k(float* out, unsigned long long* cyc, float seed) {
  f32x16 acc[4];
  f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  const float a = seed + threadIdx.x * 1e-3f, b = seed - threadIdx.x * 1e-3f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < REP; ++r) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (KIND == 0) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
      if (KIND == 1) acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 1], 0, 0, 0);
      if (KIND == 2) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 3], 0, 0, 0);
      if (KIND == 3) acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = acc4.x + acc4.y;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, double flop) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4 * 256 * 1024); hipMalloc(&cyc, 8 * 1024);
  for (int wps : {1, 2, 4}) {
    const int threads = 256 * wps;
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    const double per = s / 256 / (double)(NI * REP);
    const double tf = flop * NI * REP * wps * 4 * 256 / (ms * 1e-3) / 1e12;
    printf("%-34s waves/SIMD %d: %.1f counter ticks per instr per wave, %.1f per SIMD; kernel %.1f us -> %.1f TFLOP/s\n", name, wps, per, per / wps, ms * 1e3, tf);
  }
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("32x32x2 one accumulator", 4096);
  run<1>("32x32x2 two accumulators", 4096);
  run<2>("32x32x2 four accumulators", 4096);
  run<3>("16x16x4 one accumulator", 2048);
  return 0;
}
