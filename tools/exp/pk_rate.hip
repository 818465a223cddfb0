// pk_rate.hip -- issue cost of packed-fp32 VALU instructions on gfx950 next to plain fp32/fp64 ones, at 1, 2 and 4
// waves per SIMD (decides whether the post kernel's complex arithmetic should be written with v_pk_*).
//   hipcc --offload-arch=gfx950 -O3 -o pk_rate tools/exp/pk_rate.hip && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int NI = 64, REP = 256;

template <int KIND>
__global__ void k(float* out, unsigned long long* cyc, float seed) {
  f2 a[8];
  double d[8];
  float s[16];
  for (int i = 0; i < 8; ++i) { a[i] = (f2){seed + i, seed - i}; d[i] = seed + i; }
  for (int i = 0; i < 16; ++i) s[i] = seed + 0.5f * i;
  const f2 m = (f2){1.0001f, 0.9999f}, c = (f2){1e-6f, -1e-6f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < REP; ++r) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i & 15]) : "v"(m.x), "v"(c.x));
      if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 7]) : "v"(m), "v"(c));
      if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i & 7]) : "v"(c));
      if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i & 7]) : "v"(m));
      if (KIND == 4) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i & 7]) : "v"((double)m.x), "v"((double)c.x));
      if (KIND == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i & 15]) : "v"(c.x));
      if (KIND == 6) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i & 7]) : "v"(c));
      if (KIND == 7) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i & 7]) : "v"((double)c.x));
      if (KIND == 8) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i & 7]) : "v"((double)m.x));
      if (KIND == 9) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(s[i & 15]) : "v"(d[i & 7]));
      if (KIND == 10) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i & 7]) : "v"(s[i & 15]));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0.f;
  for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y + (float)d[i];
  for (int i = 0; i < 16; ++i) acc += s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4 * 256 * 1024 * 4); hipMalloc(&cyc, 8 * 1024);
  for (int waves_per_simd : {1, 2, 4}) {
    const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;     // one workgroup per CU: 4 SIMDs x waves
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    const double per = s / 256 / (double)(NI * REP);                    // cycles per instruction as one wave sees it
    printf("%-28s waves/SIMD %d: %.2f cyc per instr per wave -> %.2f cyc per instr per SIMD\n", name, waves_per_simd, per, per / waves_per_simd);
  }
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0>("v_fma_f32"); run<5>("v_add_f32"); run<1>("v_pk_fma_f32"); run<2>("v_pk_add_f32"); run<3>("v_pk_mul_f32");
  run<6>("v_pk_add_f32 op_sel/neg"); run<4>("v_fma_f64"); run<7>("v_add_f64"); run<8>("v_mul_f64");
  run<9>("v_cvt_f32_f64"); run<10>("v_cvt_f64_f32");
  return 0;
}
