// nop_rate.hip -- what the `s_nop 0` costs that the compiler puts between an inline-asm statement and a vector instruction reading
// its result (it must assume the statement writes half a register: the dst_sel forwarding hazard of gfx940+), and what the
// `v_cmp -> s_nop 1 -> v_cndmask` form of a NaN scrub costs against forms without a condition register.
//   hipcc --offload-arch=gfx950 -O3 -o nop_rate tools/exp/nop_rate.hip && ./nop_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int NI = 64, REP = 256;

template <int KIND>
__global__ void k(float* out, unsigned long long* cyc, float seed) {
  f2 a[8];
  float s[16];
  for (int i = 0; i < 8; ++i) a[i] = (f2){seed + i, seed - i};
  for (int i = 0; i < 16; ++i) s[i] = seed + 0.5f * i;
  const f2 c = (f2){1e-6f, -1e-6f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < REP; ++r) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (KIND == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(c));                       // dependent chain
      if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1\n\ts_nop 0" : "+v"(a[0]) : "v"(c));            // ... with the nop
      if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i & 7]) : "v"(c));                   // eight independent chains
      if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1\n\ts_nop 0" : "+v"(a[i & 7]) : "v"(c));
      if (KIND == 4) asm volatile("v_cmp_o_f32 vcc, %0, %0\n\ts_nop 1\n\tv_cndmask_b32 %0, 0, %0, vcc" : "+v"(s[i & 15]) : : "vcc");
      if (KIND == 5) asm volatile("v_max_f32 %0, %0, %0" : "+v"(s[i & 15]));                                // (one instruction, for scale)
      if (KIND == 6) asm volatile("v_pk_add_f32 %0, %0, %1\n\ts_nop 1" : "+v"(a[i & 7]) : "v"(c));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = 0.f;
  for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y;
  for (int i = 0; i < 16; ++i) acc += s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4 * 256 * 1024 * 4); hipMalloc(&cyc, 8 * 1024);
  for (int waves_per_simd : {1, 2, 4}) {
    const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    printf("%-44s waves/SIMD %d: %.2f ticks per statement per wave\n", name, waves_per_simd, s / 256 / (double)(NI * REP));
  }
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0>("v_pk_add dependent"); run<1>("v_pk_add dependent + s_nop 0"); run<2>("v_pk_add independent"); run<3>("v_pk_add independent + s_nop 0");
  run<6>("v_pk_add independent + s_nop 1"); run<4>("v_cmp_o / s_nop 1 / v_cndmask (3 instr)"); run<5>("v_max_f32 (1 instr)");
  return 0;
}
