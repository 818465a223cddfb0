// lds_floor.hip -- what SQ_LDS_BANK_CONFLICT reads for access patterns that are conflict-free by MI355X_MICROARCH.md's banking
// table, and for the post kernel's two patterns that are not (first radix-8 pass's padded stores; the observed-grid gather).
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/lds_floor tools/exp/lds_floor.hip
//   rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -- tools/exp/lds_floor
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define LDS __attribute__((address_space(3)))
constexpr int IT = 2000;
__shared__ float sm[16384];
template <int MODE>
__global__ void __launch_bounds__(512) k(float* out) {
  const int t = threadIdx.x;
  LDS f2* p = (LDS f2*)sm;
  LDS float* q = (LDS float*)sm;
  f2 acc = {0.f, 0.f};
  float a1 = 0.f;
  int idx;
  if (MODE == 0 || MODE == 1) idx = t;                                  // consecutive 8-byte slots
  if (MODE == 2) idx = 8 * (t & 255) + ((8 * (t & 255)) >> 5);          // first radix-8 pass: slot 8 i + (i >> 2) (+ r)
  if (MODE == 3) idx = 72 * ((t & 255) >> 3) + (t & 7);                 // second pass: 72 g + k (+ 8 r)
  if (MODE == 4 || MODE == 5) idx = (int)(1.1375f * (float)t);         // gather at 1.14 words a lane (4-byte reads)
  if (MODE == 6) idx = (int)(0.88f * (float)t);
  if (MODE == 7) idx = (int)(1.1375f * (float)t);                       // the pair (k, k + 1) as ONE 8-byte read at a 4-byte-aligned address
  for (int it = 0; it < IT; ++it) {
    if (MODE == 0 || MODE == 2 || MODE == 3) {
#pragma unroll
      for (int r = 0; r < 8; ++r) { f2 v = {(float)it, (float)r}; *(volatile LDS f2*)(p + ((idx + r * (MODE == 3 ? 8 : (MODE == 0 ? 512 : 1))) & 8191)) = v; }
    } else if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 8; ++r) { f2 v = *(volatile LDS f2*)(p + ((idx + r * 512) & 8191)); acc += v; }
    } else if (MODE == 7) {
#pragma unroll
      for (int r = 0; r < 8; ++r) { const int j = (idx + r * 583) & 16382; f2 v; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(j * 4) : "memory"); acc += v; }
    } else if (MODE == 4 || MODE == 6) {
#pragma unroll
      for (int r = 0; r < 8; ++r) { a1 += *(volatile LDS float*)(q + ((idx + r * 583) & 16383)); }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) { const int j = (idx + r * 583) & 16382; a1 += *(volatile LDS float*)(q + j); a1 += *(volatile LDS float*)(q + j + 1); }
    }
  }
  out[blockIdx.x * 512 + t] = acc.x + acc.y + a1;
}
int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, d);
  hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, d);
  hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, d);
  hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, d);
  hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, d);
  hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, d);
  hipLaunchKernelGGL(k<6>, dim3(256), dim3(512), 0, 0, d);
  hipLaunchKernelGGL(k<7>, dim3(256), dim3(512), 0, 0, d);
  hipDeviceSynchronize();
  printf("done\n");
  return 0;
}
