// kernarg_preload.hip -- how long does a kernel wait for its first argument, loaded by s_load from the kernarg segment, and does the
// runtime on this box preload leading scalar arguments into SGPRs (-mllvm -amdgpu-kernarg-preload-count=N)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/kernarg_plain tools/exp/kernarg_preload.hip
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=8 -o tools/exp/kernarg_pre tools/exp/kernarg_preload.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
struct Big { int pad[100]; };
__global__ void k(unsigned long long* out, int n, int m, Big b) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int v = n * 3 + m;                       // needs the arguments
  asm volatile("s_nop 0" :: "s"(v));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = (unsigned long long)(v + b.pad[7]); }
}
int main() {
  unsigned long long* d; (void)hipMalloc(&d, 2 * 512 * 8);
  Big b{};
  std::vector<unsigned long long> h(1024);
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, d, rep + 1, 7, b);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), d, 1024 * 8, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> t;
    for (int i = 0; i < 512; ++i) t.push_back(h[2 * i]);
    std::sort(t.begin(), t.end());
    printf("launch %d: cycles from kernel entry to the arguments in registers: min %llu median %llu p90 %llu max %llu\n", rep, t[0], t[256], t[460], t[511]);
  }
  return 0;
}
