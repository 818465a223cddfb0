// l2_persist.hip -- does an XCD's L2 keep READ-ONLY data across a kernel boundary, and does a workgroup's index tell its XCD?
// Kernel K: the 32 workgroups of XCD x (blockIdx & 7) each read the whole slice x (1 MB) of an 8 MB buffer once.  Timed (a) cold, after
// a 1 GB memset and a sync; (b) again right after (a); (c) after the memset and a PREFETCH kernel of 64 workgroups of 256 threads
// whose XCD-x workgroups read slice x; (d) after a prefetch kernel that reads the WRONG slice (x + 1).  It also prints the hardware's
// XCC id of the first sixteen workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_persist tools/exp/l2_persist.hip && /tmp/l2_persist
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(512) K(const f4* __restrict__ src, float* out, int* xcc) {
  const int xcd = blockIdx.x & 7;
  const f4* s = src + (size_t)xcd * (1 << 16);              // 1 MB = 65536 f4
  f4 a = {0, 0, 0, 0};
  for (int i = threadIdx.x + ((blockIdx.x >> 3) * 2048) % 65536, n = 0; n < 65536 / 512; ++n, i = (i + 512) & 65535) { const f4 v = s[i]; a += v; }
  out[blockIdx.x * 512 + threadIdx.x] = a.x + a.y + a.z + a.w;
  if (threadIdx.x == 0 && xcc) xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);   // HW_REG_XCC_ID, bits 3:0
}
__global__ void __launch_bounds__(256) P(const f4* __restrict__ src, float* out, int shift) {
  const int xcd = ((blockIdx.x & 7) + shift) & 7, j = blockIdx.x >> 3, per = gridDim.x >> 3;
  const f4* s = src + (size_t)xcd * (1 << 16);
  f4 a = {0, 0, 0, 0};
  for (int i = j * 256 + threadIdx.x; i < 65536; i += per * 256) { const f4 v = s[i]; a += v; }
  out[blockIdx.x * 256 + threadIdx.x] = a.x + a.y + a.z + a.w;
}
int main() {
  f4* d; float* out; char* big; int* xcc;
  (void)hipMalloc(&d, 8 << 20); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&big, 1u << 30); (void)hipMalloc(&xcc, 256 * 4);
  (void)hipMemset(d, 0, 8 << 20);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto timeK = [&](const char* what) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(K, dim3(256), dim3(512), 0, 0, d, out, xcc);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("  %-46s %.1f us\n", what, ms * 1e3);
  };
  for (int rep = 0; rep < 3; ++rep) {
    printf("round %d\n", rep);
    (void)hipMemset(big, 1, 1u << 30); (void)hipDeviceSynchronize();
    timeK("(a) cold");
    timeK("(b) again");
    (void)hipMemset(big, 1, 1u << 30); (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(P, dim3(64), dim3(256), 0, 0, d, out, 0);
    timeK("(c) after the prefetch kernel");
    (void)hipMemset(big, 1, 1u << 30); (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(P, dim3(64), dim3(256), 0, 0, d, out, 1);
    timeK("(d) after a prefetch of the neighbour's slice");
  }
  int h[256]; (void)hipMemcpy(h, xcc, sizeof(h), hipMemcpyDeviceToHost);
  printf("XCC id of workgroups 0..15:"); for (int i = 0; i < 16; ++i) printf(" %d", h[i] & 15); printf("\n");
  return 0;
}
