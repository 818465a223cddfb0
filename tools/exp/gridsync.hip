// Feasibility probe for a fused persistent kernel: cost of grid-wide synchronisation on MI355X.
//   a) cooperative_groups grid.sync()   b) hand-written barrier (device-scope atomics, no cache maintenance)
//   c) producer/consumer flag hand-off with a device-scope release/acquire pair
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }

__global__ void k_cgsync(int n, unsigned long long* out) {
  cg::grid_group g = cg::this_grid();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) g.sync();
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (t1 - t0) / n;
}

// sense-free counting barrier: every workgroup adds 1, waits until count >= target (monotonic counter)
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 24)) break;           // watchdog: never hang the box
    }
  }
  __syncthreads();
}
__global__ void k_mybar(int n, unsigned* ctr, unsigned long long* out) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) grid_barrier(ctr, (unsigned)(i + 1) * gridDim.x);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) out[1] = (t1 - t0) / n;
}
// the same with full release/acquire fences at device scope around it (what real data hand-off needs)
__global__ void k_mybar_fenced(int n, unsigned* ctr, unsigned long long* out, float* data) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
    data[(size_t)blockIdx.x * 256 + threadIdx.x] = (float)i;                 // something to publish
    __atomic_thread_fence(__ATOMIC_RELEASE);                                    // device scope in HIP
    grid_barrier(ctr, (unsigned)(i + 1) * gridDim.x);
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    float v = data[(size_t)((blockIdx.x + 37) % gridDim.x) * 256 + threadIdx.x];  // read a neighbour's value
    if (v != (float)i && threadIdx.x == 0) atomicAdd((unsigned*)&out[3], 1u);   // stale reads counted
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) out[2] = (t1 - t0) / n;
}

int main() {
  const int G = 512, T = 256, N = 50;
  unsigned long long* out; unsigned* ctr; float* data;
  hipMalloc(&out, 64); hipMemset(out, 0, 64);
  hipMalloc(&ctr, 4);
  hipMalloc(&data, (size_t)G * T * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  // launch cost of empty kernels back to back
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_empty, dim3(G), dim3(T), 65536, 0, nullptr);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(G), dim3(T), 65536, 0, nullptr);
  hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  printf("empty kernel (512 x 256, 64 KB LDS) back to back: %.2f us each\n", ms * 1e3 / 200);
  int n = N; void* args1[] = {&n, &out};
  hipError_t e = hipLaunchCooperativeKernel((void*)k_cgsync, dim3(G), dim3(T), args1, 65536, 0);
  printf("cooperative launch: %s\n", hipGetErrorString(e));
  hipDeviceSynchronize();
  hipMemset(ctr, 0, 4);
  void* args2[] = {&n, &ctr, &out};
  e = hipLaunchCooperativeKernel((void*)k_mybar, dim3(G), dim3(T), args2, 65536, 0);
  hipDeviceSynchronize();
  hipMemset(ctr, 0, 4);
  void* args3[] = {&n, &ctr, &out, &data};
  e = hipLaunchCooperativeKernel((void*)k_mybar_fenced, dim3(G), dim3(T), args3, 65536, 0);
  hipDeviceSynchronize();
  unsigned long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
  printf("grid.sync(): %llu cycles | atomic barrier: %llu cycles | + release/acquire fences: %llu cycles, stale reads %llu\n", h[0], h[1], h[2], h[3]);
  // event-timed versions (wall) for the same
  hipMemset(ctr, 0, 4);
  hipEventRecord(e0);
  hipLaunchCooperativeKernel((void*)k_mybar_fenced, dim3(G), dim3(T), args3, 65536, 0);
  hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  printf("fenced barrier kernel, %d barriers: %.1f us total -> %.2f us per barrier (incl. launch)\n", N, ms * 1e3, ms * 1e3 / N);
  return 0;
}
