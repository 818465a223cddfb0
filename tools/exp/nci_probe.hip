// Latency of one wave's inverse normal CDF: the device library's normcdfinv against a Chebyshev form (Clenshaw), dependent calls.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/nci_probe.hip -o gpurun_out/nci_probe && gpurun_out/nci_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__device__ __forceinline__ double clenshaw(const double* a, int n, double z) {
  double b1 = 0.0, b2 = 0.0;
  const double z2 = 2.0 * z;
  for (int k = n - 1; k > 0; --k) { const double t = fma(z2, b1, a[k] - b2); b2 = b1; b1 = t; }
  return fma(z, b1, a[0] - b2);
}
__constant__ double kA[29];
template <int MODE>
__global__ void probe(double* io, unsigned long long* cyc, int reps) {
  double p = io[threadIdx.x];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  double acc = 0.0;
  for (int i = 0; i < reps; ++i) {
    double x;
    if (MODE == 0) x = normcdfinv(p);
    else if (MODE == 1) { const double q = p - 0.5, r = q * q; x = q * clenshaw(kA, 29, (r - 0.0903125) * (1.0 / 0.0903125)); }
    else { const double t = sqrt(-log(p)); x = -clenshaw(kA, 25, (t - 3.3) * (1.0 / 1.7)); }
    acc += x;
    p = p + x * 1e-18 + 1e-9;          // dependent on the result, numerically the same argument
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  io[threadIdx.x] = acc;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  double h[64]; double* d; unsigned long long* c; unsigned long long hc;
  double a[29]; for (int i = 0; i < 29; ++i) a[i] = 1.0 / (1 + i * i);
  hipMemcpyToSymbol(HIP_SYMBOL(kA), a, sizeof(a));
  hipMalloc(&d, sizeof(h)); hipMalloc(&c, 8);
  const int reps = 200;
  for (int mode = 0; mode < 3; ++mode)
    for (int which = 0; which < 3; ++which) {
      for (int i = 0; i < 64; ++i) h[i] = which == 0 ? 0.3 + 0.005 * i : which == 1 ? 0.01 + 0.0005 * i : (i == 5 ? 0.003 : 0.3 + 0.005 * i);
      hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
      for (int w = 0; w < 2; ++w) {
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, d, c, reps);
        if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, d, c, reps);
        if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, d, c, reps);
        hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
      }
      printf("mode %d (%s) args %s: %.0f cycles a call\n", mode, mode == 0 ? "normcdfinv" : mode == 1 ? "central Clenshaw 29" : "log+sqrt+Clenshaw 25",
             which == 0 ? "central" : which == 1 ? "tail" : "mixed", (double)hc / reps);
    }
  return 0;
}
