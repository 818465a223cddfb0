// stage_rate.hip -- how fast does a compute unit bring L2-resident operand tiles into LDS: by LDS-DMA (global_load_lds_dwordx4)
// or through registers (global_load_dwordx4 + ds_write_b128)?  One 512-thread workgroup per CU stages 36 KB "k-steps" (the
// output layer's stage: 576 rows of 64 bytes) out of a buffer that every workgroup walks from its own offset, wrapping within 1 MB
// (L2 resident) or 12 MB (Infinity-Cache resident),
// NSTEP steps, AHEAD steps requested ahead; a barrier per step as in the GEMM.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stage_rate tools/exp/stage_rate.hip && /tmp/stage_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int STAGE = 36864, NSTEP = 400;
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE, int AHEAD>
__global__ void __launch_bounds__(512) k(const unsigned char* __restrict__ src, size_t span, size_t wrap, float* out, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t off0 = ((size_t)blockIdx.x * 36864) % wrap;            // every workgroup walks the SAME `wrap` bytes, from its own start
  float acc = 0.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 0) {                                        // LDS-DMA: 36 pieces of 1 KiB per stage, waves 0-3 five, 4-7 four
    const int np = wave < 4 ? 5 : 4, p0 = wave < 4 ? wave * 5 : 20 + (wave - 4) * 4;
    auto issue = [&](int st, int step) {
      for (int j = 0; j < np; ++j) {
        const unsigned char* g = src + (off0 + (size_t)step * STAGE + (size_t)(p0 + j) * 1024) % wrap + lane * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(sm + st * STAGE + (p0 + j) * 1024), 16, 0, 0);
      }
    };
    for (int q = 0; q < AHEAD; ++q) issue(q, q);
    for (int s = 0; s < NSTEP; ++s) {
      if (AHEAD == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (AHEAD == 2) { if (wave < 4) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
      else { if (wave < 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
      asm volatile("s_barrier" ::: "memory");
      acc += *reinterpret_cast<const float*>(sm + (s % (AHEAD + 1)) * STAGE + tid * 16);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue((s + AHEAD) % (AHEAD + 1), s + AHEAD);
    }
  } else {                                                // registers: 4.5 x 16 bytes per thread and stage
    f4 r[AHEAD][5];
    auto load = [&](int slot, int step) {
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int idx = tid + 512 * j;                     // 16-byte unit of the stage (2304 of them)
        const unsigned char* g = src + (off0 + (size_t)step * STAGE + (size_t)(idx < 2304 ? idx : 2303) * 16) % wrap;
        r[slot][j] = *reinterpret_cast<const f4*>(g);
      }
    };
    auto store = [&](int slot, int st) {
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int idx = tid + 512 * j;
        if (idx < 2304) *reinterpret_cast<f4*>(sm + st * STAGE + idx * 16) = r[slot][j];
      }
    };
#pragma unroll
    for (int q = 0; q < AHEAD; ++q) load(q, q);
#pragma unroll 1
    for (int s0 = 0; s0 < NSTEP; s0 += AHEAD) {
#pragma unroll
      for (int q = 0; q < AHEAD; ++q) {
        const int s = s0 + q;
        store(q, s & 1);                                   // (waits for slot q's loads)
        load(q, s + AHEAD);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        acc += *reinterpret_cast<const float*>(sm + (s & 1) * STAGE + tid * 16);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 512 + tid] = acc;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int AHEAD>
void run(const char* name, const unsigned char* src, size_t span, size_t wrap, int grid = 256) {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8);
  const size_t lds = (size_t)(AHEAD + 1 > 2 ? AHEAD + 1 : 2) * STAGE;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, AHEAD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, AHEAD>), dim3(grid), dim3(512), lds, 0, src, span, wrap, out, cyc);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<MODE, AHEAD>), dim3(grid), dim3(512), lds, 0, src, span, wrap, out, cyc);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256);
  (void)hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < grid; ++i) s += (double)h[i];
  const double per = s / grid / NSTEP;
  printf("%-44s %6.0f cycles per 36 KB step = %5.1f B/clk/CU; kernel %.1f us -> %.2f TB/s chip-wide\n", name, per, STAGE / per, ms * 1e3,
         (double)grid * NSTEP * STAGE / (ms * 1e-3) / 1e12);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  const size_t span = 32u << 20;
  unsigned char* d; (void)hipMalloc(&d, span); (void)hipMemset(d, 1, span);
  for (size_t wrap : {(size_t)1 << 20, (size_t)3 << 20, (size_t)24 << 20}) {       // every workgroup's walk wraps within `wrap` bytes: 1 and 3 MB fit an XCD's L2, 24 MB the Infinity Cache
    printf("walk wraps every %zu MB\n", wrap >> 20);
    run<0, 1>("LDS-DMA, one step ahead", d, span, wrap);
    run<0, 2>("LDS-DMA, two steps ahead", d, span, wrap);
    run<0, 3>("LDS-DMA, three steps ahead", d, span, wrap);
    run<1, 1>("registers, one step ahead", d, span, wrap);
    run<1, 2>("registers, two steps ahead", d, span, wrap);
  }
  // few workgroups (a whole net's layers by ONE workgroup per row block would be 16-32 of them): what one CU can pull when its XCD's L2 is its own
  for (int grid : {64, 32, 16, 8}) {
    printf("grid %d, walk wraps every 1 MB\n", grid);
    run<0, 2>("LDS-DMA, two steps ahead", d, span, (size_t)1 << 20, grid);
    run<0, 3>("LDS-DMA, three steps ahead", d, span, (size_t)1 << 20, grid);
    run<1, 2>("registers, two steps ahead", d, span, (size_t)1 << 20, grid);
  }
  return 0;
}
