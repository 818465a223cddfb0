// select.hpp -- NaN-ignoring median of an arbitrarily long row by one workgroup, without sorting it: most-significant-
// digit radix selection on the order-preserving integer image of the doubles (8 passes of 8 bits; each pass one
// histogram sweep over the row).  Used where np.nanmedian appears on the path with no size limit in the reference: the
// continuum normalisation (Payne/predict/ystpred.py:199-201) and the LSF grid (Payne/utils/smoothing.py:528-531).
// The rows short enough for LDS keep their bitonic sort; this is the form for everything longer.
#pragma once
#ifdef __HIPCC__

namespace payne {

// order-preserving map double -> uint64 (NaN excluded by the caller)
__device__ __forceinline__ unsigned long long key_of(double x) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double value_of(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)b);
}

// The element of rank `rank` (0-based, ascending) among the non-NaN values F(0..n-1).  `hist` = 256 ints of LDS,
// `bcast` = 2 unsigned long long of LDS.  Every thread of the (NT-thread) workgroup must call it; all get the result.
template <int NT, class F>
__device__ double select_rank(F&& value, int n, long long rank, int* hist, unsigned long long* bcast) {
  const int tid = threadIdx.x;
  unsigned long long prefix = 0ull;                     // the digits chosen so far, in their place
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = 56 - 8 * pass;
    const unsigned long long himask = pass == 0 ? 0ull : (~0ull << (shift + 8));
    for (int b = tid; b < 256; b += NT) hist[b] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
      const double x = value(i);
      if (x == x) {
        const unsigned long long k = key_of(x);
        if ((k & himask) == prefix) atomicAdd(&hist[(int)((k >> shift) & 0xFFull)], 1);
      }
    }
    __syncthreads();
    if (tid == 0) {
      long long r = rank;
      int b = 0;
      for (; b < 255; ++b) { const int cnt = hist[b]; if (r < cnt) break; r -= cnt; }
      bcast[0] = prefix | ((unsigned long long)b << shift);
      bcast[1] = (unsigned long long)r;
    }
    __syncthreads();
    prefix = bcast[0];
    rank = (long long)bcast[1];
    __syncthreads();
  }
  return value_of(prefix);
}

// np.nanmedian of F(0..n-1): NaN when every value is NaN; the mean of the two middle values for an even count.
template <int NT, class F>
__device__ double nanmedian_select(F&& value, int n, int* hist, unsigned long long* bcast, int* count_lds) {
  const int tid = threadIdx.x;
  if (tid == 0) *count_lds = 0;
  __syncthreads();
  int nv = 0;
  for (int i = tid; i < n; i += NT) { const double x = value(i); nv += (x == x) ? 1 : 0; }
  atomicAdd(count_lds, nv);
  __syncthreads();
  const int m = *count_lds;
  __syncthreads();
  if (m == 0) return __builtin_nan("");
  const double hi = select_rank<NT>(value, n, m >> 1, hist, bcast);
  if (m & 1) return hi;
  const double lo = select_rank<NT>(value, n, (m >> 1) - 1, hist, bcast);
  return 0.5 * (lo + hi);
}

}  // namespace payne
#endif
