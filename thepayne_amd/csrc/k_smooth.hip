// k_smooth.hip -- PayneSpecPredict.smoothspec's branches that are NOT on the sampler's path (Payne/utils/smoothing.py):
//   smooth_vel (:171-210) and smooth_wave (:339-393): direct Gaussian quadrature per output pixel, trapezoid rule over the
//   input pixels within nsigma;  smooth_lsf (:435-480): the dense (nout x nin) kernel product;  smooth_wave_fft (:395-433):
//   linear pow-2 resampling, Gaussian FFT convolution in wavelength, clamped np.interp.
// The reference calls them "insanely slow, but general and correct": one thread per output pixel in fp64 here (O(nout nin),
// milliseconds), the FFT branch through the post kernel's own transform.  Caller-supplied HOST arrays in, host array out;
// the call is synchronous (this is an analysis helper of the public class, nothing the likelihood calls).
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "../../include/payne_hip.h"
#include "post_seq.hpp"

using namespace payne;

#include "post_kernels.hpp"

namespace {

struct DirectArgs {
  int kind;                       // PAYNE_SMOOTH_*
  const double* wave; const double* spec; int n;
  const double* outwave; int nout;
  const double* sigma; int nsig;  // scalar or vector
  double inres; int in_vel; double nsigma;
  double* out;
};

// sigma_eff of input pixel j (smooth_vel: constant, in ln-lambda units; smooth_wave: Angstrom, possibly per pixel)
__device__ __forceinline__ double sigma_eff_at(const DirectArgs& a, int j) {
  const double s = a.sigma[a.nsig > 1 ? j : 0];
  if (a.kind == PAYNE_SMOOTH_VEL_DIRECT) return sqrt(s * s - a.inres * a.inres) / kCkms;      // smoothing.py:190-195
  double sq;
  if (a.inres <= 0.0) sq = s * s;                                                             // :372-380
  else if (a.in_vel) { const double q = a.wave[j] / a.inres; sq = s * s - q * q; }
  else sq = s * s - a.inres * a.inres;
  return sqrt(sq);
}

// smooth_vel / smooth_wave: flux_i = trapz(f s, x) / trapz(f, x) over the pixels with |x| < nsigma, x in input order
__global__ void payne_smooth_quad_kernel(DirectArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.nout) return;
  const double w = a.outwave[i];
  const double lw = log(w);
  double num = 0.0, den = 0.0, xp = 0.0, fp = 0.0, sp = 0.0;
  bool have = false;
  for (int j = 0; j < a.n; ++j) {
    const double se = sigma_eff_at(a, j);
    const double x = (a.kind == PAYNE_SMOOTH_VEL_DIRECT) ? (lw - log(a.wave[j])) / se : (a.wave[j] - w) / se;
    if (a.nsigma > 0.0 && !(fabs(x) < a.nsigma)) continue;          // `good = np.abs(x) < nsigma`
    const double f = exp(-0.5 * (x * x));
    const double s = a.spec[j];
    if (have) {                                                     // np.trapz over the selected points, in order
      const double dx = x - xp;
      num += dx * (f * s + fp * sp) / 2.0;
      den += dx * (f + fp) / 2.0;
    }
    xp = x; fp = f; sp = s; have = true;
  }
  a.out[i] = num / den;
}

// smooth_lsf: kernel_ij = exp(-(w_i - w_j)^2 / 2 sigma_i^2) dw_j / (sigma_i sqrt(2 pi)), rows normalised to one
__global__ void payne_smooth_lsf_dense_kernel(DirectArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.nout) return;
  const double w = a.outwave[i], sg = a.sigma[a.nsig > 1 ? i : 0];
  const double pre = 1.0 / (sg * sqrt(kPi * 2.0));
  double num = 0.0, den = 0.0;
  for (int j = 0; j < a.n; ++j) {
    const double dw = (j == 0) ? (a.wave[1] - a.wave[0]) : ((j == a.n - 1) ? (a.wave[a.n - 1] - a.wave[a.n - 2])
                                                                            : (a.wave[j + 1] - a.wave[j - 1]) / 2.0);   // np.gradient
    const double d = w - a.wave[j];
    const double k = pre * exp(-(d * d) / (2.0 * (sg * sg))) * dw;
    num += k * a.spec[j];
    den += k;
  }
  a.out[i] = num / den;
}

// np.interp(x, xp, fp) without left / right: clamped
__device__ double interp_clamped(double x, const double* xp, const float* fp, int n, float base) {
  if (x >= xp[n - 1]) return (double)fp[n - 1] + base;
  if (x <= xp[0]) return (double)fp[0] + base;
  int lo = 0, hi = n;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (xp[mid] <= x) lo = mid; else hi = mid; }
  const double f0 = fp[lo], f1 = fp[lo + 1];
  return (f1 - f0) / (xp[lo + 1] - xp[lo]) * (x - xp[lo]) + f0 + base;
}

// smooth_wave_fft: one workgroup; buffers in global memory (wg[n2] doubles, bufA / bufB floats, tw[n2] complex)
struct WaveFftArgs {
  const double* wave; const double* spec; int n;
  const double* outwave; int nout;
  double sigma_out, inres;
  int n2;
  double* wg; float* bufA; float* bufB; c32* tw;
  double* out;
};
__global__ void __launch_bounds__(256) payne_smooth_wave_fft_kernel(WaveFftArgs a) {
  __shared__ double red_dummy[8];
  (void)red_dummy;
  const int tid = threadIdx.x, n2 = a.n2;
  // resample_wave(linear=True): w = np.linspace(wmin, wmax, n2); s = np.interp(w, wave, spec)   (smoothing.py:649-668)
  const double wmin = a.wave[0], wmax = a.wave[a.n - 1];
  const double step = (wmax - wmin) / (double)(n2 - 1);
  for (int k = tid; k < n2; k += 256) {
    const double w = (k == n2 - 1) ? wmax : __dadd_rn(__dmul_rn((double)k, step), wmin);
    a.wg[k] = w;
    double v;
    if (w >= wmax) v = a.spec[a.n - 1];
    else if (w <= wmin) v = a.spec[0];
    else {
      int lo = 0, hi = a.n;
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.wave[mid] <= w) lo = mid; else hi = mid; }
      v = (a.spec[lo + 1] - a.spec[lo]) / (a.wave[lo + 1] - a.wave[lo]) * (w - a.wave[lo]) + a.spec[lo];
    }
    a.bufA[k] = (float)(v - 1.0);                          // shifted flux: every step below has unit DC gain
  }
  for (int j = tid; j < n2; j += 256) {                    // exp(-2 pi i j / n2)
    double s, c;
    sincos(-2.0 * kPi * (double)j / (double)n2, &s, &c);
    a.tw[j] = {(float)c, (float)s};
  }
  __syncthreads();
  // smooth_fft(dw, spec, sigma): taper exp(-2 pi^2 sigma^2 ss^2), ss = k / (n2 dw)   (smoothing.py:588-608)
  const double sig = sqrt(a.sigma_out * a.sigma_out - a.inres * a.inres);
  const double val = 1.0 / ((double)n2 * step);
  PostTables T{};
  T.tw = a.tw; T.nmax = n2; T.n1 = n2;
  TaperArgs ta{};
  ta.g_c2 = (float)(-2.0 * (kPi * kPi) * (sig * sig) * (val * val) * 1.4426950408889634);
  DevExecT<false, false> ex;
  bool no_edge = false;
  const float* conv = conv_stage<0, 256, false>(ex, T, a.tw, a.bufA, a.bufB, n2, ta, no_edge);
  for (int i = tid; i < a.nout; i += 256) a.out[i] = interp_clamped(a.outwave[i], a.wg, conv, n2, 1.0f);
}

__global__ void payne_interp_clamped_kernel(const double* wave, const double* spec, int n, const double* outwave, int nout, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nout) return;
  const double x = outwave[i];
  if (x >= wave[n - 1]) { out[i] = spec[n - 1]; return; }
  if (x <= wave[0]) { out[i] = spec[0]; return; }
  int lo = 0, hi = n;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (wave[mid] <= x) lo = mid; else hi = mid; }
  out[i] = (spec[lo + 1] - spec[lo]) / (wave[lo + 1] - wave[lo]) * (x - wave[lo]) + spec[lo];
}

struct DevBuf {
  std::vector<void*> p;
  ~DevBuf() { for (void* q : p) (void)hipFree(q); }
  template <class V> V* put(const V* host, size_t n) {
    void* d = nullptr;
    if (hipMalloc(&d, (n ? n : 1) * sizeof(V)) != hipSuccess) return nullptr;
    p.push_back(d);
    if (host && n && hipMemcpy(d, host, n * sizeof(V), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return reinterpret_cast<V*>(d);
  }
};

}  // namespace

extern "C" int payne_smooth_direct(int device, int kind, const double* wave, const double* spec, int n, const double* outwave,
                                   int nout, const double* sigma, int nsig, double inres, int in_vel, double nsigma, double* out) {
  if (!wave || !spec || !outwave || !out || n < 2 || nout < 1) return PAYNE_E_INVALID;
  if (kind < PAYNE_SMOOTH_VEL_DIRECT || kind > PAYNE_SMOOTH_INTERP) return PAYNE_E_INVALID;
  if (kind != PAYNE_SMOOTH_INTERP && (!sigma || nsig < 1)) return PAYNE_E_INVALID;
  if (kind == PAYNE_SMOOTH_WAVE_DIRECT && nsig > 1 && nsig != n) return PAYNE_E_INVALID;
  if (kind == PAYNE_SMOOTH_LSF_DIRECT && nsig > 1 && nsig != nout) return PAYNE_E_INVALID;
  if ((kind == PAYNE_SMOOTH_VEL_DIRECT || kind == PAYNE_SMOOTH_WAVE_FFT) && nsig != 1) return PAYNE_E_INVALID;
  if (kind == PAYNE_SMOOTH_WAVE_DIRECT) {                    // "Desired wavelength sigma is lower than the value possible" (:381-383)
    for (int j = 0; j < (nsig > 1 ? n : 1); ++j) {
      const double s = sigma[j];
      const double sq = inres <= 0.0 ? s * s : (in_vel ? s * s - (wave[j] / inres) * (wave[j] / inres) : s * s - inres * inres);
      if (sq < 0.0) return PAYNE_E_SIGMA;
    }
  }
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (hipSetDevice(device) != hipSuccess) return PAYNE_E_HIP;
  int rc = PAYNE_OK;
  {
    DevBuf B;
    const double* dw = B.put(wave, (size_t)n);
    const double* ds = B.put(spec, (size_t)n);
    const double* dow = B.put(outwave, (size_t)nout);
    const double* dsig = sigma ? B.put(sigma, (size_t)nsig) : nullptr;
    double* dout = B.put<double>(nullptr, (size_t)nout);
    if (!dw || !ds || !dow || !dout || (sigma && !dsig)) rc = PAYNE_E_HIP;
    if (!rc) {
      const dim3 grid((nout + 127) / 128), block(128);
      if (kind == PAYNE_SMOOTH_INTERP) {
        hipLaunchKernelGGL(payne_interp_clamped_kernel, grid, block, 0, 0, dw, ds, n, dow, nout, dout);
      } else if (kind == PAYNE_SMOOTH_WAVE_FFT) {
        int n2 = 1;
        while (n2 < n) n2 <<= 1;
        if (n2 < 16) rc = PAYNE_E_UNSUPPORTED;
        WaveFftArgs a{dw, ds, n, dow, nout, sigma[0], inres, n2, nullptr, nullptr, nullptr, nullptr, dout};
        a.wg = B.put<double>(nullptr, (size_t)n2);
        a.bufA = B.put<float>(nullptr, (size_t)n2 + n2 / 4);
        a.bufB = B.put<float>(nullptr, (size_t)n2 + n2 / 4);
        a.tw = B.put<c32>(nullptr, (size_t)n2);
        if (!a.wg || !a.bufA || !a.bufB || !a.tw) rc = PAYNE_E_HIP;
        if (!rc) hipLaunchKernelGGL(payne_smooth_wave_fft_kernel, dim3(1), dim3(256), 0, 0, a);
      } else {
        DirectArgs a{kind, dw, ds, n, dow, nout, dsig, nsig, inres, in_vel, nsigma, dout};
        if (kind == PAYNE_SMOOTH_LSF_DIRECT) hipLaunchKernelGGL(payne_smooth_lsf_dense_kernel, grid, block, 0, 0, a);
        else hipLaunchKernelGGL(payne_smooth_quad_kernel, grid, block, 0, 0, a);
      }
      if (!rc && hipGetLastError() != hipSuccess) rc = PAYNE_E_HIP;
      if (!rc && hipDeviceSynchronize() != hipSuccess) rc = PAYNE_E_HIP;
      if (!rc && hipMemcpy(out, dout, (size_t)nout * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = PAYNE_E_HIP;
    }
  }
  (void)hipSetDevice(prev);
  return rc;
}
