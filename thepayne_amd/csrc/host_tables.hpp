// host_tables.hpp -- theta-independent tables of the spectrum pipeline, built once
// per context on the host in fp64 (plain C++; shared by the HIP library and the CPU
// emulation).  They restate, for the static grids, what the reference recomputes on
// every call: resample_wave's pow-2 log grid and the two np.interp index/weight maps of
// the vsini stage (Payne/utils/smoothing.py:649-668, :300-312), ln(lambda) of the ANN
// and observed grids, polycalc's abscissa (Payne/fitting/fitutils.py:11-15) and the
// FFT twiddles.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "post_core.hpp"

namespace payne {

struct HostTables {
  int npix = 0, nobs = 0, n1 = 0, nmax = 0;
  std::vector<double> lnlam, lam, lnobs, xcheb;
  std::vector<c32> tw, twf;
  std::vector<int> rs1_idx, bk1_idx;
  std::vector<float> rs1_frac, bk1_frac, obs_f1, obs_ivar;
  std::vector<ObsRec> obs_rec;
  std::vector<double> obs_wave;     // the observed grid itself (LSF path: np.interp in wavelength)
  double vs_val = 0, obs_min = 0, obs_max = 0, geo_inv_dln = 0;
  bool has_flux = false;
  int geo = 0, rot_identity = 0;
  double ln0 = 0, dln = 0, ln_last = 0;
  std::vector<double> vs_tab;      // fp64 table sb(i*kVsTabStep)
  std::vector<float> vs_tab32;     // what the kernel interpolates
};

// copy the model-side scalars into the device-facing struct (pointers are set by the owner)
inline void fill_model_scalars(const HostTables& H, PostTables& T) {
  T.npix = H.npix; T.n1 = H.n1; T.nmax = H.nmax; T.vs_val = H.vs_val;
  T.geo_inv_dln = H.geo_inv_dln; T.geo = H.geo; T.ln0 = H.ln0; T.dln = H.dln; T.ln_last = H.ln_last;
  T.vs_tab_n = (int)H.vs_tab32.size() - 1;      // (without the leading mirror entry)
  T.twf_n = (int)H.twf.size();
  T.rot_identity = H.rot_identity;
  T.inv_lam0 = H.lam.empty() ? 0.f : (float)(1.0 / H.lam[0]);
  T.inv_dln32 = (float)H.geo_inv_dln;
  T.bk_r = H.npix > 1 ? (double)(H.n1 - 1) / (double)(H.npix - 1) : 1.0;
  const float hs = (float)(0.5 * H.dln / T.bk_r);                    // half a step of the stage's grid in ln(lambda)
  T.bk_c1 = 2.3283064365386963e-10f * (1.0f - hs); T.bk_c2 = 5.421010862427522e-20f * hs;
}

// numpy.linspace(start, stop, n)
inline void linspace(double start, double stop, int n, std::vector<double>& y) {
  y.resize(n);
  const double step = (stop - start) / (double)(n - 1);
  for (int j = 0; j < n; ++j) y[j] = (double)j * step + start;
  y[n - 1] = stop;
}

// index/weight of np.interp(x, xp, .) for one x; nan_outside mirrors left=right=NaN
inline void interp_map(double x, const std::vector<double>& xp, bool nan_outside, int& idx, float& frac) {
  const int n = (int)xp.size();
  if (nan_outside && (x < xp[0] || x > xp[n - 1])) { idx = -1; frac = 0.f; return; }
  if (x <= xp[0]) { idx = 0; frac = 0.f; return; }
  if (x >= xp[n - 1]) { idx = n - 2; frac = 1.f; return; }
  int k = (int)(std::upper_bound(xp.begin(), xp.end(), x) - xp.begin()) - 1;   // xp[k] <= x < xp[k+1]
  idx = k;
  frac = (float)((x - xp[k]) / (xp[k + 1] - xp[k]));
}

// What the post kernel's first convolution stage holds between its forward transform and the taper, for a real vector v of n points:
// Z[k] = sum_m (v[2m] + i v[2m+1]) exp(-2 pi i m k / (n/2)), k < n/2, written (re, im) interleaved -- n doubles.  The map v -> Z is
// linear and the same for every candidate, so a network whose output layer is linear can carry it in its weights (freq_rows below).
inline void packed_half_transform(const double* v, int n, double* out) {
  const int M = n / 2;
  std::vector<double> re(M), im(M);
  int lg = 0;
  while ((1 << lg) < M) ++lg;
  for (int m = 0; m < M; ++m) {
    int r = 0;
    for (int b = 0; b < lg; ++b) r |= ((m >> b) & 1) << (lg - 1 - b);
    re[r] = v[2 * m]; im[r] = v[2 * m + 1];
  }
  const double pi = 3.14159265358979323846264338327950288;
  for (int len = 2; len <= M; len <<= 1) {
    const int h = len / 2;
    for (int j = 0; j < h; ++j) {
      const double a = -2.0 * pi * (double)j / (double)len, wr = std::cos(a), wi = std::sin(a);
      for (int s = j; s < M; s += len) {
        const double xr = re[s + h] * wr - im[s + h] * wi, xi = re[s + h] * wi + im[s + h] * wr;
        re[s + h] = re[s] - xr; im[s + h] = im[s] - xi;
        re[s] += xr; im[s] += xi;
      }
    }
  }
  for (int k = 0; k < M; ++k) { out[2 * k] = re[k]; out[2 * k + 1] = im[k]; }
}
// ... in the order the post kernel reads it (post_core.hpp slots_issue): the conjugate pair (Z[j], Z[M - j]) side by side in slot j
// (four floats), j = 1 .. M/2 - 1; slot 0 = (Z[0], Z[M/2]).  M = n/2.
inline void pair_layout(const double* z, int n, double* out) {
  const int M = n / 2;
  out[0] = z[0]; out[1] = z[1]; out[2] = z[M]; out[3] = z[M + 1];
  for (int j = 1; j < M / 2; ++j) {
    out[4 * j] = z[2 * j]; out[4 * j + 1] = z[2 * j + 1];
    out[4 * j + 2] = z[2 * (M - j)]; out[4 * j + 3] = z[2 * (M - j) + 1];
  }
}
// ... and in the order the on-chip stage of a 65 536-point spectrum holds it (post_onchip.hpp chip_taper_pairs): virtual thread
// vt = 32 h + l (1024 of them) combines the pairs k = low + 1024 r, r < 16, low = h + 32 l; slot r * 1024 + vt = (Z[k], Z[M - k]),
// slot 0 = (Z[0], Z[M/2]).  Consecutive lanes read consecutive 16-byte slots.
inline void chip_layout(const double* z, int n, double* out) {
  const int M = n / 2;                                       // 32768
  for (int r = 0; r < 16; ++r)
    for (int vt = 0; vt < 1024; ++vt) {
      const int low = (vt >> 5) + 32 * (vt & 31), k = low + 1024 * r, kb = (k == 0) ? M / 2 : M - k;
      double* o = out + 4 * ((size_t)r * 1024 + vt);
      o[0] = z[2 * k]; o[1] = z[2 * k + 1]; o[2] = z[2 * kb]; o[3] = z[2 * kb + 1];
    }
}
// ... and of a 32 768-point spectrum of the two-candidates-at-a-time stage (post_onchip2.hpp chip2_taper_pairs): after its third
// radix stage the virtual thread (h, l) of candidate c = bit 2 of l holds k = low + 512 r, low = h + 32 k2a, k2a = 4 (l >> 3) + (l & 3);
// slot r * 512 + 16 h + k2a of the candidate's row = (Z[k], Z[M - k]), slot 0 = (Z[0], Z[M/2]).
inline void chip2_layout(const double* z, int n, double* out) {
  const int M = n / 2;                                       // 16384
  for (int r = 0; r < 16; ++r)
    for (int h = 0; h < 32; ++h)
      for (int k2a = 0; k2a < 16; ++k2a) {
        const int low = h + 32 * k2a, k = low + 512 * r, kb = (k == 0) ? M / 2 : M - k;
        double* o = out + 4 * ((size_t)r * 512 + 16 * h + k2a);
        o[0] = z[2 * k]; o[1] = z[2 * k + 1]; o[2] = z[2 * kb]; o[3] = z[2 * kb + 1];
      }
}
// The output layer of a network (W [npix][K] row-major, bias [npix], spectrum = W a + bias + shift) restated for rows handed over as
// the half transform of the spectrum: layout 0 pair_layout, 1 chip_layout (n = 65536), 2 chip2_layout (n = 32768).  Wz [n][K], bz [n].
// rs_idx / rs_frac (n entries; null: npix == n, the spectrum itself): the rotation stage's static resampling onto its power-of-two
// log grid (resample_wave, smoothing.py:649-668: point j = pixel rs_idx[j] + rs_frac[j] towards the next) is LINEAR as well and is
// folded in: the rows are then the transform of the RESAMPLED spectrum -- what the rotation stage transforms -- for model grids of
// any length.
inline void freq_rows(const float* W, const float* bias, double shift, int n, int K, std::vector<float>& Wz, std::vector<float>& bz,
                      int layout = 0, const int* rs_idx = nullptr, const float* rs_frac = nullptr, int npix = 0) {
  if (!rs_idx) npix = n;
  Wz.assign((size_t)n * K, 0.f); bz.assign((size_t)n, 0.f);
  std::vector<double> pix(npix), v(n), z(n), zn(n);
  for (int h = 0; h <= K; ++h) {
    for (int i = 0; i < npix; ++i) pix[i] = h < K ? (double)W[(size_t)i * K + h] : (double)bias[i] + shift;
    if (rs_idx) for (int j = 0; j < n; ++j) { const int k = rs_idx[j]; v[j] = pix[k] + (pix[k + 1] - pix[k]) * (double)rs_frac[j]; }
    else v = pix;
    packed_half_transform(v.data(), n, zn.data());
    if (layout == 1) chip_layout(zn.data(), n, z.data());
    else if (layout == 2) chip2_layout(zn.data(), n, z.data());
    else pair_layout(zn.data(), n, z.data());
    if (h < K) for (int i = 0; i < n; ++i) Wz[(size_t)i * K + h] = (float)z[i];
    else for (int i = 0; i < n; ++i) bz[i] = (float)z[i];
  }
}

// returns 0 on success, <0 if the wavelength grid is unusable
inline int build_model_tables(const double* wave, int npix, HostTables& H) {
  if (npix < 16) return -1;
  for (int i = 1; i < npix; ++i)
    if (!(wave[i] > wave[i - 1])) return -2;             // must be strictly increasing
  H.npix = npix;
  H.n1 = pow2ceil(npix);
  H.nmax = H.n1;
  H.lam.assign(wave, wave + npix);
  H.lnlam.resize(npix);
  for (int i = 0; i < npix; ++i) H.lnlam[i] = std::log(wave[i]);
  H.geo_inv_dln = (double)(npix - 1) / (H.lnlam[npix - 1] - H.lnlam[0]);
  H.ln0 = H.lnlam[0]; H.ln_last = H.lnlam[npix - 1];
  H.dln = (H.ln_last - H.ln0) / (double)(npix - 1);
  double dev = 0.0;
  for (int i = 0; i < npix; ++i) dev = std::max(dev, std::fabs(H.lnlam[i] - (H.ln0 + (double)i * H.dln)));
  H.geo = (dev < 1e-12) ? 1 : 0;            // < 3e-7 of a pixel even at R ~ 1e5
  // vsini taper table sb(i*h), i = 0 .. kVsTabMax/h + 2
  const int nt = (int)(kVsTabMax / kVsTabStep) + 3;
  H.vs_tab.resize(nt);
  H.vs_tab[0] = 1.0;
  for (int i = 1; i < nt; ++i) {
    const double u = (double)i * kVsTabStep;
    // below u ~ 0.3 the closed form cancels catastrophically even in fp64 (terms ~ 1/u^2):
    // use the even power series sb = sum_n c_n u^(2n), c_n from J1(u)/u and 3(sin u - u cos u)/(2u^3)
    if (u < 0.5) {
      const double z = u * u;
      // J1(u)/u = 1/2 - z/16 + z^2/384 - z^3/18432 + z^4/1474560 - ...
      // 3(sin u - u cos u)/(2 u^3) = 1/2 - z/20 + z^2/560 - z^3/30240 + z^4/2661120 - ...
      const double a = 0.5 + z * (-1.0 / 16 + z * (1.0 / 384 + z * (-1.0 / 18432 + z * (1.0 / 1474560 + z * (-1.0 / 176947200)))));
      const double b = 0.5 + z * (-1.0 / 20 + z * (1.0 / 560 + z * (-1.0 / 30240 + z * (1.0 / 2661120 + z * (-1.0 / 345945600)))));
      H.vs_tab[i] = a + b;
    } else {
      H.vs_tab[i] = vsini_sb_exact(u);
    }
  }
  // what the kernel interpolates: fp32, with ONE LEADING entry sb(-h) = sb(h) so that the four values around any position are
  // contiguous (PostTables::vs_tab points at the entry of u = 0, i.e. at vs_tab32[1])
  H.vs_tab32.assign(1, (float)H.vs_tab[1]);
  H.vs_tab32.insert(H.vs_tab32.end(), H.vs_tab.begin(), H.vs_tab.end());
  // vsini grid: w = exp(linspace(ln wmin, ln wmax, n1))
  std::vector<double> lnw, w(H.n1);
  linspace(std::log(wave[0]), std::log(wave[npix - 1]), H.n1, lnw);
  for (int j = 0; j < H.n1; ++j) w[j] = std::exp(lnw[j]);
  H.rs1_idx.resize(H.n1); H.rs1_frac.resize(H.n1);
  for (int j = 0; j < H.n1; ++j) interp_map(w[j], H.lam, false, H.rs1_idx[j], H.rs1_frac[j]);
  H.bk1_idx.resize(npix); H.bk1_frac.resize(npix);
  for (int i = 0; i < npix; ++i) interp_map(wave[i], w, true, H.bk1_idx[i], H.bk1_frac[i]);
  // identity maps?  (geometric grid, npix a power of two: the resampled grid IS the ANN grid up
  // to fp64 rounding; weights within 1e-8 of 0/1 change nothing in fp32 -- a flux moves by less than 1e-8 |b - a|, a sixth
  // of an fp32 ulp of 1 -- and the rounding grows with the grid: 3.6e-10 at 4096 pixels, 1.04e-9 at 65 536, where a bound of
  // 1e-9 sent every candidate through two gather passes over the spectrum, a fifth of the streaming kernel's time)
  H.rot_identity = (H.n1 == npix) ? 1 : 0;
  for (int j = 0; j < H.n1 && H.rot_identity; ++j) {
    const int src = H.rs1_idx[j] + (H.rs1_frac[j] > 0.5f ? 1 : 0);
    const float off = H.rs1_frac[j] > 0.5f ? 1.f - H.rs1_frac[j] : H.rs1_frac[j];
    if (src != j || off > 1e-8f) H.rot_identity = 0;
  }
  for (int i = 1; i + 1 < npix && H.rot_identity; ++i) {      // end points are overwritten afterwards
    if (H.bk1_idx[i] < 0) { H.rot_identity = 0; break; }
    const int src = H.bk1_idx[i] + (H.bk1_frac[i] > 0.5f ? 1 : 0);
    const float off = H.bk1_frac[i] > 0.5f ? 1.f - H.bk1_frac[i] : H.bk1_frac[i];
    if (src != i || off > 1e-8f) H.rot_identity = 0;
  }
  // dv = ckms * median(diff(log(w)))   (smoothing.py:306-307)
  std::vector<double> d(H.n1 - 1);
  for (int j = 0; j + 1 < H.n1; ++j) d[j] = std::log(w[j + 1]) - std::log(w[j]);
  std::sort(d.begin(), d.end());
  const int m = (int)d.size();
  const double med = (m & 1) ? d[m / 2] : 0.5 * (d[m / 2 - 1] + d[m / 2]);
  H.vs_val = 1.0 / ((double)H.n1 * (kCkms * med));
  H.tw.resize(H.nmax);
  for (int j = 0; j < H.nmax; ++j) {
    const double a = 2.0 * kPi * (double)j / (double)H.nmax;
    H.tw[j] = {(float)std::cos(a), (float)(-std::sin(a))};
  }
  // pass-ordered twiddles of the fixed-geometry (n1/2)-point FFT + the real-FFT split factors
  {
    const int M = H.n1 / 2;
    H.twf.clear();
    for (int P = 1; P < M; P *= plan_radix(M, P)) {
      const int R = plan_radix(M, P);
      if (P == 1) continue;
      for (int j = 0; j < plan_tw_rows(R); ++j) {            // the powers that are read: 1, 2, 4 (radix 8) or 1 (post_core.hpp, plan)
        const int r = 1 << j;
        for (int k = 0; k < P; ++k) {
          const double a = 2.0 * kPi * (double)k * (double)r / ((double)P * (double)R);
          H.twf.push_back({(float)std::cos(a), (float)(-std::sin(a))});
        }
      }
    }
    for (int k = 0; k < M / 2; ++k) {
      const double a = 2.0 * kPi * (double)k / (double)(2 * M);
      H.twf.push_back({(float)std::cos(a), (float)(-std::sin(a))});
    }
  }
  return 0;
}

inline int build_obs_tables(const double* wave, const double* flux, const double* eflux, int nobs, HostTables& H) {
  H.nobs = nobs;
  H.lnobs.resize(nobs); H.xcheb.resize(nobs);
  if (nobs <= 0) return 0;
  double mn = wave[0], mx = wave[0];
  for (int i = 0; i < nobs; ++i) { mn = std::min(mn, wave[i]); mx = std::max(mx, wave[i]); }
  H.obs_min = mn; H.obs_max = mx;
  double xmax = 0.0;
  for (int i = 0; i < nobs; ++i) xmax = std::max(xmax, wave[i] - mn);
  for (int i = 0; i < nobs; ++i) {
    H.lnobs[i] = std::log(wave[i]);
    H.xcheb[i] = 2.0 * ((wave[i] - mn) / xmax) - 1.0;   // fitutils.py:13-14
  }
  H.has_flux = (flux != nullptr) && (eflux != nullptr);
  if (H.has_flux) {
    H.obs_f1.resize(nobs); H.obs_ivar.resize(nobs);
    for (int i = 0; i < nobs; ++i) {
      H.obs_f1[i] = (float)(flux[i] - 1.0);
      H.obs_ivar[i] = (float)(1.0 / (eflux[i] * eflux[i]));
    }
  }
  H.obs_wave.assign(wave, wave + nobs);
  // padded to a multiple of kObsPad records with copies of the last pixel that weigh nothing (1/sigma^2 = 0): the likelihood-only
  // loop (post_core.hpp obs_loop_fast) then runs whole blocks of pixels without an index clamp or a validity test per pixel
  H.obs_rec.resize(((size_t)nobs + kObsPad - 1) / kObsPad * kObsPad);
  for (int i = 0; i < nobs; ++i)
    H.obs_rec[i] = ObsRec{H.lnobs[i], H.has_flux ? H.obs_f1[i] : 0.f, H.has_flux ? H.obs_ivar[i] : 0.f};
  for (size_t i = (size_t)nobs; i < H.obs_rec.size(); ++i) H.obs_rec[i] = ObsRec{H.lnobs[nobs - 1], 0.f, 0.f};
  return 0;
}

}  // namespace payne
