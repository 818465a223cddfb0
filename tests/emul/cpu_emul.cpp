// cpu_emul.cpp -- host execution of the SAME phase code the gfx950 post_kernel runs
// (thepayne_amd/csrc/post_core.hpp + post_seq.hpp), with the workgroup's threads
// stepped serially between barriers.  Test infrastructure: lets the CPU suite (and
// -fsanitize builds) check the pipeline against the oracle without a GPU.  It is not
// part of the product and nothing in thepayne_amd/ loads it.
#include <cstring>
#include <vector>

#include "../../thepayne_amd/csrc/host_tables.hpp"
#include "../../thepayne_amd/csrc/post_seq.hpp"

using namespace payne;

static int g_fast_windows = 0;   // candidates whose R-stage window came from the setup-time probe
extern "C" int payne_emul_fast_windows() { return g_fast_windows; }

struct HostExec {
  static constexpr bool kTwLds = false;
  int nthr;
  template <class F> void par(F&& f) { for (int t = 0; t < nthr; ++t) f(t, nthr); }
  int nthreads() const { return nthr; }
  void mark(int) {}
  template <class F> void single(F&& f) { f(nthr); }
  c32* tile_ = nullptr;                                    // scratch of the four-step transform (null: plain passes)
  c32* tile() const { return tile_; }
  static c32* buf(c32* p) { return p; }                    // address-space hooks of the device executor
  static const c32* twid(const c32* p) { return p; }
  static c32* lds(c32* p) { return p; }
};

// the same instantiation choice the GPU launcher makes (compile-time FFT geometry needs 256 threads)
template <int LOG2N>
static void run_one(HostExec& ex, const PostTables& T, const double* th, double factor, const float* raw, float* a,
                    float* b, CandState& S, double* red, float* out, int stage, double* x2, const CandState* prep) {
  run_candidate<LOG2N, kPostThreads>(ex, T, T.twf, th, factor, raw, a, b, S, red, out, stage, x2, prep);
}

// The output layer restated for rows in the frequency domain (host_tables.hpp freq_rows: what payne_ctx_create uploads), for the CPU
// test that checks it against numpy's FFT.  layout: 0 pair_layout, 1 chip_layout (n = 65536), 2 chip2_layout (n = 32768).
extern "C" int payne_emul_freq_rows(const float* W, const float* bias, double shift, int n, int K, int layout, float* Wz, float* bz) {
  std::vector<float> wz, b;
  freq_rows(W, bias, shift, n, K, wz, b, layout);
  std::memcpy(Wz, wz.data(), wz.size() * sizeof(float));
  std::memcpy(bz, b.data(), b.size() * sizeof(float));
  return 0;
}

extern "C" int payne_emul_post(const double* wave, int npix, double r_ann, const double* obs_wave,
                               const double* obs_flux, const double* obs_eflux, int nobs, int npoly,
                               const double* theta, int ncols, int B, double instr_factor,
                               const float* raw_m1, int out_stage, float* out, int ld_out,
                               double* chi2, int* info, int nthreads, int force_general, int use_prep) {
  HostTables H;
  int rc = build_model_tables(wave, npix, H);
  if (rc) return rc;
  build_obs_tables(obs_wave, obs_flux, obs_eflux, nobs, H);
  PostTables T;
  std::memset(&T, 0, sizeof(T));
  fill_model_scalars(H, T);
  T.nobs = nobs; T.vs_tab = H.vs_tab32.data() + 1;
  T.lnlam = H.lnlam.data(); T.lam = H.lam.data(); T.tw = H.tw.data(); T.twf = H.twf.data();
  T.rs1_idx = H.rs1_idx.data(); T.rs1_frac = H.rs1_frac.data();
  T.bk1_idx = H.bk1_idx.data(); T.bk1_frac = H.bk1_frac.data();
  T.lnobs = H.lnobs.data(); T.xcheb = H.xcheb.data(); T.obs_rec = H.obs_rec.data();
  T.obs_f1 = H.has_flux ? H.obs_f1.data() : nullptr;
  T.obs_ivar = H.has_flux ? H.obs_ivar.data() : nullptr;
  T.obs_min = H.obs_min; T.obs_max = H.obs_max; T.r_ann = r_ann;
  T.ln_obs_min = std::log(H.obs_min); T.ln_obs_max = std::log(H.obs_max);
  T.npoly = npoly;
  if (force_general == 1 || force_general == 2) { T.geo = 0; T.rot_identity = 0; }    // (3: the four-step transform on the geometric grid)
  // 4: the rows handed over the way the restated output layer writes them (freq_rows): their half transform in pair layout,
  // rounded to fp32
  std::vector<float> zrows;
  if (force_general == 4) {
    if (!T.rot_identity || H.n1 != npix) return -78;
    zrows.resize((size_t)B * npix);
    std::vector<double> v(npix), z(npix), zp(npix);
    for (int c = 0; c < B; ++c) {
      for (int i = 0; i < npix; ++i) v[i] = (double)raw_m1[(size_t)c * npix + i];
      packed_half_transform(v.data(), npix, z.data());
      pair_layout(z.data(), npix, zp.data());
      for (int i = 0; i < npix; ++i) zrows[(size_t)c * npix + i] = (float)zp[i];
    }
    raw_m1 = zrows.data();
    T.raw_freq = 1;
    force_general = 0;
  }
  HostExec ex{nthreads};
  std::vector<c32> tilebuf;
  if (force_general >= 2) {                                // + the four-step transform of the global-workspace kernel
    tilebuf.resize(2 * (size_t)fft_tile_complex());
    ex.tile_ = tilebuf.data();
  }
  std::vector<float> a(fft_buf_floats(H.n1)), b(fft_buf_floats(H.n1));
  std::vector<double> red(scratch_doubles(nthreads));
  const bool fixed = (nthreads == kPostThreads) && !force_general;
  for (int c = 0; c < B; ++c) {
    CandState S;
    std::memset(&S, 0, sizeof(S));
    double x2 = 0.0;
    const double* th = theta + (size_t)c * ncols;
    const float* rw = raw_m1 + (size_t)c * npix;
    float* o = out ? out + (size_t)c * ld_out : nullptr;
    // the record the first dense launch would have written (prep_candidate), when asked for
    CandState P;
    const CandState* prep = nullptr;
    if (use_prep) { prep_candidate(T, th, instr_factor, P); prep = &P; }
    if (fixed && H.n1 == 4096) run_one<12>(ex, T, th, instr_factor, rw, a.data(), b.data(), S, red.data(), o, out_stage, &x2, prep);
    else if (fixed && H.n1 == 2048) run_one<11>(ex, T, th, instr_factor, rw, a.data(), b.data(), S, red.data(), o, out_stage, &x2, prep);
    else if (fixed && H.n1 == 1024) run_one<10>(ex, T, th, instr_factor, rw, a.data(), b.data(), S, red.data(), o, out_stage, &x2, prep);
    else if (fixed && H.n1 == 8192) run_one<13>(ex, T, th, instr_factor, rw, a.data(), b.data(), S, red.data(), o, out_stage, &x2, prep);
    else run_candidate<0, kPostThreads>(ex, T, T.twf, th, instr_factor, rw, a.data(), b.data(), S, red.data(), o, out_stage, &x2, prep);
    if (chi2) chi2[c] = x2;
    // the setup-time probe must agree with the full mask count (phase_mask_count over every thread)
    int full_below = -1, full_notabove = -1;
    if (S.do_smooth && out_stage != 0 && out_stage != 1) {
      std::vector<int> cnt(nthreads);
      for (int t = 0; t < nthreads; ++t) phase_mask_count(t, nthreads, T, S, cnt.data());
      full_below = full_notabove = 0;
      for (int s = 0; s < n_slots(nthreads); ++s) { full_below += cnt[s] & 0xffff; full_notabove += cnt[s] >> 16; }
      if (S.win_ready) ++g_fast_windows;
      if (S.win_ready && (S.win_below != full_below || S.win_notabove != full_notabove)) return -77;
    }
    if (info) {   // mask first / count / FFT length / whether the probe path was taken
      info[3 * c] = info[3 * c + 1] = info[3 * c + 2] = -1;
      if (full_below >= 0) {
        const Window W = window_from_counts(T, S.dop, S.g_a, full_below, full_notabove);
        info[3 * c] = full_below; info[3 * c + 1] = full_notabove - full_below; info[3 * c + 2] = W.n2;
      }
    }
  }
  return 0;
}


// prep_candidate's mask counts against the table, candidate by candidate (tests/test_emul_pipeline.py): the counts by arithmetic on
// geometric grids (and the table path for the few whose position sits within 1e-5 of a pixel boundary) must be THE counts of the
// brute-force comparison lam[i] (1 + rv/c) <= wl / < wh over every pixel.  Returns the number of mismatches; *n_arith: how many took
// the arithmetic path (re-derived here: the same test prep_candidate makes).
extern "C" int payne_emul_prep_counts(const double* wave, int npix, double r_ann, const double* obs_wave, int nobs,
                                      const double* rv, const double* r_in, int n, double instr_factor, int* n_arith, int* first_bad) {
  HostTables H;
  int rc = build_model_tables(wave, npix, H);
  if (rc) return -1;
  std::vector<double> of(nobs, 1.0), oe(nobs, 0.01);
  build_obs_tables(obs_wave, of.data(), oe.data(), nobs, H);
  PostTables T;
  std::memset(&T, 0, sizeof(T));
  fill_model_scalars(H, T);
  T.nobs = nobs; T.lnlam = H.lnlam.data(); T.lam = H.lam.data();
  T.obs_min = H.obs_min; T.obs_max = H.obs_max; T.r_ann = r_ann;
  T.ln_obs_min = std::log(H.obs_min); T.ln_obs_max = std::log(H.obs_max);
  int bad = 0, arith = 0;
  if (first_bad) *first_bad = -1;
  for (int c = 0; c < n; ++c) {
    double th[12];
    for (int k = 0; k < 12; ++k) th[k] = 0.0;
    th[4] = rv[c]; th[5] = 1.0; th[7] = r_in[c];
    CandState S;
    std::memset(&S, 0, sizeof(S));
    prep_candidate(T, th, instr_factor, S);
    if (!S.win_ready) continue;
    int below = 0, notabove = 0;
    for (int i = 0; i < npix; ++i) {
      const double v = H.lam[i] * S.one_plus;
      if (!(v > S.wl)) ++below;
      if (v < S.wh) ++notabove;
    }
    if (T.geo) {
      const double pad = 20.0 / (r_in[c] * instr_factor);
      const double tl = ((T.ln_obs_min + log1p_series(pad * -1.0)) - S.dop - T.ln0) * T.geo_inv_dln;
      const double th_ = ((T.ln_obs_max + log1p_series(pad * 1.0)) - S.dop - T.ln0) * T.geo_inv_dln;
      const double fl = std::floor(tl), fh = std::floor(th_);
      if ((tl - fl > 1e-5) && (tl - fl < 1.0 - 1e-5) && (th_ - fh > 1e-5) && (th_ - fh < 1.0 - 1e-5)) ++arith;
    }
    if (below != S.win_below || notabove != S.win_notabove) { if (!bad && first_bad) *first_bad = c; ++bad; }
  }
  if (n_arith) *n_arith = arith;
  return bad;
}
