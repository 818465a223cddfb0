// payne_hip.hip -- the C ABI of include/payne_hip.h: contexts, launches, entry points.  One translation
// unit; the device code is included from dense_kernels.hpp (ANN layers), post_kernels.hpp + post_core.hpp +
// post_seq.hpp (spectrum pipeline), sed_kernel.hpp (photometry), sampler_kernels.hpp (sampler step), the
// host-only parts from host_tables.hpp (theta-independent tables) and ns_core.hpp (sampler bookkeeping).
//
// Kernels (all hand-written for CDNA4, wave64):
//   payne_dense_kernel  fp32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32 fma chain) dense layer
//                       Y = act(X W^T + b) over the batch of candidates; LDS-tiled 64xBNx32,
//                       register-prefetch double buffering, XCD-aware tile order.  With FUSE_L0
//                       the A operand is produced on the fly from theta: label encoding
//                       (ystpred.py:47-50) + first layer + activation, so a YST1 forward pass
//                       (ystpred.py:52-58) is two launches.
//   payne_post_kernel   one 256-thread workgroup per candidate; the spectrum lives in LDS from
//                       the ANN output to chi^2: vsini FFT stage, Doppler, masked pow-2
//                       resample, Gaussian FFT stage, interpolation to the observed grid,
//                       blaze, chi^2 (phases in post_core.hpp, order in post_seq.hpp).
//   payne_sed_kernel    one wave per (candidate, filter): the stacked photometric nets of
//                       photANN.fastANN + highAv + the magnitude formulae of predictsed.py.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/payne_hip.h"
#include "host_tables.hpp"
#include "post_seq.hpp"
#include "ns_core.hpp"

using namespace payne;

#include "dense_kernels.hpp"

#include "post_kernels.hpp"

#include "sed_kernel.hpp"
#include "select.hpp"

// photometry-only fits: lnL = -0.5 chi2_sed
__global__ void payne_photonly_kernel(const double* mags, const double* obs, const double* err, int F, int B, double* lnl) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) lnl[b] = -0.5 * sed_chi2(mags + (size_t)b * F, obs, err, F);
}

// ============================================================================
// context
// ============================================================================
static std::string g_create_error;

struct payne_ctx {
  int device = 0;
  payne_opts opts{};
  std::string err;
  std::vector<void*> owned;       // freed at destroy
  std::vector<void*> obs_owned;   // freed when the observed grid is re-bound
  // spectral model
  bool has_model = false;
  int n_layers = 0;
  payne_layer layers[PAYNE_MAX_LAYERS];
  int n_labels = 0;
  double xmin[PAYNE_MAX_LABELS], xden[PAYNE_MAX_LABELS];
  HostTables H;
  PostTables T{};
  float* hid[2] = {nullptr, nullptr};
  int ld_hid = 0;
  float* raw = nullptr;
  const float* w_out_pad = nullptr;     // output layer's weights [N][w_out_kp], k zero-padded to a multiple of 32 (LDS-DMA kernel)
  const float* w_hid_pad[PAYNE_MAX_LAYERS] = {};   // hidden layers: [N][ld_hid] copies, zero beyond K (hk_tile's LDS-DMA staging)
  int w_out_kp = 0;
  bool dma_ok = false;                  // hidden buffers are zero beyond the last hidden width
  // 3 x bf16 planes for payne_dense_dma3_kernel: the output layer's weights [3][N][w_out_kp], the last hidden layer's output
  // [3][b_max][ld_hid] (written by the hidden-layer kernel's epilogue; zero beyond the hidden width)
  unsigned short* w_out_p3 = nullptr; unsigned short* hid_p3 = nullptr;
  // the same output layer restated for rows in the frequency domain (host_tables.hpp freq_rows): what the likelihood and the
  // predictions past stage 0 run when the post kernel can start from the transform (freq_ok); raw_freq: the rows now in c->raw
  unsigned short* w_out_p3z = nullptr; const float* bias_z = nullptr; bool freq_ok = false, raw_freq = false;
  const float* w_out_padz = nullptr;    // the restated layer as fp32 [n][w_out_kp] (payne_dense_dma3f_kernel splits it on the way into LDS)
  // two fp16 planes an operand (payne_dense_dma2h_kernel): the weights' planes [2][n][w_out_kp] with row n scaled by 2^e[n], rscale[n] =
  // 2^-e[n] / act_scale; act_scale: the power of two the last hidden layer is written with (calibrated on the label box; 0 = not available)
  unsigned short* w_out_h2 = nullptr; unsigned short* w_out_h2z = nullptr; const float* rscale = nullptr; const float* rscalez = nullptr;
  float act_scale = 0.f;
  // the second layer on fp16 pairs (hk_tile_h2): its weights as two fp16 planes [2][n_out][304], rows scaled; rs1 = 2^-e / a0_scale
  unsigned short* w1_h2 = nullptr; const float* rs1 = nullptr; float a0_scale = 0.f; int w1_rows = 0;
  // ... and the hidden layers past the second (hk_tile_h2x): layer l's weights as planes, rs = 2^-e / hs[l - 1]; hs[l]: the power of two
  // layer l's output is written with; hid_h2: the activations between two such layers as planes [2][b_max][304] (two buffers taking turns)
  unsigned short* wl_h2[PAYNE_MAX_LAYERS] = {}; const float* rsl[PAYNE_MAX_LAYERS] = {}; float hs[PAYNE_MAX_LAYERS] = {};
  unsigned short* hid_h2[2] = {nullptr, nullptr};
  // payne_dense_chain_kernel: the row blocks' hop counters [ceil(b_max / 32)][kChainMax] (never reset) and the calls made so far
  unsigned long long* chain_flags = nullptr; unsigned long long chain_calls = 0;
  // freq_rs: the restated layer is that of the RESAMPLED spectrum (model grids that are not a power of two long: n1 rows of n1
  // values); a batch with a candidate that does not rotate falls back to pixels ON THE DEVICE (rot_flag: a word the records'
  // writers set to rot_seq, read by the output layer and the post kernel of the same batch; freq_rs_now: this batch was launched so)
  bool freq_rs = false, freq_rs_now = false; unsigned long long* rot_flag = nullptr; unsigned long long rot_seq = 0;
  size_t post_lds = 0;
  void (*post_fn_lean)(PAYNE_POST_SIG) = nullptr;   // likelihood-only instantiation (same LDS)
  bool post_tw_lds = false;
  int n_cu = 256;                       // compute units of the device (MI355X: 256)
  bool lean_available = false;          // a likelihood-only instantiation exists for this spectrum length (it can carry a walk's tail)
  // a sampler's walk in progress asks the likelihood batch being enqueued to make the next step's proposals ahead (rwalk_spec_wave,
  // in the hidden-layer launch): set by lnlike_impl for the duration of the call; spec_launched: that launch carried them
  const void* spec_walk = nullptr; const void* spec_w = nullptr; int spec_step = 0, spec_K = 0; bool spec_launched = false;
  PostTables* d_T = nullptr;          // device copy of T
  post_kernel_fn post_fn = nullptr;
  float* big_ws = nullptr;            // global spectrum buffers of payne_post_big_kernel (n1 > 16384)
  int big_grid = 0;
  bool big_tiled = false;             // ... with the four-step transform (LDS tile attribute set at create)
  bool big_chip = false;              // ... or with the convolution stages on the compute unit (65 536 points: payne_post_chip_kernel)
  bool big_chip2 = false;             // ... 32 768 points, two candidates at a time (payne_post_chip2_kernel)
  // optional continuum network (payne_ctx_set_continuum; ystpred.py:81-85, 191-209)
  bool has_cont = false;
  int cn_layers = 0, cn_npix = 0, cn_ld_hid = 0;
  payne_layer clayers[PAYNE_MAX_LAYERS];
  double cxmin[PAYNE_MAX_LABELS], cxden[PAYNE_MAX_LABELS];
  float* chid[2] = {nullptr, nullptr};
  float* cont_raw = nullptr;            // [b_max][cn_npix] continuum ANN output (F_nu)
  const double* cont_scale = nullptr;   // [cn_npix] (lam_ref / lam_c)^2 : F_nu -> F_lambda up to a constant the median removes
  const int* cont_idx = nullptr;        // [npix] np.interp(modwave, modcontwave, .) map: left pixel (-1: outside -> NaN)
  const double* cont_frac = nullptr;    // [npix] weight of the right pixel
  std::vector<void*> cont_owned;
  // optional LSF vector (payne_ctx_set_lsf): dispersion in AA per bound observed pixel; replaces Inst_R
  bool has_lsf = false;
  const double* d_obs_wave = nullptr;   // [nobs] (obs_owned)
  const double* lsf = nullptr;          // [n_lsf] dispersions ...
  const double* lsf_wave = nullptr;     // ... at these wavelengths (the observed grid itself for payne_ctx_set_lsf)
  int n_lsf = 0;
  float* lsf_spec = nullptr;            // [lsf_chunk][npix] spectra after vsini, shifted
  double* lsf_ws = nullptr;             // [lsf_chunk][2 npix + n1]
  float* lsf_fws = nullptr;             // global form: [lsf_chunk][2 fft_buf_floats(n1)] FFT buffers
  bool lsf_global = false;              // spectra too long for LDS (n1 > 8192): buffers in global memory, median by selection
  int lsf_chunk = 0;                    // candidates per launch (the global form walks the batch in chunks: bounded workspace)
  std::vector<void*> lsf_owned;
  bool obs_bound = false;
  CandState* prep = nullptr;      // [b_max] per-candidate records of the post kernel (written by the first dense launch)
  bool prep_valid = false;        // ... as of the last run_ann
  // photometry
  bool has_phot = false, has_obs_phot = false;
  PhotTables P{};
  double* mags_ws = nullptr;
  double *obs_mag = nullptr, *obs_err = nullptr;
  int ncols = 0;
  // optional per-kernel HIP-event timing (payne_profile): kinds 0 output dense layer,
  // 1 post, 2 sed, 3 hidden dense layers
  bool prof = false;
  struct ProfRec { hipEvent_t e0, e1; int kind; };
  std::vector<ProfRec> prof_pool;
  size_t prof_used = 0;
  double prof_ms[4] = {0, 0, 0, 0};
  long long prof_n[4] = {0, 0, 0, 0};
  const char* last_kernel[4] = {"", "", "", ""};   // what the last call launched, per kind (payne_last_kernel)
};

// RAII bracket of one timed launch.  The event pair is handed to the launch itself (hipExtLaunchKernelGGL:
// start/stop are the kernel's own dispatch timestamps -- what rocprofv3 reports); events recorded AROUND a
// launch on the stream read ~2 us more (their own packets).  A scope that sees no PAYNE_LAUNCH gives its
// record back; a scope with several launches times the first.
struct ProfScope;
static thread_local ProfScope* g_prof_scope = nullptr;
struct ProfScope {
  payne_ctx* c; hipStream_t s; payne_ctx::ProfRec* r = nullptr; bool used = false; ProfScope* outer = nullptr; int kind;
  ProfScope(payne_ctx* c_, hipStream_t s_, int kind_) : c(c_), s(s_), kind(kind_) {
    outer = g_prof_scope; g_prof_scope = this;
    if (!c->prof) return;
    if (c->prof_used == c->prof_pool.size()) {
      payne_ctx::ProfRec n{};
      if (hipEventCreate(&n.e0) != hipSuccess || hipEventCreate(&n.e1) != hipSuccess) return;
      c->prof_pool.push_back(n);
    }
    r = &c->prof_pool[c->prof_used++];
    r->kind = kind;
  }
  ~ProfScope() {
    g_prof_scope = outer;
    if (r && !used) --c->prof_used;                      // nothing was launched under this scope
  }
};
#include <hip/hip_ext.h>
#define PAYNE_LAUNCH(kernel, grid, block, lds, stream, ...)                                               \
  do {                                                                                                    \
    ProfScope* ps_ = g_prof_scope;                                                                        \
    if (ps_) ps_->c->last_kernel[ps_->kind] = #kernel;                                                    \
    if (ps_ && ps_->r && !ps_->used) {                                                                    \
      ps_->used = true;                                                                                   \
      hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)(lds), stream, ps_->r->e0, ps_->r->e1, 0, __VA_ARGS__); \
    } else {                                                                                              \
      hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                  \
    }                                                                                                     \
  } while (0)

#define HIPCHK(ctx, call)                                                                 \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                     \
      return PAYNE_E_HIP;                                                                 \
    }                                                                                     \
  } while (0)

template <class V>
static int upload(payne_ctx* c, const std::vector<V>& v, const V** out, std::vector<void*>& bag) {
  void* d = nullptr;
  size_t bytes = v.size() * sizeof(V);
  if (bytes == 0) { *out = nullptr; return PAYNE_OK; }
  HIPCHK(c, hipMalloc(&d, bytes));
  bag.push_back(d);
  HIPCHK(c, hipMemcpy(d, v.data(), bytes, hipMemcpyHostToDevice));
  *out = reinterpret_cast<const V*>(d);
  return PAYNE_OK;
}
template <class V>
static int dev_alloc(payne_ctx* c, size_t n, V** out, std::vector<void*>& bag, bool zero = true) {
  void* d = nullptr;
  HIPCHK(c, hipMalloc(&d, n * sizeof(V)));
  bag.push_back(d);
  if (zero) HIPCHK(c, hipMemset(d, 0, n * sizeof(V)));
  *out = reinterpret_cast<V*>(d);
  return PAYNE_OK;
}

static int fail(payne_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg; else g_create_error = msg;
  return code;
}

static int sync_tables(payne_ctx* c) {
  if (!c->d_T) return PAYNE_OK;
  HIPCHK(c, hipMemcpy(c->d_T, &c->T, sizeof(PostTables), hipMemcpyHostToDevice));
  return PAYNE_OK;
}

static int bind_obs(payne_ctx* c, const payne_obs_desc* obs) {
  for (void* p : c->obs_owned) (void)hipFree(p);
  c->obs_owned.clear();
  c->obs_bound = false;
  for (void* p : c->lsf_owned) (void)hipFree(p);      // an LSF vector belongs to the grid it was given on
  c->lsf_owned.clear(); c->has_lsf = false; c->d_obs_wave = nullptr;
  c->T.obs_sorted = 0;
  c->T.nobs = 0; c->T.lnobs = nullptr; c->T.obs_rec = nullptr; c->T.xcheb = nullptr; c->T.obs_f1 = nullptr; c->T.obs_ivar = nullptr;
  if (!obs || obs->nobs <= 0) return sync_tables(c);
  if (!obs->wave) return fail(c, PAYNE_E_INVALID, "obs.wave is NULL");
  if ((obs->flux == nullptr) != (obs->eflux == nullptr)) return fail(c, PAYNE_E_INVALID, "obs.flux and obs.eflux must both be given or both NULL");
  build_obs_tables(obs->wave, obs->flux, obs->eflux, obs->nobs, c->H);
  int rc;
  if ((rc = upload(c, c->H.lnobs, &c->T.lnobs, c->obs_owned))) return rc;
  if ((rc = upload(c, c->H.obs_rec, &c->T.obs_rec, c->obs_owned))) return rc;
  if ((rc = upload(c, c->H.obs_wave, &c->d_obs_wave, c->obs_owned))) return rc;
  if ((rc = upload(c, c->H.xcheb, &c->T.xcheb, c->obs_owned))) return rc;
  if (c->H.has_flux) {
    if ((rc = upload(c, c->H.obs_f1, &c->T.obs_f1, c->obs_owned))) return rc;
    if ((rc = upload(c, c->H.obs_ivar, &c->T.obs_ivar, c->obs_owned))) return rc;
  }
  c->T.nobs = obs->nobs;
  c->T.obs_min = c->H.obs_min;
  c->T.obs_max = c->H.obs_max;
  c->T.ln_obs_min = std::log(c->H.obs_min); c->T.ln_obs_max = std::log(c->H.obs_max);
  c->T.obs_sorted = 1;
  for (int i = 1; i < obs->nobs; ++i) if (!(obs->wave[i] >= obs->wave[i - 1])) { c->T.obs_sorted = 0; break; }
  c->obs_bound = true;
  return sync_tables(c);
}

extern "C" int payne_version(void) { return PAYNE_ABI_VERSION; }

extern "C" const char* payne_last_error(const payne_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int payne_theta_cols(const payne_ctx* ctx) { return ctx ? ctx->ncols : PAYNE_E_INVALID; }

extern "C" const char* payne_kernel_name(int which) {
  switch (which) {
    case 0: return "payne_dense_kernel";
    case 1: return "payne_post_kernel";
    case 2: return "payne_sed_kernel";
    default: return "";
  }
}

__global__ void payne_activation_kernel(const float* __restrict__ z, int n, int act, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = act_apply(z[i], act);
}
extern "C" int payne_activation_batch(const float* z, int n, int act, float* out, void* stream) {
  if (!z || !out || n < 0 || act < PAYNE_ACT_NONE || act > PAYNE_ACT_SIGMOID) return PAYNE_E_INVALID;
  if (n == 0) return PAYNE_OK;
  hipLaunchKernelGGL(payne_activation_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), z, n, act, out);
  return hipGetLastError() == hipSuccess ? PAYNE_OK : PAYNE_E_HIP;
}
extern "C" const char* payne_last_kernel(const payne_ctx* c, int kind) {
  if (c && kind == 4) return c->raw_freq ? "frequency" : "pixels";        // what the output layer handed to the post kernel
  return (c && kind >= 0 && kind < 4) ? c->last_kernel[kind] : "";
}

extern "C" void payne_ctx_destroy(payne_ctx* c) {
  if (!c) return;
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(c->device);
  for (void* p : c->owned) (void)hipFree(p);
  for (void* p : c->obs_owned) (void)hipFree(p);
  for (void* p : c->cont_owned) (void)hipFree(p);
  for (void* p : c->lsf_owned) (void)hipFree(p);
  for (auto& r : c->prof_pool) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  (void)hipSetDevice(prev);
  delete c;
}

static hipError_t set_dense_attributes();
// ---- two fp16 planes an operand (dense_kernels.hpp split2h) -----------------------------------------------------------
// The largest |activation| of the last hidden layer over the label box (corners, centre, 64 points of a fixed sequence, 20 % beyond
// the box on every side), evaluated on the host in fp64 from the layers as the context holds them.  0 when the net has no hidden
// layer or produces something that is not finite.
static double hidden_amax(const payne_model_desc* m, std::string& why, double* per_layer = nullptr) {      // (per_layer[l]: the same for layer l's output, l < n_layers - 1)
  const int nl = m->n_layers, D = m->n_labels;
  if (nl < 3) { why = "no hidden layer pair"; return 0.0; }
  std::vector<std::vector<float>> W(nl - 1), b(nl - 1);
  for (int l = 0; l + 1 < nl; ++l) {
    const payne_layer& L = m->layers[l];
    W[l].resize((size_t)L.n_out * L.n_in); b[l].resize((size_t)L.n_out);
    if (hipMemcpy(W[l].data(), L.w, W[l].size() * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(b[l].data(), L.b, b[l].size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { why = "hipMemcpy"; return 0.0; }
  }
  auto act = [](double z, int a) { return a == PAYNE_ACT_LRELU ? (z > 0 ? z : 0.01 * z) : (a == PAYNE_ACT_SIGMOID ? 1.0 / (1.0 + std::exp(-z)) : z); };
  std::vector<std::vector<double>> pts;
  for (int cidx = 0; cidx < (1 << D); ++cidx) { std::vector<double> x(D); for (int d = 0; d < D; ++d) x[d] = ((cidx >> d) & 1) ? 0.6 : -0.6; pts.push_back(x); }
  pts.push_back(std::vector<double>(D, 0.0));
  unsigned long long st = 0x9E3779B97F4A7C15ull;
  for (int i = 0; i < 64; ++i) {
    std::vector<double> x(D);
    for (int d = 0; d < D; ++d) { st = st * 6364136223846793005ull + 1442695040888963407ull; x[d] = ((double)(st >> 11) / 9007199254740992.0 - 0.5) * 1.2; }
    pts.push_back(x);
  }
  double amax = 0.0;
  std::vector<double> a, y;
  for (const auto& x : pts) {
    a = x;
    for (int l = 0; l + 1 < nl; ++l) {
      const payne_layer& L = m->layers[l];
      y.assign((size_t)L.n_out, 0.0);
      for (int o = 0; o < L.n_out; ++o) {
        double z = b[l][o];
        const float* w = &W[l][(size_t)o * L.n_in];
        for (int k = 0; k < L.n_in; ++k) z += (double)w[k] * a[k];
        y[o] = act(z, L.act);
      }
      a.swap(y);
      if (per_layer) for (double v : a) { if (std::isfinite(v)) per_layer[l] = std::max(per_layer[l], std::fabs(v)); else per_layer[l] = 1e300; }
    }
    for (double v : a) { if (!std::isfinite(v)) { why = "non-finite activation"; return 0.0; } amax = std::max(amax, std::fabs(v)); }
  }
  return amax;
}
// per-row power-of-two scales of a padded weight matrix [n][kp] (host copy) and its two fp16 planes on the device
static int make_h2_planes(payne_ctx* c, const float* d_w, const std::vector<float>& h_w, int n, int kp, float act_scale,
                          unsigned short** planes, const float** rscale) {
  std::vector<float> sc((size_t)n), rs((size_t)n);
  for (int i = 0; i < n; ++i) {
    float mx = 0.f;
    for (int k = 0; k < kp; ++k) mx = std::max(mx, std::fabs(h_w[(size_t)i * kp + k]));
    int e = 0;
    if (mx > 0.f && std::isfinite(mx)) e = (int)std::floor(std::log2(16384.0 / (double)mx));
    e = std::max(-100, std::min(100, e));
    sc[i] = (float)std::ldexp(1.0, e);
    rs[i] = (float)(std::ldexp(1.0, -e) / (double)act_scale);
  }
  const float* d_sc = nullptr;
  std::vector<void*> tmp;
  int rc = upload(c, sc, &d_sc, tmp);
  if (!rc) rc = upload(c, rs, rscale, c->owned);
  const size_t nw = (size_t)n * kp;
  if (!rc) rc = dev_alloc(c, 2 * nw, planes, c->owned);
  hipError_t he = hipSuccess;
  if (!rc) {
    hipLaunchKernelGGL(payne_split2h_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, nullptr, d_w, n, kp, d_sc, *planes, nw);
    he = hipDeviceSynchronize();
  }
  for (void* q : tmp) (void)hipFree(q);
  if (rc) return rc;
  if (he != hipSuccess) return fail(c, PAYNE_E_HIP, std::string("weight split (fp16 planes): ") + hipGetErrorString(he));
  return PAYNE_OK;
}

extern "C" int payne_ctx_create(const payne_model_desc* model, const payne_obs_desc* obs, const payne_phot_desc* phot,
                                const payne_opts* opts, int device, payne_ctx** out) {
  if (!out) return fail(nullptr, PAYNE_E_INVALID, "out is NULL");
  *out = nullptr;
  if (!opts || opts->b_max <= 0) return fail(nullptr, PAYNE_E_INVALID, "opts.b_max must be > 0");
  if (opts->npoly < 0 || opts->npoly > PAYNE_MAX_POLY) return fail(nullptr, PAYNE_E_INVALID, "opts.npoly out of range");
  if (!model && !phot) return fail(nullptr, PAYNE_E_INVALID, "need a spectral model and/or a photometric model");
  hipError_t he = hipSetDevice(device);
  if (he != hipSuccess) return fail(nullptr, PAYNE_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(he));
  payne_ctx* c = new payne_ctx();
  c->device = device;
  { int ncu = 0; if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) c->n_cu = ncu; }
  c->opts = *opts;
  c->ncols = 8 + opts->npoly + 4;
  int rc = PAYNE_OK;
  auto bail = [&](int code) { g_create_error = c->err; payne_ctx_destroy(c); return code; };
  he = set_dense_attributes();
  if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipFuncSetAttribute(dense): ") + hipGetErrorString(he)));

  if (model) {
    if (model->n_layers < 2 || model->n_layers > PAYNE_MAX_LAYERS) return bail(fail(c, PAYNE_E_INVALID, "model.n_layers must be 2..8"));
    if (model->n_labels < 1 || model->n_labels > PAYNE_MAX_LABELS) return bail(fail(c, PAYNE_E_INVALID, "model.n_labels must be 1..5"));
    if (!model->xmin || !model->xmax || !model->wavelength) return bail(fail(c, PAYNE_E_INVALID, "model.xmin/xmax/wavelength missing"));
    if (model->layers[0].n_in != model->n_labels) return bail(fail(c, PAYNE_E_INVALID, "first layer n_in != n_labels"));
    if (model->layers[model->n_layers - 1].n_out != model->npix) return bail(fail(c, PAYNE_E_INVALID, "last layer n_out != npix"));
    int maxh = 0;
    for (int l = 0; l < model->n_layers; ++l) {
      const payne_layer& L = model->layers[l];
      if (!L.w || !L.b || L.n_in <= 0 || L.n_out <= 0) return bail(fail(c, PAYNE_E_INVALID, "layer with null weights or bad shape"));
      if (l > 0 && L.n_in != model->layers[l - 1].n_out) return bail(fail(c, PAYNE_E_INVALID, "layer shapes do not chain"));
      c->layers[l] = L;
      if (l > 0 && (L.n_in & 3)) {   // float4 tile loads need K % 4 == 0: keep a zero-padded copy
        const int Kp = (L.n_in + 3) & ~3;
        float* wp = nullptr;
        if ((rc = dev_alloc(c, (size_t)L.n_out * Kp, &wp, c->owned))) return bail(rc);
        he = hipMemcpy2D(wp, (size_t)Kp * 4, L.w, (size_t)L.n_in * 4, (size_t)L.n_in * 4, L.n_out, hipMemcpyDeviceToDevice);
        if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipMemcpy2D: ") + hipGetErrorString(he)));
        c->layers[l].w = wp;
        c->layers[l].n_in = Kp;     // padded K (extra columns are zero)
      }
      if (l + 1 < model->n_layers) maxh = std::max(maxh, L.n_out);
    }
    {   // k-padded copy of the output layer's weights for the LDS-DMA kernel (operands cannot be masked on the way)
      const payne_layer& L = c->layers[model->n_layers - 1];
      const int Kp = (L.n_in + 31) & ~31;
      float* wp = nullptr;
      if ((rc = dev_alloc(c, (size_t)L.n_out * Kp, &wp, c->owned))) return bail(rc);
      he = hipMemcpy2D(wp, (size_t)Kp * 4, L.w, (size_t)L.n_in * 4, (size_t)L.n_in * 4, L.n_out, hipMemcpyDeviceToDevice);
      if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipMemcpy2D: ") + hipGetErrorString(he)));
      c->w_out_pad = wp; c->w_out_kp = Kp;
      // the activations' pad columns are zero only if no wider layer ever wrote them: all hidden widths equal
      bool same = model->n_layers >= 3;
      for (int l = 1; l + 1 < model->n_layers; ++l) same = same && model->layers[l].n_out == model->layers[0].n_out;
      c->dma_ok = same;
      // hidden layers past the second (launch_hidden<false>): operand tiles straight into LDS need rows >= HK_PITCH floats apart that
      // may be read to their end -- the activations' buffers are (pitch ld_hid, pad columns zero while all widths are equal); the
      // weights get a copy with the same pitch
      const int ldh = (maxh + 31) & ~31;
      if (same && ldh >= HK_PITCH) {
        for (int l = 1; l + 1 < model->n_layers; ++l) {
          const payne_layer& Lh = c->layers[l];
          if (Lh.n_in > HK_KC) continue;
          float* wh = nullptr;
          if ((rc = dev_alloc(c, (size_t)Lh.n_out * ldh, &wh, c->owned))) return bail(rc);
          he = hipMemcpy2D(wh, (size_t)ldh * 4, Lh.w, (size_t)Lh.n_in * 4, (size_t)Lh.n_in * 4, Lh.n_out, hipMemcpyDeviceToDevice);
          if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipMemcpy2D: ") + hipGetErrorString(he)));
          c->w_hid_pad[l] = wh;
        }
      }
      if (same) {
        const size_t nw = (size_t)L.n_out * Kp;
        if ((rc = dev_alloc(c, 3 * nw, &c->w_out_p3, c->owned))) return bail(rc);
        hipLaunchKernelGGL(payne_split3_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, nullptr, wp, nw, c->w_out_p3, nw);
        he = hipDeviceSynchronize();
        if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("weight split: ") + hipGetErrorString(he)));
        // ... and as two fp16 planes, the activations' scale calibrated on the label box (a factor of 8 to spare below fp16's range)
        std::string why;
        double amaxl[PAYNE_MAX_LAYERS] = {};
        const bool want_h2 = (opts->variant & (PAYNE_V_OUT_BF16X3 | PAYNE_V_OUT_PLANES | PAYNE_V_OUT_F32 | PAYNE_V_OUT_GENERIC | PAYNE_V_OUT_BK64)) == 0 ||
                             !(opts->variant & PAYNE_V_HID_F32);
        const double amax = want_h2 ? hidden_amax(model, why, amaxl) : 0.0;
        const double amax0 = amaxl[0];
        // the second layer's weights for hk_tile_h2: widths whose padded K is the tile's 304 columns
        const payne_layer& L1 = model->layers[1];
        if (amax0 > 0.0 && amax0 < 1e30 && L1.n_in > 288 && L1.n_in <= 304 && c->w_hid_pad[1] && !(opts->variant & PAYNE_V_HID_F32)) {
          c->a0_scale = (float)std::ldexp(1.0, std::max(-60, std::min(60, (int)std::floor(std::log2(4096.0 / amax0)))));
          std::vector<float> h1((size_t)L1.n_out * L1.n_in), hp((size_t)L1.n_out * 304, 0.f);
          he = hipMemcpy(h1.data(), L1.w, h1.size() * 4, hipMemcpyDeviceToHost);
          if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipMemcpy(second layer): ") + hipGetErrorString(he)));
          for (int i = 0; i < L1.n_out; ++i) std::copy(h1.begin() + (size_t)i * L1.n_in, h1.begin() + (size_t)(i + 1) * L1.n_in, hp.begin() + (size_t)i * 304);
          const float* d_hp = nullptr;
          std::vector<void*> tmp1;
          if ((rc = upload(c, hp, &d_hp, tmp1))) return bail(rc);
          rc = make_h2_planes(c, d_hp, hp, L1.n_out, 304, c->a0_scale, &c->w1_h2, &c->rs1);
          for (void* q : tmp1) (void)hipFree(q);
          if (rc) return bail(rc);
          c->w1_rows = L1.n_out;
          c->hs[0] = c->a0_scale;
          // deeper nets (LinNet: five hidden layers): the layers past the second the same way, their activations handed on as planes
          bool deep = model->n_layers > 3;
          for (int l = 1; l + 1 < model->n_layers; ++l) deep = deep && amaxl[l] > 0.0 && amaxl[l] < 1e30 && model->layers[l].n_out == L1.n_out;
          if (deep) {
            for (int l = 1; l + 1 < model->n_layers; ++l)
              c->hs[l] = (float)std::ldexp(1.0, std::max(-60, std::min(60, (int)std::floor(std::log2(4096.0 / amaxl[l])))));
            for (int l = 2; l + 1 < model->n_layers; ++l) {
              const payne_layer& Ll = model->layers[l];
              std::vector<float> hl((size_t)Ll.n_out * Ll.n_in), hq((size_t)Ll.n_out * 304, 0.f);
              he = hipMemcpy(hl.data(), Ll.w, hl.size() * 4, hipMemcpyDeviceToHost);
              if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipMemcpy(hidden layer): ") + hipGetErrorString(he)));
              for (int i = 0; i < Ll.n_out; ++i) std::copy(hl.begin() + (size_t)i * Ll.n_in, hl.begin() + (size_t)(i + 1) * Ll.n_in, hq.begin() + (size_t)i * 304);
              const float* d_hq = nullptr;
              std::vector<void*> tmp2;
              if ((rc = upload(c, hq, &d_hq, tmp2))) return bail(rc);
              rc = make_h2_planes(c, d_hq, hq, Ll.n_out, 304, c->hs[l - 1], &c->wl_h2[l], &c->rsl[l]);
              for (void* q : tmp2) (void)hipFree(q);
              if (rc) return bail(rc);
            }
            for (int q = 0; q < 2; ++q)
              if ((rc = dev_alloc(c, (size_t)2 * opts->b_max * 304, &c->hid_h2[q], c->owned))) return bail(rc);
            if ((rc = dev_alloc(c, (size_t)((opts->b_max + 31) / 32) * kChainMax, &c->chain_flags, c->owned))) return bail(rc);
          }
        }
        if (amax > 0.0 && amax < 1e30 && !(opts->variant & (PAYNE_V_OUT_BF16X3 | PAYNE_V_OUT_PLANES | PAYNE_V_OUT_F32 | PAYNE_V_OUT_GENERIC | PAYNE_V_OUT_BK64))) {
          c->act_scale = (float)std::ldexp(1.0, std::max(-60, std::min(60, (int)std::floor(std::log2(4096.0 / amax)))));
          std::vector<float> hw(nw);
          he = hipMemcpy(hw.data(), wp, nw * 4, hipMemcpyDeviceToHost);
          if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipMemcpy(output layer): ") + hipGetErrorString(he)));
          if ((rc = make_h2_planes(c, wp, hw, L.n_out, Kp, c->act_scale, &c->w_out_h2, &c->rscale))) return bail(rc);
        }
      }
    }
    c->n_layers = model->n_layers;
    c->n_labels = model->n_labels;
    for (int d = 0; d < model->n_labels; ++d) { c->xmin[d] = model->xmin[d]; c->xden[d] = model->xmax[d] - model->xmin[d]; }
    rc = build_model_tables(model->wavelength, model->npix, c->H);
    if (rc == -1) return bail(fail(c, PAYNE_E_INVALID, "model.npix must be >= 16"));
    if (rc == -2) return bail(fail(c, PAYNE_E_INVALID, "model.wavelength must be strictly increasing"));
    if (c->H.n1 > (1 << 20)) return bail(fail(c, PAYNE_E_UNSUPPORTED, "npix > 2^20"));
    PostTables& T = c->T;
    fill_model_scalars(c->H, T);
    T.r_ann = model->resolution; T.npoly = opts->npoly;
    if ((rc = upload(c, c->H.vs_tab32, &T.vs_tab, c->owned))) return bail(rc);
    T.vs_tab += 1;                                           // (the table's entry of u = 0: one mirror entry sits in front of it)
    if ((rc = dev_alloc(c, 1, &c->d_T, c->owned))) return bail(rc);
    if ((rc = upload(c, c->H.lnlam, &T.lnlam, c->owned))) return bail(rc);
    if ((rc = upload(c, c->H.lam, &T.lam, c->owned))) return bail(rc);
    if ((rc = upload(c, c->H.tw, &T.tw, c->owned))) return bail(rc);
    if ((rc = upload(c, c->H.twf, &T.twf, c->owned))) return bail(rc);
    if ((rc = upload(c, c->H.rs1_idx, &T.rs1_idx, c->owned))) return bail(rc);
    if ((rc = upload(c, c->H.rs1_frac, &T.rs1_frac, c->owned))) return bail(rc);
    if ((rc = upload(c, c->H.bk1_idx, &T.bk1_idx, c->owned))) return bail(rc);
    if ((rc = upload(c, c->H.bk1_frac, &T.bk1_frac, c->owned))) return bail(rc);
    c->ld_hid = (maxh + 31) & ~31;
    if (model->n_layers > 2) {
      if ((rc = dev_alloc(c, (size_t)opts->b_max * c->ld_hid, &c->hid[0], c->owned))) return bail(rc);
      if ((rc = dev_alloc(c, (size_t)opts->b_max * c->ld_hid, &c->hid[1], c->owned))) return bail(rc);
      if (c->w_out_p3 && (rc = dev_alloc(c, (size_t)3 * opts->b_max * c->ld_hid, &c->hid_p3, c->owned))) return bail(rc);
    }
    if ((rc = dev_alloc(c, (size_t)opts->b_max * std::max(model->npix, T.n1 <= 16384 ? T.n1 : 0), &c->raw, c->owned, false))) return bail(rc);   // (rows of n1 values: freq_rs)
    if ((rc = dev_alloc(c, (size_t)opts->b_max, &c->prep, c->owned, false))) return bail(rc);
    if (T.n1 > 16384) {                // spectrum larger than LDS: global-workspace kernel
      c->big_grid = opts->b_max < 256 ? opts->b_max : 256;
      // 32 768-point spectra on a geometric grid: the on-chip stages for two candidates at a time (four buffers per workgroup)
      c->big_chip2 = T.n1 == kChip2N1 && c->H.geo && !(opts->variant & PAYNE_V_BIG_WORKSPACE);
      if ((rc = dev_alloc(c, (size_t)c->big_grid * (c->big_chip2 ? 4 : 2) * T.n1, &c->big_ws, c->owned, false))) return bail(rc);
      if (c->big_chip2) {
        he = hipFuncSetAttribute(reinterpret_cast<const void*>(payne_post_chip2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)kChipLdsBytes);
        if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipFuncSetAttribute(chip2): ") + hipGetErrorString(he)));
      }
      c->big_tiled = true;                 // the four-step transform where the length allows it (fft_tiled_ok); plain radix-8 passes otherwise
      // 65 536-point spectra on a geometric grid: the convolution stages stay on the compute unit (payne_post_chip_kernel)
      c->big_chip = T.n1 == kChipN1 && c->H.geo && !(opts->variant & PAYNE_V_BIG_WORKSPACE);
      if (c->big_chip) {
        he = hipFuncSetAttribute(reinterpret_cast<const void*>(payne_post_chip_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)kChipLdsBytes);
        if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipFuncSetAttribute(chip): ") + hipGetErrorString(he)));
      }
      if (c->big_tiled) {
        he = hipFuncSetAttribute(reinterpret_cast<const void*>(payne_post_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(2 * fft_tile_complex() * sizeof(c32)));
        if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipFuncSetAttribute(big): ") + hipGetErrorString(he)));
      }
    }
    c->post_lds = (size_t)(T.n1 > 16384 ? 64 : fft_buf_floats(T.n1)) * 8 + (size_t)scratch_doubles(kPostThreads) * 8 + ((sizeof(CandState) + 15) & ~(size_t)15) + 16;
    // twiddles in LDS while two workgroups still fit a CU (160 KiB); larger spectra read them from L2
    const size_t tw_bytes = c->H.twf.size() * sizeof(c32);         // ~0.75 n1 entries
    c->post_tw_lds = (c->post_lds + tw_bytes) <= 80 * 1024;
    if (opts->variant & PAYNE_V_TW_GLOBAL) c->post_tw_lds = false;
    if (c->post_tw_lds) c->post_lds += tw_bytes;
#ifdef PAYNE_STAMPS
    if (const char* e = getenv("PAYNE_DIAG_LDS")) c->post_lds = std::max(c->post_lds, (size_t)atoi(e));   // diagnostic build: one workgroup per CU
#endif
    const int geom_n1 = (opts->variant & PAYNE_V_POST_GENERIC) ? 0 : T.n1;
    c->post_fn = pick_post_kernel(geom_n1, c->post_tw_lds);
    c->post_fn_lean = (opts->variant & PAYNE_V_POST_FULL) ? c->post_fn : pick_post_kernel(geom_n1, c->post_tw_lds, true);
    c->lean_available = c->post_fn_lean != c->post_fn;
    he = hipFuncSetAttribute(reinterpret_cast<const void*>(c->post_fn_lean), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->post_lds);
    if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(he)));
    he = hipFuncSetAttribute(reinterpret_cast<const void*>(c->post_fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->post_lds);
    if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(he)));
    // The first convolution stage's forward transform is linear and the same for every candidate: with identity vsini maps and a
    // compile-time geometry the output layer writes the rows already transformed (weights = the transform of each hidden unit's
    // pixel vector, computed here once in fp64), and the post kernel starts at the taper.
    {
      const bool fixed = geom_n1 != 0 && ((c->post_tw_lds && (T.n1 == 1024 || T.n1 == 2048 || T.n1 == 4096)) || (!c->post_tw_lds && T.n1 == 8192));
      // (65 536 / 32 768 points with the stages on the compute unit: the rows in the order those kernels' registers hold the transform)
      // (a model grid of any other length -- what a trained network has: readc3k.py:441-447 -- is resampled by the rotation stage first,
      //  a static linear map as well: the LDS kernels' lengths take rows of the RESAMPLED spectrum's transform, freq_rs)
      const bool ident = T.rot_identity && T.n1 == model->npix;
      if (((fixed || c->big_chip || c->big_chip2) && ident || (fixed && !ident)) && c->w_out_p3 && c->hid_p3 && !(opts->variant & PAYNE_V_ROWS_PIXEL)) {
        const payne_layer& L = model->layers[model->n_layers - 1];
        const int K = L.n_in, Kp = c->w_out_kp, n = ident ? L.n_out : T.n1;
        std::vector<float> W((size_t)L.n_out * K), b((size_t)L.n_out), Wz, bz;
        he = hipMemcpy(W.data(), L.w, W.size() * 4, hipMemcpyDeviceToHost);
        if (he == hipSuccess) he = hipMemcpy(b.data(), L.b, b.size() * 4, hipMemcpyDeviceToHost);
        if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipMemcpy(output layer): ") + hipGetErrorString(he)));
        if (ident) freq_rows(W.data(), b.data(), -(double)kBase, n, K, Wz, bz, c->big_chip ? 1 : (c->big_chip2 ? 2 : 0));   // (rows are kept shifted by -1, as the pixel rows are)
        else {
          freq_rows(W.data(), b.data(), -(double)kBase, n, K, Wz, bz, 0, c->H.rs1_idx.data(), c->H.rs1_frac.data(), L.n_out);
          c->freq_rs = true;
          if ((rc = dev_alloc(c, (size_t)1, &c->rot_flag, c->owned))) return bail(rc);            // (zeroed; sequence numbers start at 1)
        }
        std::vector<float> Wp((size_t)n * Kp, 0.f);
        for (int i = 0; i < n; ++i) std::copy(Wz.begin() + (size_t)i * K, Wz.begin() + (size_t)(i + 1) * K, Wp.begin() + (size_t)i * Kp);
        const float* d_wp = nullptr;
        std::vector<void*> tmp;
        if ((rc = upload(c, Wp, &d_wp, c->owned))) return bail(rc);      // (kept: the fp32 form is what the one-tile-per-CU kernel reads)
        c->w_out_padz = d_wp;
        const size_t nw = (size_t)n * Kp;
        rc = dev_alloc(c, 3 * nw, &c->w_out_p3z, c->owned);
        if (!rc) {
          hipLaunchKernelGGL(payne_split3_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, nullptr, d_wp, nw, c->w_out_p3z, nw);
          he = hipDeviceSynchronize();
        }
        for (void* q : tmp) (void)hipFree(q);
        if (rc) return bail(rc);
        if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("weight split: ") + hipGetErrorString(he)));
        if ((rc = upload(c, bz, &c->bias_z, c->owned))) return bail(rc);
        if (c->w_out_h2 && (rc = make_h2_planes(c, d_wp, Wp, n, Kp, c->act_scale, &c->w_out_h2z, &c->rscalez))) return bail(rc);
        c->freq_ok = true;
      }
    }
    c->has_model = true;
    if ((rc = bind_obs(c, obs))) return bail(rc);
  }

  if (phot) {
    if (phot->n_filters <= 0 || phot->hidden <= 0 || phot->hidden > 2048) return bail(fail(c, PAYNE_E_INVALID, "phot.n_filters/hidden out of range"));
    if (!phot->w1 || !phot->b1 || !phot->w2 || !phot->b2 || !phot->w3 || !phot->b3 || !phot->xmin || !phot->xmax)
      return bail(fail(c, PAYNE_E_INVALID, "phot descriptor has NULL members"));
    PhotTables& P = c->P;
    const int F = phot->n_filters, H = phot->hidden;
    P.F = F; P.H = H;
    P.w1 = phot->w1; P.b1 = phot->b1; P.b2 = phot->b2; P.w3 = phot->w3; P.b3 = phot->b3;
    for (int d = 0; d < 6; ++d) { P.xmin[d] = phot->xmin[d]; P.xden[d] = phot->xmax[d] - phot->xmin[d]; }
    {   // w2 -> [F][k][h] so that lanes (h) read consecutive addresses
      std::vector<float> w2((size_t)F * H * H), w2t((size_t)F * H * H);
      he = hipMemcpy(w2.data(), phot->w2, w2.size() * 4, hipMemcpyDeviceToHost);
      if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipMemcpy(w2): ") + hipGetErrorString(he)));
      for (int f = 0; f < F; ++f)
        for (int h = 0; h < H; ++h)
          for (int k = 0; k < H; ++k) w2t[((size_t)f * H + k) * H + h] = w2[((size_t)f * H + h) * H + k];
      if ((rc = upload(c, w2t, &P.w2t, c->owned))) return bail(rc);
    }
    if (phot->hiav) {
      std::vector<double> hv(phot->hiav, phot->hiav + (size_t)F * 5);
      if ((rc = upload(c, hv, &P.hiav, c->owned))) return bail(rc);
    }
    if (phot->obs_mag && phot->obs_err) {
      std::vector<double> m(phot->obs_mag, phot->obs_mag + F), e(phot->obs_err, phot->obs_err + F);
      const double *dm, *de;
      if ((rc = upload(c, m, &dm, c->owned))) return bail(rc);
      if ((rc = upload(c, e, &de, c->owned))) return bail(rc);
      c->obs_mag = const_cast<double*>(dm); c->obs_err = const_cast<double*>(de);
      c->has_obs_phot = true;
    }
    if ((rc = dev_alloc(c, (size_t)opts->b_max * F, &c->mags_ws, c->owned))) return bail(rc);
    if ((size_t)H * 16 > 48 * 1024) {
      he = hipFuncSetAttribute(reinterpret_cast<const void*>(payne_sed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, H * 16);
      if (he != hipSuccess) return bail(fail(c, PAYNE_E_HIP, std::string("hipFuncSetAttribute(sed): ") + hipGetErrorString(he)));
    }
    c->has_phot = true;
  }
  *out = c;
  return PAYNE_OK;
}

extern "C" int payne_ctx_set_obs(payne_ctx* c, const payne_obs_desc* obs) {
  if (!c) return PAYNE_E_INVALID;
  if (!c->has_model) return fail(c, PAYNE_E_INVALID, "context has no spectral model");
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (prev != c->device) (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();           // no kernel may still read the old tables
  int rc = bind_obs(c, obs);
  if (prev != c->device) (void)hipSetDevice(prev);
  return rc;
}

// Continuum network (ystpred.PayneSpecPredict(Cnnpath=...)): weights as for the spectral model; the label
// set must be the spectral model's.  cont == NULL removes it.
extern "C" int payne_ctx_set_continuum(payne_ctx* c, const payne_model_desc* cont) {
  if (!c) return PAYNE_E_INVALID;
  if (!c->has_model) return fail(c, PAYNE_E_INVALID, "context has no spectral model");
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (prev != c->device) (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (void* p : c->cont_owned) (void)hipFree(p);
  c->cont_owned.clear();
  c->has_cont = false;
  auto done = [&](int rc) { if (prev != c->device) (void)hipSetDevice(prev); return rc; };
  if (!cont) return done(PAYNE_OK);
  if (cont->n_layers < 3 || cont->n_layers > PAYNE_MAX_LAYERS) return done(fail(c, PAYNE_E_INVALID, "continuum.n_layers must be 3..8"));
  if (cont->n_labels != c->n_labels) return done(fail(c, PAYNE_E_INVALID, "continuum.n_labels != model.n_labels"));
  if (!cont->xmin || !cont->xmax || !cont->wavelength || cont->npix < 2)
    return done(fail(c, PAYNE_E_INVALID, "continuum.xmin/xmax/wavelength missing or npix < 2"));
  if (cont->layers[0].n_in != cont->n_labels || cont->layers[cont->n_layers - 1].n_out != cont->npix)
    return done(fail(c, PAYNE_E_INVALID, "continuum layer shapes do not match n_labels / npix"));
  int rc = PAYNE_OK, maxh = 0;
  for (int l = 0; l < cont->n_layers; ++l) {
    const payne_layer& L = cont->layers[l];
    if (!L.w || !L.b || L.n_in <= 0 || L.n_out <= 0) return done(fail(c, PAYNE_E_INVALID, "continuum layer with null weights or bad shape"));
    if (l > 0 && L.n_in != cont->layers[l - 1].n_out) return done(fail(c, PAYNE_E_INVALID, "continuum layer shapes do not chain"));
    c->clayers[l] = L;
    if (l > 0 && (L.n_in & 3)) {            // float4 tile loads need K % 4 == 0: zero-padded copy
      const int Kp = (L.n_in + 3) & ~3;
      float* wp = nullptr;
      if ((rc = dev_alloc(c, (size_t)L.n_out * Kp, &wp, c->cont_owned))) return done(rc);
      hipError_t he = hipMemcpy2D(wp, (size_t)Kp * 4, L.w, (size_t)L.n_in * 4, (size_t)L.n_in * 4, L.n_out, hipMemcpyDeviceToDevice);
      if (he != hipSuccess) return done(fail(c, PAYNE_E_HIP, std::string("hipMemcpy2D: ") + hipGetErrorString(he)));
      c->clayers[l].w = wp; c->clayers[l].n_in = Kp;
    }
    if (l + 1 < cont->n_layers) maxh = std::max(maxh, L.n_out);
  }
  c->cn_layers = cont->n_layers; c->cn_npix = cont->npix; c->cn_ld_hid = (maxh + 31) & ~31;
  for (int d = 0; d < cont->n_labels; ++d) { c->cxmin[d] = cont->xmin[d]; c->cxden[d] = cont->xmax[d] - cont->xmin[d]; }
  if ((rc = dev_alloc(c, (size_t)c->opts.b_max * c->cn_ld_hid, &c->chid[0], c->cont_owned))) return done(rc);
  if ((rc = dev_alloc(c, (size_t)c->opts.b_max * c->cn_ld_hid, &c->chid[1], c->cont_owned))) return done(rc);
  if ((rc = dev_alloc(c, (size_t)c->opts.b_max * cont->npix, &c->cont_raw, c->cont_owned, false))) return done(rc);
  // host tables: F_nu -> F_lambda factor (constants cancel in the median normalisation) and the np.interp map
  const int npc = cont->npix, npix = c->T.npix;
  std::vector<double> scale(npc), frac(npix);
  std::vector<int> idx(npix);
  const double* wc = cont->wavelength;
  for (int i = 1; i < npc; ++i)
    if (!(wc[i] > wc[i - 1])) return done(fail(c, PAYNE_E_INVALID, "continuum.wavelength must be strictly increasing"));
  const double lref = wc[npc / 2];
  for (int i = 0; i < npc; ++i) { const double r = lref / wc[i]; scale[i] = r * r; }
  for (int i = 0; i < npix; ++i) {
    const double x = c->H.lam[i];
    if (x < wc[0] || x > wc[npc - 1]) { idx[i] = -1; frac[i] = 0.0; continue; }   // left = right = NaN
    int k = (int)(std::upper_bound(wc, wc + npc, x) - wc) - 1;
    if (k > npc - 2) k = npc - 2;
    idx[i] = k;
    frac[i] = (x - wc[k]) / (wc[k + 1] - wc[k]);
  }
  if ((rc = upload(c, scale, &c->cont_scale, c->cont_owned))) return done(rc);
  if ((rc = upload(c, idx, &c->cont_idx, c->cont_owned))) return done(rc);
  if ((rc = upload(c, frac, &c->cont_frac, c->cont_owned))) return done(rc);
  c->has_cont = true;
  return done(PAYNE_OK);
}

// LSF vector for the instrumental broadening: `lsf[n]` = Gaussian dispersion (AA) at each pixel of the bound
// observed grid (getspec(inst_R=array, outwave=...), ystpred.py:248-269).  While set, theta's Inst_R column
// is ignored.  NULL removes it; re-binding the observed grid removes it too.
static int set_lsf_impl(payne_ctx* c, const double* lsf_wave, const double* lsf, int n) {
  if (!c) return PAYNE_E_INVALID;
  if (!c->has_model) return fail(c, PAYNE_E_INVALID, "context has no spectral model");
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (prev != c->device) (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  for (void* p : c->lsf_owned) (void)hipFree(p);
  c->lsf_owned.clear();
  c->has_lsf = false;
  auto done = [&](int rc) { if (prev != c->device) (void)hipSetDevice(prev); return rc; };
  if (!lsf) return done(PAYNE_OK);
  if (!c->obs_bound) return done(fail(c, PAYNE_E_INVALID, "bind the observed grid before its LSF vector"));
  if (!lsf_wave && n != c->T.nobs) return done(fail(c, PAYNE_E_INVALID, "the LSF vector must have one entry per observed pixel"));
  if (lsf_wave && n < 2) return done(fail(c, PAYNE_E_INVALID, "an LSF vector on its own wavelengths needs at least two entries"));
  for (int i = 0; i < n; ++i)
    if (!(lsf[i] > 0.0)) return done(fail(c, PAYNE_E_INVALID, "LSF dispersions must be positive"));
  if (lsf_wave)
    for (int i = 1; i < n; ++i)
      if (!(lsf_wave[i] > lsf_wave[i - 1])) return done(fail(c, PAYNE_E_INVALID, "the LSF vector's wavelengths must be strictly increasing"));
  std::vector<double> v(lsf, lsf + n);
  int rc;
  if ((rc = upload(c, v, &c->lsf, c->lsf_owned))) return done(rc);
  c->n_lsf = n;
  c->lsf_wave = c->d_obs_wave;
  if (lsf_wave) {
    std::vector<double> w(lsf_wave, lsf_wave + n);
    if ((rc = upload(c, w, &c->lsf_wave, c->lsf_owned))) return done(rc);
  }
  c->lsf_global = c->T.n1 > 8192 || (c->opts.variant & PAYNE_V_LSF_GLOBAL);
  c->lsf_chunk = c->lsf_global ? std::min(c->opts.b_max, 256) : c->opts.b_max;
  if ((rc = dev_alloc(c, (size_t)c->lsf_chunk * c->T.npix, &c->lsf_spec, c->lsf_owned, false))) return done(rc);
  if ((rc = dev_alloc(c, (size_t)c->lsf_chunk * (2 * (size_t)c->T.npix + c->T.n1), &c->lsf_ws, c->lsf_owned, false))) return done(rc);
  if (c->lsf_global) {
    if ((rc = dev_alloc(c, (size_t)c->lsf_chunk * 2 * fft_buf_floats(c->T.n1), &c->lsf_fws, c->lsf_owned, false))) return done(rc);
  } else {
    const size_t lds = (size_t)c->T.n1 * 8 + 2 * (size_t)fft_buf_floats(c->T.n1) * 4 + (256 + 8) * 8;
    hipError_t he = hipFuncSetAttribute(reinterpret_cast<const void*>(payne_lsf_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (he != hipSuccess) return done(fail(c, PAYNE_E_HIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString(he)));
  }
  c->has_lsf = true;
  return done(PAYNE_OK);
}
extern "C" int payne_ctx_set_lsf(payne_ctx* c, const double* lsf, int n) { return set_lsf_impl(c, nullptr, lsf, n); }
extern "C" int payne_ctx_set_lsf_on(payne_ctx* c, const double* lsf_wave, const double* lsf, int n) {
  if (c && lsf && !lsf_wave) return fail(c, PAYNE_E_INVALID, "lsf_wave is NULL");
  return set_lsf_impl(c, lsf_wave, lsf, n);
}

// ---- launches --------------------------------------------------------------
// Dynamic-LDS limits of the dense kernels, set on the context's device when the context is created (the attribute belongs
// to the function ON A DEVICE: a flag shared by every context would leave a second device of the same process without it).
static hipError_t set_dense_attributes() {
  hipError_t e = hipSuccess;
  auto set = [&](const void* f, size_t lds) {
    const hipError_t r = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = r;
  };
  set(reinterpret_cast<const void*>(payne_dense_kernel<64, 64, 32, true>), dense_lds_bytes<64, 64, 32>());
  set(reinterpret_cast<const void*>(payne_dense_kernel<64, 64, 32, false>), dense_lds_bytes<64, 64, 32>());
  set(reinterpret_cast<const void*>(payne_dense_dma_kernel<4, 32, 0, 4, true>), dm_lds_bytes<4, 32, 4>());
  set(reinterpret_cast<const void*>(payne_dense_dma_kernel<4, 32, 10, 4, true>), dm_lds_bytes<4, 32, 4>());
  set(reinterpret_cast<const void*>(payne_dense_dma_kernel<4, 32, 0, 3, false>), dm_lds_bytes<4, 32, 3>());
  set(reinterpret_cast<const void*>(payne_dense_dma_kernel<4, 64, 0, 3, true>), dm_lds_bytes<4, 64, 3>());
  set(reinterpret_cast<const void*>(payne_dense_dma_kernel<4, 64, 5, 3, true>), dm_lds_bytes<4, 64, 3>());
  set(reinterpret_cast<const void*>(payne_dense_dma3_kernel<0, 4, true>), d3_lds_bytes<4>());
  set(reinterpret_cast<const void*>(payne_dense_dma3_kernel<10, 4, true>), d3_lds_bytes<4>());
  set(reinterpret_cast<const void*>(payne_dense_dma3_kernel<0, 2, false>), d3_lds_bytes<2>());
  set(reinterpret_cast<const void*>(payne_dense_dma3f_kernel<10>), d3_lds_bytes<4>());
  set(reinterpret_cast<const void*>(payne_dense_dma2h_kernel<10, 32>), d2_lds_bytes<32>());
  set(reinterpret_cast<const void*>(payne_dense_dma2h_kernel<0, 32>), d2_lds_bytes<32>());
  set(reinterpret_cast<const void*>(payne_dense_dma2h_kernel<5, 64>), d2_lds_bytes<64>());
  set(reinterpret_cast<const void*>(payne_dense_dma2hh_kernel<5>), d2hh_lds_bytes<5>());
  set(reinterpret_cast<const void*>(payne_dense_big3_kernel<false>), b3_lds_bytes(false));
  set(reinterpret_cast<const void*>(payne_dense_big3_kernel<true>), b3_lds_bytes(true));
  set(reinterpret_cast<const void*>(payne_dense_chain_kernel), HK_LDS_BYTES);
  set(reinterpret_cast<const void*>(payne_dense_hidden_kernel<true, 4>), HK_LDS_BYTES);
  set(reinterpret_cast<const void*>(payne_dense_hidden_kernel<true, PAYNE_MAX_LABELS>), HK_LDS_BYTES);
  set(reinterpret_cast<const void*>(payne_dense_hidden_kernel<false, 4>), HK_LDS_BYTES);
  set(reinterpret_cast<const void*>(payne_dense_hidden_kernel<true, 4, 8>), HK_LDS_BYTES);
  set(reinterpret_cast<const void*>(payne_dense_hidden_kernel<true, PAYNE_MAX_LABELS, 8>), HK_LDS_BYTES);
  return e;
}

template <int BM, int BN, int BK, bool FUSE>
static void launch_dense(DenseParams& p, hipStream_t s) {
  p.grid_m = (p.B + BM - 1) / BM;
  p.grid_n = (p.N + BN - 1) / BN;
  constexpr size_t lds = dense_lds_bytes<BM, BN, BK>();
#ifdef PAYNE_STAMPS
  p.stamps = FUSE ? nullptr : g_dense_stamps;
#endif
  PAYNE_LAUNCH((payne_dense_kernel<BM, BN, BK, FUSE>), dim3(p.grid_m * p.grid_n), dim3(256), lds, s, p);
}

// Output layer, LDS-DMA form (64 x 128 tiles, 512 threads, BK-deep stages; NK: k-steps fixed at compile time or 0; NS: ring;
// PIPE: schedule -- see the kernel).
template <int BK, int NK, int NS, bool PIPE>
static void launch_out_dma_nk(DenseParams& p, hipStream_t s) {
  constexpr int WN = 4;
  constexpr size_t lds = dm_lds_bytes<WN, BK, NS>();
  PAYNE_LAUNCH((payne_dense_dma_kernel<WN, BK, NK, NS, PIPE>), dim3(p.grid_m * p.grid_n), dim3(128 * WN), lds, s, p);
}
template <int BK>
static void launch_out_dma(payne_ctx* c, DenseParams& p, hipStream_t s) {
  constexpr int WN = 4;
  p.k_real = p.K;                                          // the layer's own width: the padded tail is skipped
  p.W = c->w_out_pad; p.K = c->w_out_kp;                   // padded pitch; X's pitch (ld_hid) is a multiple of 32 too
  p.grid_m = (p.B + 63) / 64;
  p.grid_n = (p.N + 32 * WN - 1) / (32 * WN);
#ifdef PAYNE_STAMPS
  p.stamps = g_dense_stamps;
#endif
  constexpr int NKC = 320 / BK;                            // H = 300 -> 320 columns
  const bool fixed = p.K == NKC * BK && !(c->opts.variant & PAYNE_V_OUT_ROLLED);
  if constexpr (BK == 64) {                                // (variant: three 64-deep stages)
    if (fixed) launch_out_dma_nk<64, NKC, 3, true>(p, s); else launch_out_dma_nk<64, 0, 3, true>(p, s);
  } else if (p.grid_m * p.grid_n <= c->n_cu) {             // one tile per CU at most: the pipelined schedule, four stages
    if (fixed) launch_out_dma_nk<32, NKC, 4, true>(p, s); else launch_out_dma_nk<32, 0, 4, true>(p, s);
  } else {
    launch_out_dma_nk<32, 0, 3, false>(p, s);              // many tiles per CU: two workgroups per CU, the plain schedule
  }
}

// Output layer as six bf16 products (payne_dense_dma3_kernel): equal hidden widths.
static bool out_dma3_ok(const payne_ctx* c, int, int) {
  return c->w_out_p3 && c->hid_p3 && c->dma_ok && c->ld_hid >= c->w_out_kp && !(c->opts.variant & (PAYNE_V_OUT_F32 | PAYNE_V_OUT_GENERIC | PAYNE_V_OUT_BK64));
}
// ... as three fp16-pair products: many whole 128 x 256 tiles (payne_dense_big3_kernel<true>), else 64 x 128 tiles (payne_dense_dma2h_kernel)
static bool out_big_tiles(const payne_ctx* c, int B, int N, bool sel) {
  return B % B3_TM == 0 && N % B3_TN == 0 && (B / B3_TM) * (N / B3_TN) >= 2 * c->n_cu && !(c->opts.variant & PAYNE_V_OUT_SMALL_TILES) && !sel;
}
static bool out_dma2h_ok(const payne_ctx* c, int B, int N, bool sel = false) {
  // (whatever the batch: a candidate's rows do not depend on how many others share its batch -- many whole 128 x 256 tiles take
  //  payne_dense_big3_kernel<true>, everything else payne_dense_dma2h_kernel, one tile a workgroup, same products in the same order)
  (void)sel;
  // (payne_dense_dma2h_kernel addresses an operand plane by 32-bit byte offsets: rows x pitch x 2 bytes below 2^31)
  const unsigned long long plane_w = 2ull * (unsigned long long)std::max(N, c->T.n1) * (unsigned long long)c->w_out_kp;
  const unsigned long long plane_x = 2ull * (unsigned long long)c->opts.b_max * (unsigned long long)c->ld_hid;
  if (plane_w >= (1ull << 31) || plane_x >= (1ull << 31)) return false;
  return out_dma3_ok(c, B, N) && c->w_out_h2 && c->act_scale > 0.f && !(c->opts.variant & (PAYNE_V_OUT_BF16X3 | PAYNE_V_OUT_PLANES));
}
static void launch_out_dma2h(payne_ctx* c, DenseParams& p, hipStream_t s, bool freq) {
  p.k_real = p.K;
  p.K = c->w_out_kp;
  p.Wp = freq ? c->w_out_h2z : c->w_out_h2; p.plane_w = (size_t)p.N * c->w_out_kp;
  p.rscale = freq ? c->rscalez : c->rscale;
  if (freq) { p.bias = c->bias_z; p.bias_shift = 0.f; }
  p.Xp = c->hid_p3; p.plane_x = (size_t)c->opts.b_max * c->ld_hid; p.ldp = c->ld_hid;
  if (out_big_tiles(c, p.B, p.N, p.sel != nullptr)) {      // many whole 128 x 256 tiles per CU (C5): persistent workgroups
    p.grid_m = p.B / B3_TM; p.grid_n = p.N / B3_TN;
    PAYNE_LAUNCH(payne_dense_big3_kernel<true>, dim3(c->n_cu), dim3(512), b3_lds_bytes(true), s, p);
    return;
  }
  p.grid_m = (p.B + 63) / 64;
  p.grid_n = (p.N + 127) / 128;
#ifdef PAYNE_STAMPS
  p.stamps = g_dense_stamps;
#endif
  const dim3 grid(p.grid_m * p.grid_n), block(512);
  // 300-wide nets: five 64-deep steps when every tile has a compute unit to itself (C2: 144 KB of LDS a workgroup), ten 32-deep ones
  // otherwise; other widths, and PAYNE_V_OUT_ROLLED: 32-deep steps counted at run time.  Same products in the same order in all three.
  const bool k320 = p.K == 320 && !(c->opts.variant & PAYNE_V_OUT_ROLLED);
  const bool deep = (int)grid.x <= c->n_cu;
  // (... finished in two halves: the left half's rows leave under the right half's products -- payne_dense_dma2hh_kernel, same rows to the bit)
  if (k320 && deep && !(c->opts.variant & PAYNE_V_OUT_WHOLE_TILE))
    PAYNE_LAUNCH((payne_dense_dma2hh_kernel<5>), grid, block, d2hh_lds_bytes<5>(), s, PAYNE_D3_LEAD_ARGS(p), p);
  else if (k320 && deep) PAYNE_LAUNCH((payne_dense_dma2h_kernel<5, 64>), grid, block, d2_lds_bytes<64>(), s, PAYNE_D3_LEAD_ARGS(p), p);
  else if (k320) PAYNE_LAUNCH((payne_dense_dma2h_kernel<10, 32>), grid, block, d2_lds_bytes<32>(), s, PAYNE_D3_LEAD_ARGS(p), p);
  else PAYNE_LAUNCH((payne_dense_dma2h_kernel<0, 32>), grid, block, d2_lds_bytes<32>(), s, PAYNE_D3_LEAD_ARGS(p), p);
}
static void launch_out_dma3(payne_ctx* c, DenseParams& p, hipStream_t s, bool freq) {
  p.k_real = p.K;
  p.K = c->w_out_kp;
  p.Wp = freq ? c->w_out_p3z : c->w_out_p3; p.plane_w = (size_t)p.N * c->w_out_kp;
  if (freq) { p.bias = c->bias_z; p.bias_shift = 0.f; }
  p.Xp = c->hid_p3; p.plane_x = (size_t)c->opts.b_max * c->ld_hid; p.ldp = c->ld_hid;
  p.grid_m = (p.B + 63) / 64;
  p.grid_n = (p.N + 127) / 128;
#ifdef PAYNE_STAMPS
  p.stamps = g_dense_stamps;
#endif
  const dim3 block(512);
  if (out_big_tiles(c, p.B, p.N, p.sel != nullptr)) {
    // many whole 128 x 256 tiles per CU (C5): persistent workgroups, half the operand bytes per product
    p.grid_m = p.B / B3_TM; p.grid_n = p.N / B3_TN;
    PAYNE_LAUNCH(payne_dense_big3_kernel<false>, dim3(c->n_cu), block, b3_lds_bytes(false), s, p);
    return;
  }
  const dim3 grid(p.grid_m * p.grid_n);
  if ((int)grid.x > c->n_cu) PAYNE_LAUNCH((payne_dense_dma3_kernel<0, 2, false>), grid, block, d3_lds_bytes<2>(), s, PAYNE_D3_LEAD_ARGS(p), p);   // many tiles per CU
  else if (p.K == 320 && !(c->opts.variant & PAYNE_V_OUT_ROLLED)) {
    const float* wf = freq ? c->w_out_padz : c->w_out_pad;
    if (wf && !(c->opts.variant & PAYNE_V_OUT_PLANES) && (!p.sel || c->w_out_pad)) {
      // the weights as fp32 through the port, split into the planes on their way into LDS (28 instead of 36 KB a step)
      p.W_alt = c->w_out_pad;
      const unsigned short* keep = p.Wp;
      p.Wp = reinterpret_cast<const unsigned short*>(wf);
      PAYNE_LAUNCH((payne_dense_dma3f_kernel<10>), grid, block, d3_lds_bytes<4>(), s, PAYNE_D3_LEAD_ARGS(p), p);
      p.Wp = keep;
    }
    else PAYNE_LAUNCH((payne_dense_dma3_kernel<10, 4, true>), grid, block, d3_lds_bytes<4>(), s, PAYNE_D3_LEAD_ARGS(p), p);
  }
  else PAYNE_LAUNCH((payne_dense_dma3_kernel<0, 4, true>), grid, block, d3_lds_bytes<4>(), s, PAYNE_D3_LEAD_ARGS(p), p);
}

// The hidden-layer kernel's leading arguments carry two 16-bit values a dword (n_prep: 15 bits): what does not fit is refused
// by run_net before anything is launched (post_lead_fits is the post kernel's counterpart).
static bool hk_lead_fits(int B, int N, int K, int K0, int ldx, int ldwd, int ld_theta, int spec_K) {
  const long long n_gemm = (long long)((B + 31) / 32) * ((N + 31) / 32);
  return n_gemm <= 0xffff && (N + 31) / 32 <= 0xffff && (B + 255) / 256 <= 0x7fff && K <= 0xffff && K0 <= 0xffff && ldx <= 0xffff &&
         ldwd <= 0xffff && ld_theta <= 0xffff && (spec_K + kSpecChainsPerWg - 1) / kSpecChainsPerWg <= 0xffff;
}
template <bool FUSE>
static void launch_hidden(DenseParams& p, PrepArgs& pa, hipStream_t s, int n_cu = 256, int spec_K = 0, bool waves4 = false) {
  p.grid_m = (p.B + 31) / 32;
  p.grid_n = (p.N + 31) / 32;
  pa.n_gemm = p.grid_m * p.grid_n;
  if (!FUSE) { pa.out = nullptr; pa.sed_mags = nullptr; }
  pa.n_prep = pa.out ? (p.B + 255) / 256 : 0;
  pa.n_spec = pa.spec_walk ? (spec_K + kSpecChainsPerWg - 1) / kSpecChainsPerWg : 0;
  int n_sed = 0;
  if (pa.sed_mags) {
    // candidates per photometric tile: as few as keeps F x blocks within the workgroup slots the GEMM tiles, the record writers
    // and the proposals made ahead leave free -- TWO per CU (the launch's 79.9 KB of LDS): 16-candidate tiles at C3 (224 of them,
    // hidden launch 10.2 -> 9.4 us; a tile's time is mostly its fixed cost, 32-candidate tiles on one slot per CU measured 10.6)
    const int idle = 2 * n_cu - pa.n_gemm - pa.n_prep - pa.n_spec, nblk = idle >= pa.P.F ? idle / pa.P.F : 1;
    int cb = ((p.B + nblk - 1) / nblk + 15) & ~15;
    cb = cb < 16 ? 16 : (cb > kSedCandsMax ? kSedCandsMax : cb);
    pa.sed_cb = cb;
    n_sed = pa.P.F * ((p.B + cb - 1) / cb);
  }
#ifdef PAYNE_STAMPS
  p.stamps = FUSE ? g_hidden_stamps : nullptr;
#endif
  pa.n_sed = n_sed;
  // Eight waves for hk_tile_h2's tiles where the first layer is the expensive phase -- a sigmoid net's: forty activations a lane with four
  // waves (LinNet300: -1.2 us a step); a leaky-ReLU net's tile gains nothing (C2: 5.16 against 5.21 us) -- and no photometric tile rides
  // along: those are sized for two workgroups a CU.  PAYNE_HK_WAVES = 4 | 8 overrides (A/B runs).
  static const int force_waves = [] { const char* e = getenv("PAYNE_HK_WAVES"); return e ? atoi(e) : 0; }();
  const bool wide = FUSE && p.h2_tiles && n_sed == 0 && !waves4 && (force_waves ? force_waves == 8 : p.act0 == PAYNE_ACT_SIGMOID);
  const dim3 grid(pa.n_gemm + pa.n_prep + n_sed + pa.n_spec), block(wide ? 512 : 256);
  // the kernel's leading scalar parameters (preloaded into registers at wave start; two 16-bit values a dword)
  p.dma_tiles = (FUSE && p.Wd != nullptr) ? 1 : 0;
  const unsigned i0 = (unsigned)pa.n_spec | ((unsigned)pa.n_prep << 16) | ((unsigned)p.dma_tiles << 31), i1 = (unsigned)pa.n_gemm | ((unsigned)p.grid_n << 16);
  const unsigned i2 = FUSE ? ((unsigned)p.ld_theta | ((unsigned)p.n_labels << 16) | ((unsigned)(p.h2_tiles ? 1 : 0) << 31)) : ((unsigned)p.ldx | ((unsigned)p.ldwd << 16));
  const unsigned i4 = (unsigned)p.K | ((unsigned)(FUSE ? p.K0 : 0) << 16) | ((!FUSE && p.h2_tiles) ? 0x80000000u : 0u);
  const void* p0 = FUSE ? static_cast<const void*>(p.theta) : static_cast<const void*>(p.X);
  const float* p1 = FUSE ? p.W0 : p.Wd;
  if (!FUSE) PAYNE_LAUNCH((payne_dense_hidden_kernel<false, 4>), grid, block, HK_LDS_BYTES, s, p0, p1, p.b0, p.bias, i0, i1, i2, p.B, i4, p.N, p, pa);
  else if (p.n_labels <= 4) {
    if (wide) PAYNE_LAUNCH((payne_dense_hidden_kernel<true, 4, 8>), grid, block, HK_LDS_BYTES, s, p0, p1, p.b0, p.bias, i0, i1, i2, p.B, i4, p.N, p, pa);
    else PAYNE_LAUNCH((payne_dense_hidden_kernel<true, 4>), grid, block, HK_LDS_BYTES, s, p0, p1, p.b0, p.bias, i0, i1, i2, p.B, i4, p.N, p, pa);
  } else {
    if (wide) PAYNE_LAUNCH((payne_dense_hidden_kernel<true, PAYNE_MAX_LABELS, 8>), grid, block, HK_LDS_BYTES, s, p0, p1, p.b0, p.bias, i0, i1, i2, p.B, i4, p.N, p, pa);
    else PAYNE_LAUNCH((payne_dense_hidden_kernel<true, PAYNE_MAX_LABELS>), grid, block, HK_LDS_BYTES, s, p0, p1, p.b0, p.bias, i0, i1, i2, p.B, i4, p.N, p, pa);
  }
}

// ANN forward for the batch -> c->raw [B][npix] (shifted by -1)
// One network of the context: the spectral emulator or the continuum network.
struct NetRef {
  const payne_layer* layers; int n_layers; int n_labels; const double* xmin; const double* xden;
  float* const* hid; int ld_hid; float* out; int ld_out; float out_shift;
  bool spectral;                      // the spectral net owns the DMA / bf16x3 operand copies and the prep records
  bool freq = false;                  // spectral net: rows written in the frequency domain (c->w_out_p3z)
};
// `sed`: a joint likelihood's photometric nets ride in the first hidden-layer launch (sed_tile); *sed is cleared when they did.
static int run_net(payne_ctx* c, const NetRef& N, const double* theta, int B, double instr_factor, hipStream_t s, bool* sed = nullptr) {
  const int n = N.n_layers;
  for (int l = 1; l < n; ++l) {
    DenseParams p{};
    const payne_layer& L = N.layers[l];
    const bool last = (l == n - 1);
    p.W = L.w; p.K = L.n_in; p.bias = L.b; p.N = L.n_out; p.B = B; p.act = L.act;
    p.bias_shift = last ? N.out_shift : 0.f;
    p.Y = last ? N.out : N.hid[(l - 1) & 1];
    p.ldy = last ? N.ld_out : N.ld_hid;
    const bool use3 = N.spectral && out_dma3_ok(c, B, N.layers[n - 1].n_out);
    // (rows of the resampled grid: the wider of the two output layers sizes the grid)
    const int n_out_launch = (N.freq && c->freq_rs_now) ? std::max(c->T.n1, N.layers[n - 1].n_out) : N.layers[n - 1].n_out;
    const bool use2h = use3 && out_dma2h_ok(c, B, n_out_launch, N.freq && c->freq_rs_now) && (!N.freq || c->w_out_h2z);
    if (use3 && l == n - 2) {
      p.Yp = c->hid_p3; p.plane_y = (size_t)c->opts.b_max * c->ld_hid; p.ldp = c->ld_hid;
      if (use2h) { p.yp_half = 1; p.yp_scale = c->act_scale; }
    }
    // hidden layers past the second on fp16 pairs (hk_tile_h2x): layer l hands its activations to layer l + 1 as planes
    const bool chain = N.spectral && c->hid_h2[0] && c->w1_h2 && c->w1_rows == N.layers[1].n_out;
    if (chain && !last && l + 1 <= n - 2 && c->wl_h2[l + 1]) {
      p.Yp = c->hid_h2[(l - 1) & 1]; p.plane_y = (size_t)c->opts.b_max * 304; p.ldp = 304; p.yp_half = 1; p.yp_scale = c->hs[l];
    }
    ProfScope ps(c, s, last ? 0 : 3);
    if (l == 1) {
      const payne_layer& L0 = N.layers[0];
      p.theta = theta; p.ld_theta = c->ncols;
      p.W0 = L0.w; p.b0 = L0.b; p.n_labels = N.n_labels; p.act0 = L0.act; p.K0 = L0.n_out;
      for (int d = 0; d < N.n_labels; ++d) { p.xmin[d] = N.xmin[d]; p.xden[d] = N.xden[d]; }
      PrepArgs pa{};
      pa.T = c->T; pa.instr_factor = instr_factor;
      pa.out = (N.spectral && c->prep && c->obs_bound && !(c->opts.variant & PAYNE_V_NO_PREP)) ? c->prep : nullptr;
      if (N.freq && c->freq_rs_now) { pa.rot_flag = c->rot_flag; pa.rot_seq = c->rot_seq; }
      if (!last && sed && *sed && N.spectral && sed_tile_ok(c->P.H) && !(c->opts.variant & PAYNE_V_SED_OWN_LAUNCH)) {
        pa.P = c->P; pa.sed_mags = c->mags_ws; pa.sed_off = 8 + c->opts.npoly; pa.sed_photscale = c->opts.photscale;
        *sed = false;
      }
      // the walk's next proposals ride along when this batch's post kernel will run the chain step at its tail (run_post)
      const bool spec = !last && N.spectral && c->spec_walk && pa.out && c->lean_available && !c->big_ws;
      if (spec) {
        pa.spec_walk = static_cast<const WalkTail*>(c->spec_walk); pa.spec_w = *static_cast<const WalkState*>(c->spec_w);
        pa.spec_step = c->spec_step; c->spec_launched = true;
      }
      if (!last && N.spectral && c->w_hid_pad[1] && N.ld_hid >= HK_PITCH) { p.Wd = c->w_hid_pad[1]; p.ldwd = N.ld_hid; }
      if (!last && N.spectral && p.Wd && c->w1_h2 && c->w1_rows == p.N && p.K <= 304 &&
          (unsigned long long)B * (unsigned)c->ncols * 8ull < (1ull << 31)) {                // the second layer on fp16 pairs (hk_tile_h2: 32-bit byte offsets into theta)
        p.h2_tiles = 1; p.Wh = c->w1_h2; p.plane_wh = (size_t)c->w1_rows * 304; p.rs1 = c->rs1; p.a0_scale = c->a0_scale;
      }
      if (!last && !hk_lead_fits(p.B, p.N, p.K, p.K0, 0, p.ldwd, p.ld_theta, spec ? c->spec_K : 0))
        return fail(c, PAYNE_E_UNSUPPORTED, "batch x hidden width beyond what the hidden-layer kernel's packed arguments hold (65 535 tiles of 32 x 32)");
      if (last) launch_dense<64, 64, 32, true>(p, s);
      else { launch_hidden<true>(p, pa, s, c->n_cu, spec ? c->spec_K : 0, (c->opts.variant & PAYNE_V_HID_WAVES4) != 0); if (N.spectral) c->prep_valid = pa.out != nullptr; }
    } else if (!last && chain && use3 && l == 2 && n - 2 >= 3 && c->chain_flags && (c->opts.variant & PAYNE_V_HID_CHAIN) &&
               ((c->opts.b_max + 31) / 32) * ((N.layers[2].n_out + 31) / 32) <= 2 * c->n_cu && n - 3 <= kChainMax &&
               [&] { for (int q = 2; q <= n - 2; ++q) if (!c->wl_h2[q]) return false; return true; }()) {
      // layers 2 .. n - 2 in ONE launch: hand-offs inside a 32-candidate row block (payne_dense_chain_kernel; opt-in: a hop costs more
      // than a launch on this machine, NOTES R6.10)
      ChainParams cp{};
      cp.n = n - 3;
      for (int q = 2; q <= n - 2; ++q) {
        const payne_layer& Lq = N.layers[q];
        cp.Wh[q - 2] = c->wl_h2[q]; cp.rs[q - 2] = c->rsl[q]; cp.bias[q - 2] = Lq.b; cp.act[q - 2] = Lq.act;
        cp.out_scale[q - 2] = (q == n - 2) ? c->act_scale : c->hs[q];
      }
      cp.buf[0] = c->hid_h2[0]; cp.buf[1] = c->hid_h2[1]; cp.plane_x = (size_t)c->opts.b_max * 304;
      cp.first_in = 0;                                      // (layer 1 wrote hid_h2[0]: run_net above, (l - 1) & 1 at l = 1)
      cp.Yp_last = c->hid_p3; cp.plane_y_last = (size_t)c->opts.b_max * c->ld_hid; cp.ldp_last = c->ld_hid; cp.half_last = use2h ? 1 : 0;
      cp.flags = c->chain_flags; cp.grid_n = (N.layers[2].n_out + 31) / 32;
      cp.target0 = c->chain_calls * (unsigned long long)cp.grid_n;
      cp.B = B; cp.N = N.layers[2].n_out; cp.grid_m = (B + 31) / 32; cp.grid_m_max = (c->opts.b_max + 31) / 32;
      ++c->chain_calls;
      PAYNE_LAUNCH(payne_dense_chain_kernel, dim3(cp.grid_m_max * cp.grid_n), dim3(256), HK_LDS_BYTES, s, cp);
      l = n - 2;                                            // (the loop goes on with the output layer)
    } else {
      p.X = N.hid[(l - 2) & 1]; p.ldx = N.ld_hid;
      PrepArgs pa{};
      if (!last && N.spectral && c->w_hid_pad[l] && N.ld_hid >= HK_PITCH) { p.Wd = c->w_hid_pad[l]; p.ldwd = N.ld_hid; }
      if (chain && !last && c->wl_h2[l]) {                                   // both operands as planes
        p.h2_tiles = 1; p.Wh = c->wl_h2[l]; p.plane_wh = (size_t)N.layers[l].n_out * 304; p.rs1 = c->rsl[l];
        p.Xp = c->hid_h2[(l - 2) & 1]; p.plane_x = (size_t)c->opts.b_max * 304;
      }
      if (!last && !hk_lead_fits(p.B, p.N, p.K, 0, p.ldx, p.ldwd, 0, 0))
        return fail(c, PAYNE_E_UNSUPPORTED, "batch x hidden width beyond what the hidden-layer kernel's packed arguments hold (65 535 tiles of 32 x 32)");
      if (!last) launch_hidden<false>(p, pa, s);
      else if (use3) {
        if (N.freq && c->freq_rs_now) {                      // rows of the resampled grid; pixels if the batch's records say so
          p.sel = c->rot_flag; p.sel_seq = c->rot_seq;
          p.Wp_alt = c->w_out_p3; p.plane_w_alt = (size_t)p.N * c->w_out_kp; p.bias_alt = p.bias; p.bias_shift_alt = p.bias_shift;
          p.N_alt = p.N; p.ldy_alt = p.ldy;
          p.N = c->T.n1; p.ldy = c->T.n1;
          if (use2h) { p.Wp_alt = c->w_out_h2; p.rscale_alt = c->rscale; }
        }
        if (use2h) launch_out_dma2h(c, p, s, N.freq);
        else launch_out_dma3(c, p, s, N.freq);
      }
      else if (N.spectral && c->dma_ok && c->ld_hid >= c->w_out_kp && !(c->opts.variant & PAYNE_V_OUT_GENERIC)) {
        if ((c->w_out_kp % 64) == 0 && (c->opts.variant & PAYNE_V_OUT_BK64)) launch_out_dma<64>(c, p, s);
        else launch_out_dma<32>(c, p, s);
      }
      else launch_dense<64, 64, 32, false>(p, s);            // nets whose hidden widths differ, the continuum network
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, PAYNE_E_HIP, std::string("dense launch: ") + hipGetErrorString(e));
  return PAYNE_OK;
}

// ---- continuum network ------------------------------------------------------------------
// ystpred.py:191-209 for one candidate per workgroup: F_nu -> F_lambda, normalise by the NaN-ignoring
// median, interpolate onto the spectral ANN grid (NaN outside), multiply into the spectrum.
// The median is the middle of a bitonic sort in LDS (fp64 keys, NaN -> +inf); the spectrum row is stored
// shifted by -1, so (m C) - 1 = (m - 1) C + (C - 1).
__global__ void __launch_bounds__(256) payne_cont_kernel(const float* __restrict__ cont, int npc, int npc2,
                                                          const double* __restrict__ scale, const int* __restrict__ idx,
                                                          const double* __restrict__ frac, float* raw, int npix) {
  extern __shared__ __attribute__((aligned(16))) double cs[];       // [npc2] sort buffer (npc2 == 0: none, median by selection)
  __shared__ double med_s;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* row = cont + (size_t)b * npc;
  if (npc2 == 0) {                                                   // rows longer than LDS: radix selection, no size limit
    __shared__ int hist[256], cnt_s;
    __shared__ unsigned long long bc[2];
    const double med = nanmedian_select<256>([&](int i) { return (double)row[i] * scale[i]; }, npc, hist, bc, &cnt_s);
    float* r = raw + (size_t)b * npix;
    for (int i = tid; i < npix; i += 256) {
      const int k = idx[i];
      double C = __builtin_nan("");
      if (k >= 0) {
        const double q0 = (double)row[k] * scale[k], q1 = (double)row[k + 1] * scale[k + 1];
        C = (q0 + frac[i] * (q1 - q0)) / med;
      }
      const float m1 = r[i];
      r[i] = (float)((double)m1 * C + (C - 1.0));
    }
    return;
  }
  int nv = 0;
  for (int i = tid; i < npc2; i += 256) {
    double q = INFINITY;
    if (i < npc) { q = (double)row[i] * scale[i]; if (q != q) q = INFINITY; else ++nv; }
    cs[i] = q;
  }
  // number of non-NaN values: wave ballot sums through LDS would do; npc is small, use an LDS atomic
  __shared__ int nvalid;
  if (tid == 0) nvalid = 0;
  __syncthreads();
  atomicAdd(&nvalid, nv);
  for (int k = 2; k <= npc2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int i = tid; i < npc2; i += 256) {
        const int p = i ^ j;
        if (p > i) {
          const double a = cs[i], c2 = cs[p];
          const bool up = (i & k) == 0;
          if ((a > c2) == up) { cs[i] = c2; cs[p] = a; }
        }
      }
    }
  __syncthreads();
  if (tid == 0) {
    const int n = nvalid;                                            // np.nanmedian: mean of the two middle values
    med_s = n == 0 ? __builtin_nan("") : ((n & 1) ? cs[n >> 1] : 0.5 * (cs[(n >> 1) - 1] + cs[n >> 1]));
  }
  __syncthreads();
  const double med = med_s;
  float* r = raw + (size_t)b * npix;
  for (int i = tid; i < npix; i += 256) {
    const int k = idx[i];
    double C = __builtin_nan("");
    if (k >= 0) {
      const double q0 = (double)row[k] * scale[k], q1 = (double)row[k + 1] * scale[k + 1];
      C = (q0 + frac[i] * (q1 - q0)) / med;
    }
    const float m1 = r[i];
    r[i] = (float)((double)m1 * C + (C - 1.0));
  }
}

// `instr_factor`: what Inst_R is multiplied by (2.355 in the likelihood / genspec, 1 in getspec): the
// first-layer launch also writes the post kernel's per-candidate records (c->prep) for that factor.
// `pixels`: the caller reads the rows themselves (stage 0); otherwise they may be handed to the post kernel already transformed.
static int run_ann(payne_ctx* c, const double* theta, int B, double instr_factor, hipStream_t s, bool with_cont = true, bool* sed = nullptr,
                   bool pixels = false) {
  c->prep_valid = false;
  NetRef N{c->layers, c->n_layers, c->n_labels, c->xmin, c->xden, c->hid, c->ld_hid, c->raw, c->T.npix, kBase, true};
  // (the continuum multiplies pixel by pixel; out_dma3_ok: the launch that reads the restated weights is the one that runs)
  N.freq = c->freq_ok && !pixels && !(c->has_cont && with_cont) && out_dma3_ok(c, B, c->T.npix);
  // rows of a resampled grid need this batch's records (their writers report a candidate that does not rotate) and a first layer
  // fused into the hidden-layer launch (>= 3 layers), and the plain post kernel behind them
  if (N.freq && c->freq_rs && !(c->prep && c->obs_bound && !(c->opts.variant & PAYNE_V_NO_PREP) && c->n_layers >= 3 && !c->has_lsf)) N.freq = false;
  c->freq_rs_now = N.freq && c->freq_rs;
  if (c->freq_rs_now) ++c->rot_seq;
  c->raw_freq = N.freq; c->T.raw_freq = N.freq ? 1 : 0;
  int rc = run_net(c, N, theta, B, instr_factor, s, sed);
  if (rc || !c->has_cont || !with_cont) return rc;
  NetRef C{c->clayers, c->cn_layers, c->n_labels, c->cxmin, c->cxden, c->chid, c->cn_ld_hid, c->cont_raw, c->cn_npix, 0.f, false};
  if ((rc = run_net(c, C, theta, B, instr_factor, s))) return rc;
  int npc2 = 1;
  while (npc2 < c->cn_npix) npc2 <<= 1;
  if (npc2 > 4096 || (c->opts.variant & PAYNE_V_SELECT_MEDIAN)) npc2 = 0;     // median by selection instead of an LDS sort
  PAYNE_LAUNCH(payne_cont_kernel, dim3(B), dim3(256), (size_t)npc2 * 8, s, c->cont_raw, c->cn_npix, npc2, c->cont_scale,
                     c->cont_idx, c->cont_frac, c->raw, c->T.npix);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, PAYNE_E_HIP, std::string("continuum launch: ") + hipGetErrorString(e));
  return PAYNE_OK;
}

static int check_call(payne_ctx* c, const void* in, int B, const void* out) {
  if (!c) return PAYNE_E_INVALID;
  if (!in || !out) return fail(c, PAYNE_E_INVALID, "NULL input/output pointer");
  if (B <= 0) return fail(c, PAYNE_E_INVALID, "B must be > 0");
  if (B > c->opts.b_max) return fail(c, PAYNE_E_BATCH, "B exceeds opts.b_max");
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev != c->device) {
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return fail(c, PAYNE_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  }
  return PAYNE_OK;
}

static int run_sed(payne_ctx* c, const double* in, int ld, int mode, int B, double* mags, hipStream_t s) {
  {
    ProfScope ps(c, s, 2);
    PAYNE_LAUNCH(payne_sed_kernel, dim3(c->P.F, B), dim3(64), (size_t)c->P.H * 16, s, c->P, in, ld, mode,
                       8 + c->opts.npoly, c->opts.photscale, mags);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, PAYNE_E_HIP, std::string("sed launch: ") + hipGetErrorString(e));
  return PAYNE_OK;
}

static int run_post_lsf(payne_ctx* c, const double* theta, int B, int stage, float* out, int ld_out, double* lnl,
                        bool with_phot, hipStream_t s);

struct TailReq { const WalkTail* dev; int step, propose; bool done; const WalkState* spec; };   // spec: the walk (host copy) when proposals can be made ahead
static int run_post(payne_ctx* c, const double* theta, int B, double instr_factor, int stage, float* out, int ld_out,
                    double* lnl, bool with_phot, hipStream_t s, TailReq* tail = nullptr) {
  if (c->has_lsf && (stage < 0 || stage == 2 || stage == 3)) return run_post_lsf(c, theta, B, stage, out, ld_out, lnl, with_phot, s);
  PostArgs a{};
  a.theta = theta; a.ld_theta = c->ncols; a.instr_factor = instr_factor;
  a.raw = c->raw; a.ld_raw = c->T.npix;
  if (c->raw_freq && c->freq_rs_now) { a.ld_raw = c->T.n1; a.ld_raw_alt = c->T.npix; a.rot_flag = c->rot_flag; a.rot_seq = c->rot_seq; }
  a.out = out; a.ld_out = ld_out; a.out_stage = stage; a.lnl = lnl;
  if (with_phot) { a.mags = c->mags_ws; a.n_filters = c->P.F; a.obs_mag = c->obs_mag; a.obs_err = c->obs_err; }
  a.prep = c->prep_valid ? c->prep : nullptr;
  const bool lean = stage < 0 && !out && a.prep && !c->big_ws;
  if (tail && lean && c->lean_available) {
    a.tail = tail->dev; a.tail_step = tail->step; a.tail_propose = tail->propose; tail->done = true;
    a.tail_spec = (c->spec_launched && tail->propose && tail->spec) ? 1 : 0;
  }
  {
    ProfScope ps(c, s, 1);
    if (c->big_ws && c->big_chip) {
      const int grid = B < c->big_grid ? B : c->big_grid;
      PAYNE_LAUNCH(payne_post_chip_kernel, dim3(grid), dim3(kChipThreads), kChipLdsBytes, s, c->T, a, c->big_ws, B);
    } else if (c->big_ws && c->big_chip2) {
      const int pairs = (B + 1) / 2, grid = pairs < c->big_grid ? pairs : c->big_grid;
      PAYNE_LAUNCH(payne_post_chip2_kernel, dim3(grid), dim3(kChipThreads), kChipLdsBytes, s, c->T, a, c->big_ws, B);
    } else if (c->big_ws) {
      const int grid = B < c->big_grid ? B : c->big_grid;
      const int tiled = c->big_tiled ? 1 : 0;
      const size_t lds = (tiled & 1) ? 2 * (size_t)fft_tile_complex() * sizeof(c32) : 0;
      PAYNE_LAUNCH(payne_post_big_kernel, dim3(grid), dim3(kBigThreads), lds, s, c->T, a, c->big_ws, B, tiled);
    } else {
      if (!post_lead_fits(a.ld_raw, a.ld_theta, a.n_filters)) return fail(c, PAYNE_E_INVALID, "row pitch / theta columns / filters beyond what the post kernel's packed arguments hold");
      PAYNE_LAUNCH(lean ? c->post_fn_lean : c->post_fn, dim3(B), dim3(kPostThreads), c->post_lds, s, c->T.twf, a.raw, a.prep, a.theta,
                   a.rot_flag, a.mags, post_lead_ints(a.ld_raw, a.ld_theta, a.n_filters, c->T.raw_freq), (unsigned)a.rot_seq, c->T, a);
      c->last_kernel[1] = post_kernel_label((c->opts.variant & PAYNE_V_POST_GENERIC) ? 0 : c->T.n1, c->post_tw_lds, lean && c->lean_available);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, PAYNE_E_HIP, std::string("post launch: ") + hipGetErrorString(e));
  return PAYNE_OK;
}

// LSF path: spectra after rotational broadening (post kernel, stage 5) -> payne_lsf_kernel, a chunk of the batch at a time
static int run_post_lsf(payne_ctx* c, const double* theta, int B, int stage, float* out, int ld_out, double* lnl,
                        bool with_phot, hipStream_t s) {
  const bool had = c->has_lsf;
  for (int off = 0; off < B; off += c->lsf_chunk) {
    const int nb = std::min(c->lsf_chunk, B - off);
    const double* th = theta + (size_t)off * c->ncols;
    c->has_lsf = false;                                    // (the stage-5 pass below goes through run_post)
    const float* raw_all = c->raw;
    const CandState* prep_all = c->prep;
    c->raw = const_cast<float*>(raw_all) + (size_t)off * c->T.npix;              // this chunk's rows of the ANN output
    c->prep = const_cast<CandState*>(prep_all) + off;
    int rc = run_post(c, th, nb, 1.0, 5, c->lsf_spec, c->T.npix, nullptr, false, s);
    c->raw = const_cast<float*>(raw_all); c->prep = const_cast<CandState*>(prep_all);
    c->has_lsf = had;
    if (rc) return rc;
    LsfArgs a{};
    a.theta = th; a.ld_theta = c->ncols; a.spec = c->lsf_spec; a.ld_spec = c->T.npix;
    a.obs_wave = c->d_obs_wave; a.lsf = c->lsf; a.lsf_wave = c->lsf_wave; a.n_lsf = c->n_lsf;
    a.ws = c->lsf_ws; a.ws_stride = 2 * (size_t)c->T.npix + c->T.n1;
    a.fws = c->lsf_fws; a.fws_stride = 2 * (size_t)fft_buf_floats(c->T.n1);
    a.out = out ? out + (size_t)off * ld_out : nullptr; a.ld_out = ld_out; a.out_stage = stage; a.lnl = lnl ? lnl + off : nullptr;
    if (with_phot) { a.mags = c->mags_ws + (size_t)off * c->P.F; a.n_filters = c->P.F; a.obs_mag = c->obs_mag; a.obs_err = c->obs_err; }
    {
      ProfScope ps(c, s, 1);
      if (c->lsf_global) {
        PAYNE_LAUNCH(payne_lsf_kernel<true>, dim3(nb), dim3(256), (size_t)(256 + 8) * 8, s, c->T, a);
      } else {
        const size_t lds = (size_t)c->T.n1 * 8 + 2 * (size_t)fft_buf_floats(c->T.n1) * 4 + (256 + 8) * 8;
        PAYNE_LAUNCH(payne_lsf_kernel<false>, dim3(nb), dim3(256), lds, s, c->T, a);
      }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, PAYNE_E_HIP, std::string("lsf launch: ") + hipGetErrorString(e));
  }
  return PAYNE_OK;
}

static int lnlike_impl(payne_ctx* c, const double* theta, int B, double* lnl, void* stream, TailReq* tail);
extern "C" int payne_lnlike_batch(payne_ctx* c, const double* theta, int B, double* lnl, void* stream) {
  return lnlike_impl(c, theta, B, lnl, stream, nullptr);
}
static int lnlike_impl(payne_ctx* c, const double* theta, int B, double* lnl, void* stream, TailReq* tail) {
  int rc = check_call(c, theta, B, lnl);
  if (rc) return rc;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (c->has_phot && !c->has_obs_phot) return fail(c, PAYNE_E_INVALID, "photometric model without observed magnitudes");
  if (c->has_model) {
    if (!c->obs_bound || !c->T.obs_f1) return fail(c, PAYNE_E_INVALID, "no observed spectrum (flux, eflux) bound");
  }
  bool sed_pending = c->has_phot;                           // the photometric nets: in the hidden-layer launch when there is one
  TailReq* const tl = (c->has_lsf || (c->opts.variant & PAYNE_V_NO_WALK_TAIL)) ? nullptr : tail;
  c->spec_launched = false;
  if (tl && tl->propose && tl->spec && !(c->opts.variant & PAYNE_V_NO_WALK_SPEC)) { c->spec_walk = tl->dev; c->spec_w = tl->spec; c->spec_step = tl->step; c->spec_K = B; }
  if (c->has_model) rc = run_ann(c, theta, B, 2.355, s, true, &sed_pending);
  c->spec_walk = nullptr;
  if (rc) return rc;
  if (sed_pending && (rc = run_sed(c, theta, c->ncols, 1, B, c->mags_ws, s))) return rc;
  if (c->has_model) return run_post(c, theta, B, 2.355, -1, nullptr, 0, lnl, c->has_phot, s, tl);
  hipLaunchKernelGGL(payne_photonly_kernel, dim3((B + 127) / 128), dim3(128), 0, s, c->mags_ws, c->obs_mag, c->obs_err, c->P.F, B, lnl);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, PAYNE_E_HIP, std::string("photonly launch: ") + hipGetErrorString(e));
  return PAYNE_OK;
}

extern "C" int payne_predict_batch(payne_ctx* c, const double* theta, int B, int stage, unsigned flags, float* out,
                                   int ld_out, void* stream) {
  int rc = check_call(c, theta, B, out);
  if (rc) return rc;
  if (!c->has_model) return fail(c, PAYNE_E_INVALID, "context has no spectral model");
  if (stage < 0 || stage > 4) return fail(c, PAYNE_E_INVALID, "stage must be 0..4");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (stage == PAYNE_STAGE_CONT) {                       // predictcont: the continuum network's own output
    if (!c->has_cont) return fail(c, PAYNE_E_INVALID, "no continuum network bound");
    if (ld_out < c->cn_npix) return fail(c, PAYNE_E_INVALID, "ld_out too small");
    NetRef C{c->clayers, c->cn_layers, c->n_labels, c->cxmin, c->cxden, c->chid, c->cn_ld_hid, c->cont_raw, c->cn_npix, 0.f, false};
    if ((rc = run_net(c, C, theta, B, 1.0, s))) return rc;
    hipError_t he = hipMemcpy2DAsync(out, (size_t)ld_out * 4, c->cont_raw, (size_t)c->cn_npix * 4, (size_t)c->cn_npix * 4, B,
                                     hipMemcpyDeviceToDevice, s);
    if (he != hipSuccess) return fail(c, PAYNE_E_HIP, std::string("hipMemcpy2DAsync: ") + hipGetErrorString(he));
    return PAYNE_OK;
  }
  if (stage >= 2 && !c->obs_bound) return fail(c, PAYNE_E_INVALID, "no observed grid bound");
  if (ld_out < (stage >= 2 ? c->T.nobs : c->T.npix)) return fail(c, PAYNE_E_INVALID, "ld_out too small");
  if ((rc = run_ann(c, theta, B, (flags & PAYNE_F_FWHM_R) ? 2.355 : 1.0, s, stage != 0, nullptr, stage == 0))) return rc;   // stage 0 = predictspec: no continuum
  return run_post(c, theta, B, (flags & PAYNE_F_FWHM_R) ? 2.355 : 1.0, stage, out, ld_out, nullptr, false, s);
}

// smoothspec on caller-supplied spectra (PayneSpecPredict.smoothspec, ystpred.py:279-281 -> utils.smoothing.smoothspec):
// the same stages as payne_predict_batch with the ANN forward pass replaced by `spectra` (full flux on the context's
// model grid).  stage 1: rotational broadening on the model grid; stage 2/3: ... and instrumental broadening onto
// the bound observed grid.
__global__ void payne_shift_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ out, int npix, int B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)B * npix) { const size_t b = i / npix, p = i - b * npix; out[i] = in[b * ld_in + p] - kBase; }
}
extern "C" int payne_smooth_batch(payne_ctx* c, const float* spectra, int ld_spec, const double* theta, int B, int stage,
                                  unsigned flags, float* out, int ld_out, void* stream) {
  int rc = check_call(c, theta, B, out);
  if (rc) return rc;
  if (!spectra) return fail(c, PAYNE_E_INVALID, "spectra is NULL");
  if (!c->has_model) return fail(c, PAYNE_E_INVALID, "context has no spectral model");
  if (stage < 1 || stage > 4) return fail(c, PAYNE_E_INVALID, "stage must be 1..4");
  if (stage == PAYNE_SMOOTH_VSINI_TO_OBS && c->has_lsf) return fail(c, PAYNE_E_INVALID, "stage 4 with an LSF vector bound");
  if (ld_spec < c->T.npix) return fail(c, PAYNE_E_INVALID, "ld_spec too small");
  if (stage >= 2 && !c->obs_bound) return fail(c, PAYNE_E_INVALID, "no observed grid bound");
  if (ld_out < (stage >= 2 ? c->T.nobs : c->T.npix)) return fail(c, PAYNE_E_INVALID, "ld_out too small");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t n = (size_t)B * c->T.npix;
  hipLaunchKernelGGL(payne_shift_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, spectra, ld_spec, c->raw, c->T.npix, B);
  c->prep_valid = false;                                   // no dense launch wrote records for these rows
  c->raw_freq = false; c->T.raw_freq = 0;                  // ... and they are pixels
  // stage 1 here is smoothspec('vsini') itself: getspec's edge rule (ystpred.py:223-224) is not part of it
  // stage 4: ... interpolated from the stage's own resampled grid onto the bound observed grid (smoothspec('vsini', outwave=...))
  return run_post(c, theta, B, (flags & PAYNE_F_FWHM_R) ? 2.355 : 1.0, stage == 1 ? 6 : (stage == PAYNE_SMOOTH_VSINI_TO_OBS ? 7 : stage), out, ld_out,
                  nullptr, false, s);
}

extern "C" int payne_sed_batch(payne_ctx* c, const double* pars, int B, double* mags, void* stream) {
  int rc = check_call(c, pars, B, mags);
  if (rc) return rc;
  if (!c->has_phot) return fail(c, PAYNE_E_INVALID, "context has no photometric model");
  return run_sed(c, pars, 9, 0, B, mags, reinterpret_cast<hipStream_t>(stream));
}


#include "sampler_kernels.hpp"

struct payne_sampler {
  payne_ctx* ctx = nullptr;
  SamplerDev sd{};
  int k_max = 0;
  double *u_prop = nullptr, *v_prop = nullptr, *lnprior = nullptr, *lnl = nullptr, *rows = nullptr, *axes = nullptr;
  int* inside = nullptr;
  int* ell = nullptr;                     // per-chain ellipsoid index of the walk in progress
  int* nredraw = nullptr;                 // per-chain count of proposals redrawn because they left the unit cube
  double* spec = nullptr;                 // [k_max][2][kSpecStride] the next step's proposals made ahead (null: more dimensions / columns than a record holds)
  // staging of payne_ns_rwalk_queue: chains (u | v | lnprob) and counters (nacc | ncall | nredraw), device + pinned host
  double *q_dev = nullptr, *q_host = nullptr;
  // the queue's transfers as KERNELS on mapped host memory (q_host_dev: the device's address of q_host) and its completion as a word
  // the last of them writes into it (q_flag, behind the staging block; q_seq: the value the queue in flight will write) -- the
  // host reads that word instead of synchronising the stream: hipMemcpyAsync's own enqueue (17 us) and the wake-up out of
  // hipStreamSynchronize were most of the GPU's idle time between two queues (NOTES R4.16).  Null: the block has no device address.
  double* q_host_dev = nullptr; volatile unsigned long long* q_flag = nullptr; unsigned long long* q_flag_dev = nullptr; unsigned long long q_seq = 0;
  unsigned* q_arrivals = nullptr;         // device word: workgroups of the results' transfer that have finished (wraps to zero)
  // The queue's TURN on the device (payne_ns_queue_dev_*): the live set lives there (two copies, written in turn), the turn kernel
  // merges a queue's proposals into it, adapts the scale, raises the threshold and draws the next start points -- the next queue
  // is ENQUEUED before the current one has finished, and the GPU goes from one to the other without the host (NOTES R4.18).
  double *lv_u[2] = {nullptr, nullptr}, *lv_v[2] = {nullptr, nullptr}, *lv_l[2] = {nullptr, nullptr};
  int lv_n = 0, lv_cur = 0;
  bool lv_sorted = false;    // the current live set is the output of a merging turn: best first
  bool turn_lds_ok = false;  // this device lets payne_ns_turn_kernel have kTurnRowsLdsMax bytes of dynamic LDS (asked in payne_ns_queue_dev_init)
  double* dyn = nullptr;                  // device: {scale, loglstar}
  double* dq_host[2] = {nullptr, nullptr}; double* dq_host_dev[2] = {nullptr, nullptr};      // two mapped result blocks (+ flag word each)
  unsigned long long dq_seq[2] = {0, 0};
  double* dax_host[2] = {nullptr, nullptr}; double* dax_host_dev[2] = {nullptr, nullptr};    // two mapped blocks for the bound
  int dq_launched = 0, dq_collected = 0, dq_exported = 0, dq_K = 0, dq_n_ell = 0, dax_n = 0;   // (exported: queues whose results' transfer is enqueued)
  std::vector<int> pk_src, pk_heap;       // payne_ns_rwalk_queue_turn: the live set its peek predicts, by index (payne_ns::peek_index)
  std::vector<double> pk_l;
  WalkTail* tail_dev = nullptr;           // the walk in progress as the post kernel's tail reads it (written by the launch that opens the walk)
  WalkState walk{};                       // the walk in progress
  bool queue_open = false; int queue_K = 0; void* queue_stream = nullptr;   // payne_ns_rwalk_queue_begin .. _end
  bool tail_done = false;                 // the last likelihood batch ran the next step at its tail
  long long n_tail = 0, n_own = 0;        // chain steps at the post kernel's tail / as launches of their own
  std::vector<void*> owned;
  // a walk in progress (payne_rwalk_begin / payne_rwalk_step)
  struct { double *u, *v, *lnprob; int K, walks; double scale, loglstar; unsigned long long seed; int *nacc, *ncall; void* stream; bool open; bool multi; int* nredraw; } run{};
};

// doubles of the queue's staging block (device and pinned host): chains (u | v | lnprob) and the walk's three counters (what
// comes back, in ONE transfer), then the ellipsoids' axes, centres and inverse axes (they travel up with the chains in one
// transfer; the counters' slots go along unused), then (device only) the chains' ellipsoid indices
static size_t q_doubles(size_t K, size_t nd) { return K * (2 * nd + 1) + (3 * K + 1) / 2 + (size_t)PAYNE_MAX_ELL * (2 * nd * nd + nd) + (K + 1) / 2; }

extern "C" void payne_sampler_destroy(payne_sampler* s) {
  if (!s) return;
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(s->ctx->device);
  for (void* p : s->owned) (void)hipFree(p);
  if (s->q_host) (void)hipHostFree(s->q_host);
  for (int b = 0; b < 2; ++b) { if (s->dq_host[b]) (void)hipHostFree(s->dq_host[b]); if (s->dax_host[b]) (void)hipHostFree(s->dax_host[b]); }
  (void)hipSetDevice(prev);
  delete s;
}

extern "C" int payne_sampler_create(payne_ctx* c, const payne_sampler_desc* d, int k_max, payne_sampler** out) {
  if (!c || !out) return PAYNE_E_INVALID;
  *out = nullptr;
  if (!d || d->ndim <= 0 || d->ndim > PAYNE_MAX_DIM) return fail(c, PAYNE_E_INVALID, "sampler.ndim out of range");
  if (d->n_fixed < 0 || d->n_fixed > PAYNE_MAX_FIXED) return fail(c, PAYNE_E_INVALID, "sampler.n_fixed out of range");
  if (k_max <= 0 || k_max > c->opts.b_max) return fail(c, PAYNE_E_INVALID, "sampler.k_max must be in 1..opts.b_max");
  for (int i = 0; i < d->ndim; ++i)
    if (d->dims[i].theta_col >= c->ncols || d->dims[i].kind < 0 || d->dims[i].kind > PAYNE_PRIOR_TABLE)
      return fail(c, PAYNE_E_INVALID, "sampler dimension with bad theta_col / kind");
  for (int i = 0; i < d->n_fixed; ++i)
    if (d->fixed_col[i] < 0 || d->fixed_col[i] >= c->ncols) return fail(c, PAYNE_E_INVALID, "fixed parameter with bad column");
  {
    const payne_adv_priors& a = d->adv;
    int ntab = 0;
    for (int i = 0; i < d->ndim; ++i) ntab += d->dims[i].kind == PAYNE_PRIOR_TABLE;
    if (ntab > 1 || (ntab == 1 && (!a.tab_cdf || !a.tab_val || a.tab_n < 2))) return fail(c, PAYNE_E_INVALID, "PAYNE_PRIOR_TABLE needs adv.tab_* (one table per sampler)");
    if (a.dim_logg >= d->ndim || a.dim_logr >= d->ndim || a.dim_vrot >= d->ndim || a.plx_dim >= d->ndim)
      return fail(c, PAYNE_E_INVALID, "adv.dim_* out of range");
  }
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev != c->device) (void)hipSetDevice(c->device);
  payne_sampler* s = new payne_sampler();
  s->ctx = c; s->k_max = k_max;
  s->sd.ndim = d->ndim; s->sd.ncols = c->ncols; s->sd.nfixed = d->n_fixed;
  for (int i = 0; i < d->ndim; ++i) s->sd.dims[i] = d->dims[i];
  for (int i = 0; i < d->n_fixed; ++i) { s->sd.fixed_col[i] = d->fixed_col[i]; s->sd.fixed_val[i] = d->fixed_val[i]; }
  for (int k = 0; k < kMaxThetaCols; ++k) { s->sd.col_src[k] = -1; s->sd.col_val[k] = std::nan(""); }
  for (int i = 0; i < d->n_fixed; ++i) s->sd.col_val[d->fixed_col[i]] = d->fixed_val[i];
  for (int i = 0; i < d->ndim; ++i) if (d->dims[i].theta_col >= 0) s->sd.col_src[d->dims[i].theta_col] = i;
  s->sd.adv = d->adv;
  s->sd.adv.tab_cdf = nullptr; s->sd.adv.tab_val = nullptr;
  auto alloc = [&](size_t bytes, void** p) -> int {
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) return fail(c, PAYNE_E_HIP, std::string("hipMalloc(sampler): ") + hipGetErrorString(e));
    s->owned.push_back(*p);
    return PAYNE_OK;
  };
  const size_t K = (size_t)k_max, nd = (size_t)d->ndim;
  int rc;
  if ((rc = alloc(K * nd * 8, (void**)&s->u_prop)) || (rc = alloc(K * nd * 8, (void**)&s->v_prop)) ||
      (rc = alloc(K * 8, (void**)&s->lnprior)) || (rc = alloc(K * 8, (void**)&s->lnl)) ||
      (rc = alloc(K * c->ncols * 8, (void**)&s->rows)) || (rc = alloc((size_t)PAYNE_MAX_ELL * nd * nd * 8, (void**)&s->axes)) ||
      (rc = alloc(K * 4, (void**)&s->inside)) || (rc = alloc(K * 4, (void**)&s->ell)) || (rc = alloc(K * 4, (void**)&s->nredraw)) ||
      (rc = alloc(std::max<size_t>(q_doubles(K, nd), 2 * PAYNE_MAX_DIM) * 8, (void**)&s->q_dev)) ||
      (rc = alloc(sizeof(WalkTail), (void**)&s->tail_dev)) ||
      (spec_fits(d->ndim, c->ncols) && (rc = alloc(K * 2 * kSpecStride * 8, (void**)&s->spec)))) {
    payne_sampler_destroy(s);
    return rc;
  }
  if (d->adv.tab_cdf && d->adv.tab_val && d->adv.tab_n >= 2) {          // the tabulated inverse CDF (PAYNE_PRIOR_TABLE)
    double *tc = nullptr, *tv = nullptr;
    const size_t nb = (size_t)d->adv.tab_n * 8;
    if ((rc = alloc(nb, (void**)&tc)) || (rc = alloc(nb, (void**)&tv))) { payne_sampler_destroy(s); return rc; }
    s->sd.adv.tab_cdf = tc; s->sd.adv.tab_val = tv;                     // (owned by the sampler from here on)
    if (hipMemcpy(tc, d->adv.tab_cdf, nb, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(tv, d->adv.tab_val, nb, hipMemcpyHostToDevice) != hipSuccess) {
      payne_sampler_destroy(s);
      return fail(c, PAYNE_E_HIP, "upload of the tabulated prior");
    }
  }
  {   // the transforms' per-dimension constants, computed by the device functions the steps use (q_dev: staging, free here)
    double qh[2 * PAYNE_MAX_DIM];
    hipLaunchKernelGGL(payne_prior_cache_kernel, dim3(1), dim3(64), 0, nullptr, s->sd, s->q_dev);
    if (hipGetLastError() != hipSuccess || hipMemcpy(qh, s->q_dev, sizeof(qh), hipMemcpyDeviceToHost) != hipSuccess) {
      payne_sampler_destroy(s);
      return fail(c, PAYNE_E_HIP, "prior cache kernel");
    }
    for (int i = 0; i < PAYNE_MAX_DIM; ++i) { s->sd.q0[i] = qh[i]; s->sd.q1[i] = qh[PAYNE_MAX_DIM + i]; }
  }
  // the descriptor as the post kernel's tail reads it: it never changes after this point, so it is published here, once (the launch
  // that opens a walk publishes the walk's own record only -- it used to copy these 4 KB too, by one thread, in front of every queue)
  if (hipMemcpy(&s->tail_dev->sd, &s->sd, sizeof(SamplerDev), hipMemcpyHostToDevice) != hipSuccess) {
    payne_sampler_destroy(s);
    return fail(c, PAYNE_E_HIP, "upload of the sampler descriptor");
  }
  (void)hipMemset(s->inside, 0, K * 4);
  if (hipHostMalloc((void**)&s->q_host, (q_doubles(K, nd) + 8) * 8, hipHostMallocMapped) != hipSuccess) {
    payne_sampler_destroy(s);
    return fail(c, PAYNE_E_HIP, "hipHostMalloc(sampler staging)");
  }
  {                                                          // (no device address for the block: hipMemcpyAsync + hipStreamSynchronize)
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, s->q_host, 0) == hipSuccess && dp) {
      s->q_host_dev = static_cast<double*>(dp);
      s->q_flag = reinterpret_cast<volatile unsigned long long*>(s->q_host + q_doubles(K, nd));
      s->q_flag_dev = reinterpret_cast<unsigned long long*>(s->q_host_dev + q_doubles(K, nd));
      *s->q_flag = 0ull;
      void* ap = nullptr;
      if (hipMalloc(&ap, 8) == hipSuccess && hipMemset(ap, 0, 8) == hipSuccess) { s->owned.push_back(ap); s->q_arrivals = static_cast<unsigned*>(ap); }
    }
  }
  *out = s;
  return PAYNE_OK;
}

static int sampler_check(payne_sampler* s, const void* a, int K, const void* b) {
  if (!s) return PAYNE_E_INVALID;
  payne_ctx* c = s->ctx;
  if (!a || !b) return fail(c, PAYNE_E_INVALID, "NULL input/output pointer");
  if (K <= 0 || K > s->k_max) return fail(c, PAYNE_E_BATCH, "K exceeds the sampler's k_max");
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev != c->device) (void)hipSetDevice(c->device);
  return PAYNE_OK;
}

extern "C" int payne_prior_transform_batch(payne_sampler* s, const double* u, int K, double* v, void* stream) {
  int rc = sampler_check(s, u, K, v);
  if (rc) return rc;
  hipLaunchKernelGGL(payne_prior_kernel, dim3((K + 127) / 128), dim3(128), 0, reinterpret_cast<hipStream_t>(stream), s->sd, u, K,
                     v, (double*)nullptr, (double*)nullptr, 0);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(s->ctx, PAYNE_E_HIP, std::string("prior launch: ") + hipGetErrorString(e));
  return PAYNE_OK;
}

extern "C" int payne_lnprob_u_batch(payne_sampler* s, const double* u, int K, double* v, double* lnprob, void* stream) {
  int rc = sampler_check(s, u, K, v);
  if (rc) return rc;
  if (!lnprob) return fail(s->ctx, PAYNE_E_INVALID, "lnprob is NULL");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(payne_prior_kernel, dim3((K + 127) / 128), dim3(128), 0, st, s->sd, u, K, v, s->lnprior, s->rows, 1);
  if ((rc = payne_lnlike_batch(s->ctx, s->rows, K, s->lnl, stream))) return rc;
  hipLaunchKernelGGL(payne_lnprob_kernel, dim3((K + 127) / 128), dim3(128), 0, st, s->lnprior, s->lnl, K, lnprob);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(s->ctx, PAYNE_E_HIP, std::string("lnprob launch: ") + hipGetErrorString(e));
  return PAYNE_OK;
}

// The walk in two parts, so that a caller can interleave the steps of several samplers (one context and
// one HIP stream each) from one host thread: two independent batches in flight fill the idle time a single
// chain of dependent launches leaves (13.6 M against 10.6 M evaluations/s at 512 x 4096 pixels).
// (the walk's counters -- nacc, ncall, nredraw -- are zeroed by its first step: payne_rwalk_kernel with settle = 0)
static void rwalk_begin_impl(payne_sampler* s, double* u, double* v, double* lnprob, int K, const double* axes_dev, const int* ell_dev,
                             double scale, double loglstar, int walks, unsigned long long seed, int* nacc, int* ncall, int* nredraw,
                             void* stream) {
  s->run = {u, v, lnprob, K, walks, scale, loglstar, seed, nacc, ncall, stream, true, ell_dev != nullptr, nredraw};
  s->walk = WalkState{u, v, lnprob, nacc, ncall, s->u_prop, s->v_prop, s->lnprior, s->inside, s->rows, axes_dev,
                      ell_dev, nredraw, scale, loglstar, seed, K,
                      s->sd.ndim, s->sd.ncols, (s->sd.adv.imf || s->sd.adv.vrot || s->sd.adv.plx_dim >= 0) ? 1 : 0, s->spec,
                      nullptr, nullptr, nullptr, 0, nullptr};
  s->tail_done = false;
}
extern "C" int payne_rwalk_begin_ell(payne_sampler* s, double* u, double* v, double* lnprob, int K, const double* axes,
                                     int n_ell, const int* ell, double scale, double loglstar, int walks,
                                     unsigned long long seed, int* nacc, int* ncall, void* stream) {
  int rc = sampler_check(s, u, K, v);
  if (rc) return rc;
  if (!lnprob || !axes || !nacc || !ncall || walks <= 0) return fail(s->ctx, PAYNE_E_INVALID, "bad rwalk arguments");
  if (n_ell < 1 || n_ell > PAYNE_MAX_ELL || (n_ell > 1 && !ell)) return fail(s->ctx, PAYNE_E_INVALID, "bad ellipsoid list");
  if (ell)
    for (int i = 0; i < K; ++i)
      if (ell[i] < 0 || ell[i] >= n_ell) return fail(s->ctx, PAYNE_E_INVALID, "ellipsoid index out of range");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nd = s->sd.ndim;
  HIPCHK(s->ctx, hipMemcpyAsync(s->axes, axes, (size_t)n_ell * nd * nd * 8, hipMemcpyHostToDevice, st));
  if (ell) HIPCHK(s->ctx, hipMemcpyAsync(s->ell, ell, (size_t)K * 4, hipMemcpyHostToDevice, st));
  rwalk_begin_impl(s, u, v, lnprob, K, s->axes, ell ? s->ell : (const int*)nullptr, scale, loglstar, walks, seed, nacc, ncall, s->nredraw, stream);
  return PAYNE_OK;
}
extern "C" int payne_rwalk_begin(payne_sampler* s, double* u, double* v, double* lnprob, int K, const double* axes,
                                 double scale, double loglstar, int walks, unsigned long long seed, int* nacc, int* ncall,
                                 void* stream) {
  return payne_rwalk_begin_ell(s, u, v, lnprob, K, axes, 1, nullptr, scale, loglstar, walks, seed, nacc, ncall, stream);
}
// step w = 0 .. walks: settle proposal w-1, draw proposal w and evaluate it (the last step only settles)
extern "C" int payne_rwalk_step(payne_sampler* s, int w) {
  if (!s || !s->ctx) return PAYNE_E_INVALID;
  if (!s->run.open || w < 0 || w > s->run.walks) return fail(s->ctx, PAYNE_E_INVALID, "payne_rwalk_step outside a walk");
  const auto& r = s->run;
  hipStream_t st = reinterpret_cast<hipStream_t>(r.stream);
  // step w: settle proposal w-1, draw proposal w -- unless the previous likelihood batch already did it at its tail
  if (s->tail_done) s->n_tail += 1;
  else {
    s->n_own += 1;
    const dim3 grid((r.K + 3) / 4), block(256);                // one wave per chain
    hipLaunchKernelGGL(payne_rwalk_kernel, grid, block, 0, st, s->sd, s->walk, s->lnl, w, w > 0 ? 1 : 0, w < r.walks ? 1 : 0,
                       w == 0 ? s->tail_dev : (WalkTail*)nullptr);
  }
  s->tail_done = false;
  int rc = PAYNE_OK;
  if (w < r.walks) {
    TailReq tr{s->tail_dev, w + 1, w + 1 < r.walks ? 1 : 0, false, s->spec ? &s->walk : nullptr};
    rc = lnlike_impl(s->ctx, s->rows, r.K, s->lnl, r.stream, &tr);
    s->tail_done = tr.done;
  } else s->run.open = false;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(s->ctx, PAYNE_E_HIP, std::string("rwalk launch: ") + hipGetErrorString(e));
  return rc;
}
extern "C" int payne_sampler_counters(const payne_sampler* s, long long out[2]) {
  if (!s || !out) return PAYNE_E_INVALID;
  out[0] = s->n_tail; out[1] = s->n_own;
  return PAYNE_OK;
}
extern "C" int payne_rwalk_batch(payne_sampler* s, double* u, double* v, double* lnprob, int K, const double* axes,
                                 double scale, double loglstar, int walks, unsigned long long seed, int* nacc, int* ncall,
                                 void* stream) {
  int rc = payne_rwalk_begin(s, u, v, lnprob, K, axes, scale, loglstar, walks, seed, nacc, ncall, stream);
  for (int w = 0; !rc && w <= walks; ++w) rc = payne_rwalk_step(s, w);
  return rc;
}

// The random-walk queue of the batched nested sampler in one call: start points, ellipsoid assignment, upload, the walk,
// download, and the chains that moved as the proposal queue (header: payne_ns_rwalk_queue).
// In two parts, so that the caller's host work (the next bound, bookkeeping of another fit) can run while the GPU walks:
// _begin returns once everything is enqueued (start points, ellipsoid assignment, upload, walks + 1 steps, download requests),
// _end waits for the stream and selects the chains that moved.  payne_ns_rwalk_queue is the two back to back.
// `src` (payne_ns_rwalk_queue_turn): live slot i holds row src[i] of (qu, qv) with lnprob lg[i] when src[i] >= 0 -- the live set a
// queue's consumption will leave, by index (payne_ns::peek_index), never copied
// The queue's staging block up (from mapped host memory) and down (into it; ONE workgroup, which then publishes the queue's
// sequence number behind the block with system scope -- what payne_ns_rwalk_queue_end waits for).
__global__ void __launch_bounds__(256) payne_stage_in_kernel(double* __restrict__ dst, const double* __restrict__ src_host, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src_host[i];
}
__global__ void __launch_bounds__(1024) payne_stage_out_kernel(double* __restrict__ dst_host, const double* __restrict__ src, size_t n,
                                                               unsigned long long* flag, unsigned long long seq,
                                                               const double* __restrict__ src2 = nullptr, int n2 = 0,
                                                               unsigned* arrivals = nullptr) {
  // (several workgroups when `arrivals` is given: the last one to arrive publishes -- atomicInc wraps the count back to zero)
  for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (size_t)gridDim.x * 1024) dst_host[i] = src[i];
  if (blockIdx.x == 0 && src2 && (int)threadIdx.x < n2) dst_host[n + threadIdx.x] = src2[threadIdx.x];   // (the device's scale and threshold behind the block)
  // (every wave's stores are acknowledged before it passes the barrier; ONE thread's release then covers the workgroup's -- release
  // fences are cumulative --: a fence in each of the sixteen waves, each a write-back of the L2, was 6 us in the turn kernel)
  __syncthreads();
  if (threadIdx.x == 0) {
    bool last = true;
    if (arrivals && gridDim.x > 1) { __threadfence_system(); last = atomicInc(arrivals, gridDim.x - 1) == gridDim.x - 1; }
    if (last) { __threadfence_system(); __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
  }
}

// ---- the queue's turn on the device ------------------------------------------------------------------------------------
// What payne_ns_rwalk_queue_turn does on the host between two queues, as ONE workgroup: the chains that moved are the proposals;
// consuming them in order (each replaces the worst live point if it beats it) leaves the nlive LARGEST of live points and proposals
// -- thresholds only rise, so a proposal in that set beat every threshold it met, and one outside it died or never got in --: a sort
// by (lnprob descending, live points before proposals on ties: the test is strict).  Then the scale adaptation dynesty-style from
// the queue's counters, the new threshold (the largest lnprob left outside the set: the last point to die), and every chain's start point, uniform among the new live
// points (the host's splitmix of the seed).  The host replays the same queue for the evidence in its own time; its live SET is the
// same, its slot order is not (nothing on the device depends on it).
// sum of an int over the wave, in lane 63 (an inclusive scan inside each row of 16 lanes, then the rows' totals handed on)
template <int CTRL> __device__ __forceinline__ int turn_dpp_add(int x) { return x + __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ int turn_wave_sum_to_last(int x) {
  x = turn_dpp_add<0x111>(x); x = turn_dpp_add<0x112>(x); x = turn_dpp_add<0x114>(x); x = turn_dpp_add<0x118>(x);   // row_shr 1, 2, 4, 8
  x = turn_dpp_add<0x142>(x); x = turn_dpp_add<0x143>(x);                                                            // row_bcast 15, 31
  return x;
}
// (the network's exchanges at distances below 64: lane_xor_i32, sampler_core.hpp)
// one stage of the bitonic network at distance J < 64 inside runs of KK, "before" = larger lnprob, then smaller id
template <int KK, int J>
__device__ __forceinline__ void turn_cmpx(double& kv, int& iv, int i) {
  union { double d; int w[2]; } me, pa;
  me.d = kv;
  const bool upper = (i & J) != 0;
  pa.w[0] = lane_xor_i32<J>(me.w[0], upper); pa.w[1] = lane_xor_i32<J>(me.w[1], upper);
  const int ip = lane_xor_i32<J>(iv, upper);
  const bool mine_first = (kv > pa.d) || (kv == pa.d && iv < ip);
  const bool want_first = (!upper) == ((i & KK) == 0);                       // the lower place of an ascending run, the upper of a descending one
  if (mine_first != want_first) { kv = pa.d; iv = ip; }
  if constexpr (J > 1) turn_cmpx<KK, J / 2>(kv, iv, i);
}
struct TurnArgs {
  const double *lu, *lv, *ll; double *ou, *ov, *ol;    // live set in / out (the same arrays when merge == 0)
  int nlive, nd, K, merge, n2;                         // n2: power of two >= nlive + K (<= 2048)
  double *cu, *cv, *cl; int *na, *nc, *nr;             // chains: the finished queue's results in, the next queue's start points out
  double* dyn;                                         // {scale, loglstar}
  double scale0, lstar0;                               // merge == 0: what to start from
  unsigned long long seed;
  // the finished queue's results on their way to the host from HERE (its own transfer kernel was 9 us between two queues): the
  // stores are issued first and drain under the sort; the completion word follows the kernel's last statement
  double* exp_dst; int exp_n; unsigned long long* exp_flag; unsigned long long exp_seq;
  const double* ax_src; double* ax_dst; int ax_n;      // a new bound (axes, centres, inverse axes) from its mapped host block: a launch of its own was 2.8 us in front of this one
  int live_sorted;                                     // the live set comes from a merging turn: best first (rows and lnprob)
  int rows_lds;                                        // the launch carries nlive * nd * 16 bytes of dynamic LDS: the new live set's rows stay there for the start points
};
constexpr size_t kTurnRowsLdsMax = 128 * 1024;
__global__ void __launch_bounds__(1024) payne_ns_turn_kernel(TurnArgs a) {
  __shared__ double key[2048];
  __shared__ int id[2048];
  __shared__ int wsum[3][16];
  __shared__ int got_in_flag;
  extern __shared__ __attribute__((aligned(16))) double turn_rows[];   // [2][nlive * nd] when a.rows_lds
  const int tid = threadIdx.x, nl = a.nlive, nd = a.nd, K = a.K;
  // With the new rows in LDS nothing below the sort reads what this kernel stored to global memory: the barriers there need not wait
  // for the stores to be acknowledged (1.5 us each time) -- the one in front of the completion word does.
  const bool lds_rows = a.merge && a.rows_lds;
  auto lds_barrier = []() {                                  // (__syncthreads also waits for the stores towards the host to be acknowledged)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  };
  auto turn_barrier = [&]() {
    if (lds_rows) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); }
    else __syncthreads();
  };
  double scale = a.scale0, lstar = a.lstar0;
  // Everything the kernel reads before the sort is REQUESTED first, in one go: the keys (a counter and the lnprob it admits), the
  // counters, the scale and threshold, the export's values (sixteen to a thread).  One value at a time -- a load, its store towards
  // the host, the next load: the two may alias -- the export alone was 13 memory latencies end to end (3.3 us), the keys (the counter,
  // THEN the lnprob) and the counters three more.  The export's stores go out right in front of the sort, whose exchanges and
  // comparisons leave the memory path to them.
  // (Every one of these loads is unconditional, its index clamped: a load under a branch leaves the compiler without a count of the
  // loads behind it, and it waits for ALL of them -- the bound's values included, 2 us away across the bus -- in front of the sort.)
  double l_in[2]; int na_in[2];
  int s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
  for (int q = 0; q < 2; ++q) {                               // (n2 <= 2048; read whether or not this turn merges: one basic block, loads in source order)
    const int e = tid + q * 1024;
    const bool is_live = e < nl;
    const int k = (!is_live && e < nl + K) ? e - nl : 0;
    const double* lp = is_live ? a.ll + e : a.cl + k;
    l_in[q] = *lp;
    na_in[q] = a.na[k];
  }
  const int kc = tid < K ? tid : K - 1;
  const int cnt0 = a.na[kc], cnt1 = a.nc[kc], cnt2 = a.nr[kc];
  const double dyn0 = a.dyn[0], dyn1 = a.dyn[1];
  constexpr int kExpBatch = 16;
  double ex[kExpBatch];
  if (a.exp_dst) {
#pragma unroll
    for (int q = 0; q < kExpBatch; ++q) { const int e = tid + q * 1024; ex[q] = e < a.exp_n ? a.cu[e] : 0.0; }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) if (tid + q * 1024 < nl) na_in[q] = 1;
  if (tid < K) { s0 = cnt0; s1 = cnt1; s2 = cnt2; }
  if (a.merge) for (int k = tid + 1024; k < K; k += 1024) { s0 += a.na[k]; s1 += a.nc[k]; s2 += a.nr[k]; }
  auto export_out = [&]() {                                  // chains | counters (contiguous from a.cu), then the scale and threshold they ran under
    if (!a.exp_dst) return;
#pragma unroll
    for (int q = 0; q < kExpBatch; ++q) { const int e = tid + q * 1024; if (e < a.exp_n) a.exp_dst[e] = ex[q]; }
    for (int e = tid + kExpBatch * 1024; e < a.exp_n; e += 1024) a.exp_dst[e] = a.cu[e];
    if (tid < 2) a.exp_dst[a.exp_n + tid] = tid ? dyn1 : dyn0;
  };
  if (a.merge) {
    // the queue's counters: a sum per wave (a thousand atomics on three LDS words were 25 us of this kernel), met by thread 0 below
    s0 = turn_wave_sum_to_last(s0); s1 = turn_wave_sum_to_last(s1); s2 = turn_wave_sum_to_last(s2);
    if ((tid & 63) == 63) { wsum[0][tid >> 6] = s0; wsum[1][tid >> 6] = s1; wsum[2][tid >> 6] = s2; }
    if (tid == 0) got_in_flag = 0;
    auto adapted_scale = [&]() {                               // dynesty-style, from the queue's counters
      long long c0 = 0, c1 = 0, c2 = 0;
      for (int w = 0; w < 16; ++w) { c0 += wsum[0][w]; c1 += wsum[1][w]; c2 += wsum[2][w]; }
      const long long denom = c1 + c2 > 1 ? c1 + c2 : 1;
      const double frac = (double)c0 / (double)denom;         // a redrawn (out-of-cube) proposal counts as a rejection
      double sc = dyn0 * exp((frac - 0.5) / nd / 0.5);
      sc = sc > 1e-4 ? sc : 1e-4;
      return sc < 4.0 ? sc : 4.0;
    };
    double kv0[2]; int iv0[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = tid + q * 1024;
      kv0[q] = -INFINITY; iv0[q] = (1 << 30) + e;
      if (e < nl + K && na_in[q] > 0) { const double l = l_in[q]; kv0[q] = (l != l) ? -INFINITY : l; iv0[q] = e; }
    }
    // bitonic sort, "before" = larger lnprob, then smaller id
    if (a.n2 <= 1024) {
      // One element per thread, in registers.  The exchanges at distances below 64 stay inside the wave (45 of the 55 stages of 1024
      // elements), the others go through LDS, two buffers in turn: one barrier a stage (every stage as an LDS pass between two
      // barriers made this kernel 38 us).
      // A live set that comes from a merging turn arrives best first, which is the order the network's first stages (runs up to half
      // the array) would bring it to: when it IS one half of the array its threads sit those stages out.
      const int i = tid;
      const bool half_sorted = a.live_sorted && 2 * nl == a.n2 && (nl & 63) == 0;
      double kv = kv0[0];
      int iv = iv0[0];
      lds_barrier();                                          // (the flag's zero in front of its ones; the waves' counters)
      export_out();
      if (tid == 0) scale = adapted_scale();                  // (no part of the sort's result in it: here thread 0's wave has the time, with a sorted half)
      int buf = 0;
      auto lds_stages = [&](int kk, bool idle) {                // distances 64 and up
        for (int j = kk >> 1; j >= 64; j >>= 1) {
          double* kb = key + buf * 1024; int* ib = id + buf * 1024;
          buf ^= 1;
          if (i < a.n2) { kb[i] = kv; ib[i] = iv; }             // (the sorted half's threads too: their partners read them at the last level)
          lds_barrier();
          if (!idle) {
            const double kp = i < a.n2 ? kb[i ^ j] : kv; const int ip = i < a.n2 ? ib[i ^ j] : iv;
            const bool mine_first = (kv > kp) || (kv == kp && iv < ip);
            const bool want_first = (((i & j) == 0) == ((i & kk) == 0));
            if (mine_first != want_first) { kv = kp; iv = ip; }
          }
        }
      };
      if (a.n2 == 1024) {                                       // the usual size, unrolled: the distances are compile-time constants
#define PAYNE_TURN_LEVEL(KK) do { const bool idle = half_sorted && KK <= nl && i < nl; lds_stages(KK, idle); \
                                  if (!idle) turn_cmpx<KK, (KK / 2 < 32 ? KK / 2 : 32)>(kv, iv, i); } while (0)
        PAYNE_TURN_LEVEL(2); PAYNE_TURN_LEVEL(4); PAYNE_TURN_LEVEL(8); PAYNE_TURN_LEVEL(16); PAYNE_TURN_LEVEL(32);
        PAYNE_TURN_LEVEL(64); PAYNE_TURN_LEVEL(128); PAYNE_TURN_LEVEL(256); PAYNE_TURN_LEVEL(512); PAYNE_TURN_LEVEL(1024);
#undef PAYNE_TURN_LEVEL
      } else {
        for (int kk = 2; kk <= a.n2; kk <<= 1) {
          const bool idle = half_sorted && kk <= nl && i < nl;   // (the same for a whole wave: nl is a multiple of 64 here)
          lds_stages(kk, idle);
          if (idle) continue;
          for (int j = kk >> 1 < 32 ? kk >> 1 : 32; j > 0; j >>= 1) {
            const double kp = __shfl_xor(kv, j); const int ip = __shfl_xor(iv, j);
            const bool mine_first = (kv > kp) || (kv == kp && iv < ip);
            const bool want_first = (((i & j) == 0) == ((i & kk) == 0));      // the lower place of an ascending run, the upper of a descending one
            if (mine_first != want_first) { kv = kp; iv = ip; }
          }
        }
      }
      if (buf == 1) lds_barrier();                            // (the last exchange read the first buffer, which the result goes to)
      if (i < a.n2) { key[i] = kv; id[i] = iv; }
      if (i < nl && iv >= nl) got_in_flag = 1;
    } else {
#pragma unroll
      for (int q = 0; q < 2; ++q) { const int e = tid + q * 1024; if (e < a.n2) { key[e] = kv0[q]; id[e] = iv0[q]; } }
      export_out();
      lds_barrier();
      if (tid == 0) scale = adapted_scale();
      for (int kk = 2; kk <= a.n2; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
          lds_barrier();
          for (int i = tid; i < a.n2; i += 1024) {
            const int p = i ^ j;
            if (p > i) {
              const double ki = key[i], kp = key[p]; const int ii = id[i], ip = id[p];
              const bool i_first = (ki > kp) || (ki == kp && ii < ip);
              const bool up = (i & kk) == 0;                   // this run sorts "first things first"
              if (i_first != up) { key[i] = kp; key[p] = ki; id[i] = ip; id[p] = ii; }
            }
          }
        }
      lds_barrier();
      for (int r = tid; r < nl; r += 1024) if (id[r] >= nl) got_in_flag = 1;
    }
    lds_barrier();
    // the threshold the next queue walks under: the largest lnprob left outside the new set.  That is the lnprob D of the last point
    // to die (dynesty's loglstar; payne_ns::peek_index) OR a proposal that was turned away after the last replacement, which lies
    // between D and the new live minimum M -- any threshold in [D, M) is valid to walk under (a proposal is tested again, against the
    // worst live point of its iteration, when the host consumes it), and the host accepts exactly that window (nested.py
    // _fill_queue_dev).  If no proposal got in, nothing died and the old threshold stays.
    if (tid == 0) lstar = got_in_flag ? key[nl] : dyn1;       // (thread 0 alone writes it, and the scale)
    // the new live set, row r = the r-th best
    // (four elements' loads in flight per thread: one element at a time this loop was a dozen memory latencies end to end, 2.6 us)
    const float inv_nd = 1.0f / (float)nd;                   // (e < 2^15, nd <= 64: (e + 0.5) / nd is never within rounding of an integer)
    for (int e0 = tid; e0 < nl * nd; e0 += 4096) {
      double xu[4], xv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = e0 + q * 1024;
        xu[q] = 0.0; xv[q] = 0.0;
        if (e < nl * nd) {
          const int r = (int)(((float)e + 0.5f) * inv_nd), d = e - r * nd, who = id[r];
          const bool live = who < nl;
          const size_t src = (size_t)(live ? who : who - nl) * nd + d;
          xu[q] = live ? a.lu[src] : a.cu[src];
          xv[q] = live ? a.lv[src] : a.cv[src];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = e0 + q * 1024;
        if (e < nl * nd) {
          a.ou[e] = xu[q]; a.ov[e] = xv[q];
          if (lds_rows) { turn_rows[e] = xu[q]; turn_rows[nl * nd + e] = xv[q]; }
        }
      }
    }
    for (int r = tid; r < nl; r += 1024) a.ol[r] = key[r];
  }
  // a new bound: read across the bus (2 us), requested HERE -- loads return in order, whoever waits for a later one waits for these --
  // and stored at the end, behind the chains' rows
  double axv[2] = {0.0, 0.0};
  if (a.ax_n > 0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) { const int e = tid + q * 1024; if (e < a.ax_n) axv[q] = a.ax_src[e]; }
  }
  if (!a.merge) export_out();
  turn_barrier();                                            // (the chains' rows are overwritten below -- the export above has read the old values --; the new set is read back by this workgroup only)
  if (tid == 0) { a.dyn[0] = scale; a.dyn[1] = lstar; }
  // start points: uniform among the live points (queue_begin_core's draw)
  auto mix = [](unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
  };
  for (int k = tid; k < K; k += 1024) {                       // (K <= 1024; the slot of every chain once, through the sort's LDS -- id[] is free: the barrier above)
    const unsigned long long r0 = mix(a.seed ^ (0xA5A5A5A5ull + (unsigned long long)k * 0x100000001B3ull));
    const int i = (nl & (nl - 1)) == 0 ? (int)(r0 & (unsigned long long)(nl - 1)) : (int)(r0 % (unsigned long long)nl);   // (the 64-bit remainder is a routine of 200 instructions)
    id[k] = i;
    a.cl[k] = a.merge ? key[i] : a.ol[i];
    a.na[k] = 0; a.nc[k] = 0; a.nr[k] = 0;
  }
  turn_barrier();
  const float inv_nd_ = 1.0f / (float)nd;
  for (int e0 = tid; e0 < K * nd; e0 += 4096) {
    double xu[4], xv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = e0 + q * 1024;
      xu[q] = 0.0; xv[q] = 0.0;
      if (e < K * nd) {
        const int k = (int)(((float)e + 0.5f) * inv_nd_), d = e - k * nd, i = id[k];
        xu[q] = lds_rows ? turn_rows[i * nd + d] : a.ou[(size_t)i * nd + d];
        xv[q] = lds_rows ? turn_rows[nl * nd + i * nd + d] : a.ov[(size_t)i * nd + d];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = e0 + q * 1024;
      if (e < K * nd) { a.cu[e] = xu[q]; a.cv[e] = xv[q]; }
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) { const int e = tid + q * 1024; if (e < a.ax_n) a.ax_dst[e] = axv[q]; }
  for (int e = tid + 2048; e < a.ax_n; e += 1024) a.ax_dst[e] = a.ax_src[e];
  if (a.exp_dst) {
    // every wave's stores are acknowledged before it passes the barrier (__syncthreads waits for them); ONE system-scope release then
    // covers them all (release fences are cumulative) -- a fence in each of the sixteen waves, each a write-back of the L2, was 6 us
    __syncthreads();
    if (tid == 0) __hip_atomic_store(a.exp_flag, a.exp_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
static int queue_begin_core(payne_sampler* s, const double* live_u, const double* live_v, const double* live_logl,
                            int nlive, int K, const double* axes_unit, int n_ell, const double* ctr, const double* ainv,
                            double scale, double loglstar, int walks, unsigned long long seed, void* stream,
                            const int* src, const double* qu, const double* qv, const double* lg);
extern "C" int payne_ns_rwalk_queue_begin(payne_sampler* s, const double* live_u, const double* live_v, const double* live_logl,
                                          int nlive, int K, const double* axes_unit, int n_ell, const double* ctr, const double* ainv,
                                          double scale, double loglstar, int walks, unsigned long long seed, void* stream) {
  return queue_begin_core(s, live_u, live_v, live_logl, nlive, K, axes_unit, n_ell, ctr, ainv, scale, loglstar, walks, seed, stream,
                          nullptr, nullptr, nullptr, nullptr);
}
static int queue_begin_core(payne_sampler* s, const double* live_u, const double* live_v, const double* live_logl,
                            int nlive, int K, const double* axes_unit, int n_ell, const double* ctr, const double* ainv,
                            double scale, double loglstar, int walks, unsigned long long seed, void* stream,
                            const int* src, const double* qu, const double* qv, const double* lg) {
  int rc = sampler_check(s, live_u, K, live_v);
  if (rc) return rc;
  payne_ctx* c = s->ctx;
  if (s->queue_open) {
    // a queue begun and never collected (an abandoned generator, a caller of _begin that skipped _end): its transfer DOWN into
    // q_host may still be pending on its stream and would land on top of the start points written below -- wait for it first
    HIPCHK(c, hipStreamSynchronize(reinterpret_cast<hipStream_t>(s->queue_stream)));
  }
  s->queue_open = false;
  if (!live_logl || !axes_unit || nlive <= 0 || walks <= 0)
    return fail(c, PAYNE_E_INVALID, "bad payne_ns_rwalk_queue arguments");
  if (n_ell < 1 || n_ell > PAYNE_MAX_ELL || (n_ell > 1 && (!ctr || !ainv))) return fail(c, PAYNE_E_INVALID, "bad ellipsoid list");
  const int nd = s->sd.ndim;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // ---- start points: uniform among the live points (splitmix of the seed).  With several ellipsoids the walk's first step finds
  //      the one each chain steps in on the device (walk_assign_ell: on the host that loop was 8 us per ellipsoid, before the GPU
  //      could start)
  auto mix = [](unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
  };
  double* hu = s->q_host;
  double* hv = hu + (size_t)K * nd;
  double* hl = hv + (size_t)K * nd;
  for (int k = 0; k < K; ++k) {
    const unsigned long long r0 = mix(seed ^ (0xA5A5A5A5ull + (unsigned long long)k * 0x100000001B3ull));
    const int i = (int)(r0 % (unsigned long long)nlive);
    const bool q = src && src[i] >= 0;
    // (rows of a dozen doubles, 2 K of them between two queues: copied in place -- a memcpy call each was 20 us of the turn)
    const double* su = q ? qu + (size_t)src[i] * nd : live_u + (size_t)i * nd;
    const double* sv = q ? qv + (size_t)src[i] * nd : live_v + (size_t)i * nd;
    double* du_ = hu + (size_t)k * nd;
    double* dv_ = hv + (size_t)k * nd;
    for (int d = 0; d < nd; ++d) { du_[d] = su[d]; dv_[d] = sv[d]; }
    hl[k] = src ? lg[i] : live_logl[i];
  }
  // one transfer each way: chains, axes and (several ellipsoids) centres and inverse axes up; chains, then the three counters down
  const size_t nq_d = (size_t)K * (2 * nd + 1), n_cnt = ((size_t)3 * K + 1) / 2, n_ax = (size_t)n_ell * nd * nd;
  const size_t n_as = n_ell > 1 ? (size_t)n_ell * nd + n_ax : 0;
  double* hax = hl + K + n_cnt;
  std::memcpy(hax, axes_unit, n_ax * 8);
  if (n_ell > 1) {
    std::memcpy(hax + n_ax, ctr, (size_t)n_ell * nd * 8);
    std::memcpy(hax + n_ax + (size_t)n_ell * nd, ainv, n_ax * 8);
  }
  if (s->q_host_dev) {
    const size_t n_up = nq_d + n_cnt + n_ax + n_as;
    hipLaunchKernelGGL(payne_stage_in_kernel, dim3((unsigned)((n_up + 1023) / 1024)), dim3(256), 0, st, s->q_dev, s->q_host_dev, n_up);
  } else {
    HIPCHK(c, hipMemcpyAsync(s->q_dev, s->q_host, (nq_d + n_cnt + n_ax + n_as) * 8, hipMemcpyHostToDevice, st));
  }
  double* du = s->q_dev;
  double* dv = du + (size_t)K * nd;
  double* dl = dv + (size_t)K * nd;
  int* dna = reinterpret_cast<int*>(dl + K);
  int* dnc = dna + K;
  int* dnr = dnc + K;
  const double* dax = dl + K + n_cnt;
  int* dell = n_ell > 1 ? reinterpret_cast<int*>(const_cast<double*>(dax) + n_ax + n_as) : nullptr;   // (device only: written by the first step)
  rwalk_begin_impl(s, du, dv, dl, K, dax, dell, scale, loglstar, walks, seed, dna, dnc, dnr, stream);
  if (n_ell > 1) { s->walk.as_ctr = dax + n_ax; s->walk.as_ainv = dax + n_ax + (size_t)n_ell * nd; s->walk.ell_out = dell; s->walk.n_ell = n_ell; }
  for (int w = 0; !rc && w <= walks; ++w) rc = payne_rwalk_step(s, w);
  if (rc) return rc;
  if (s->q_host_dev) {
    ++s->q_seq;
    hipLaunchKernelGGL(payne_stage_out_kernel, dim3(s->q_arrivals ? 8 : 1), dim3(1024), 0, st, s->q_host_dev, s->q_dev, nq_d + n_cnt, s->q_flag_dev, s->q_seq,
                       (const double*)nullptr, 0, s->q_arrivals);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, PAYNE_E_HIP, std::string("queue staging launch: ") + hipGetErrorString(e));
  } else {
    HIPCHK(c, hipMemcpyAsync(s->q_host, s->q_dev, (nq_d + n_cnt) * 8, hipMemcpyDeviceToHost, st));
  }
  s->queue_open = true; s->queue_K = K; s->queue_stream = stream;
  return PAYNE_OK;
}
static void queue_extract(const double* hu, int K, int nd, double* qu, double* qv, double* ql, int* qnc, int* nq, long long* stats);
static bool wait_word(volatile unsigned long long* w, unsigned long long want, double seconds) {
  // Spin while the wait is as long as queues have been taking (1.5 x the wait before: a C2 queue is ~1 ms, and a timer's wake-up
  // after a 50 us sleep cost the host-turn loop 50-100 us a collect when queues ran just over a fixed 1 ms window), at least one
  // millisecond; beyond that give the core away between looks -- a queue of 65 536-pixel spectra takes seconds, which one host
  // core used to burn: first by yielding, from 20 ms on by sleeping.
  static thread_local double last_wait = 1e-3;
  const double spin_for = std::min(20e-3, std::max(1e-3, 1.5 * last_wait));
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  int polite = 0;                                            // 0 spin | 1 yield | 2 sleep
  double dt = 0.0;
  bool ok = true;
  while (__atomic_load_n(const_cast<const unsigned long long*>(w), __ATOMIC_ACQUIRE) != want) {
    if (polite == 2) std::this_thread::sleep_for(std::chrono::microseconds(50));
    else if (polite == 1) std::this_thread::yield();
    if (polite || (++spins & 0x3FFu) == 0) {
      dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (dt > seconds) { ok = false; break; }
      polite = dt > 20e-3 ? 2 : (dt > spin_for ? 1 : 0);
    }
  }
  if (ok) {
    dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    last_wait = 0.5 * last_wait + 0.5 * dt;
  }
  return ok;
}
extern "C" int payne_ns_rwalk_queue_end(payne_sampler* s, double* qu, double* qv, double* ql, int* qnc, int* nq, long long* stats) {
  if (!s) return PAYNE_E_INVALID;
  payne_ctx* c = s->ctx;
  if (!s->queue_open) return fail(c, PAYNE_E_INVALID, "payne_ns_rwalk_queue_end without a queue in flight");
  if (!qu || !qv || !ql || !qnc || !nq || !stats) return fail(c, PAYNE_E_INVALID, "bad payne_ns_rwalk_queue_end arguments");
  s->queue_open = false;
  const int K = s->queue_K, nd = s->sd.ndim;
  hipStream_t st = reinterpret_cast<hipStream_t>(s->queue_stream);
  const double* hu = s->q_host;
  if (s->q_flag) {
    // the word the queue's last kernel writes behind its results (every kernel before it on the stream has completed by then);
    // a queue that does not report within 10 s is handed to hipStreamSynchronize, which returns the fault if there was one
    if (!wait_word(s->q_flag, s->q_seq, 10.0)) HIPCHK(c, hipStreamSynchronize(st));
  } else {
    HIPCHK(c, hipStreamSynchronize(st));
  }
  queue_extract(hu, K, nd, qu, qv, ql, qnc, nq, stats);
  return PAYNE_OK;
}
// ---- the chains that moved are the queue; a chain that never moved is a copy of a live point
static void queue_extract(const double* hu, int K, int nd, double* qu, double* qv, double* ql, int* qnc, int* nq, long long* stats) {
  const double* hv = hu + (size_t)K * nd;
  const double* hl = hv + (size_t)K * nd;
  long long acc = 0, calls = 0, redraw = 0, idle_calls = 0;
  int m = 0;
  const int *na = reinterpret_cast<const int*>(hl + K), *nc = na + K, *nr = nc + K;
  for (int k = 0; k < K; ++k) {
    acc += na[k]; calls += nc[k]; redraw += nr[k];
    if (na[k] > 0) {
      const double* su = hu + (size_t)k * nd;
      const double* sv = hv + (size_t)k * nd;
      double* du_ = qu + (size_t)m * nd;
      double* dv_ = qv + (size_t)m * nd;
      for (int d = 0; d < nd; ++d) { du_[d] = su[d]; dv_[d] = sv[d]; }
      const double l = hl[k];
      ql[m] = (l != l) ? -INFINITY : l;
      qnc[m] = nc[k] > 1 ? nc[k] : 1;
      ++m;
    } else {
      idle_calls += nc[k];
    }
  }
  *nq = m;
  stats[0] = acc; stats[1] = calls; stats[2] = redraw; stats[3] = idle_calls;
}

// ---- the queue's turn on the device: host entry points (header: payne_ns_queue_dev_*) ------------------------------------------
extern "C" int payne_ns_queue_dev_init(payne_sampler* s, const double* live_u, const double* live_v, const double* live_logl, int nlive,
                                       double scale, double loglstar) {
  if (!s) return PAYNE_E_INVALID;
  payne_ctx* c = s->ctx;
  if (!live_u || !live_v || !live_logl || nlive <= 0) return fail(c, PAYNE_E_INVALID, "bad payne_ns_queue_dev_init arguments");
  if (!s->q_host_dev) return fail(c, PAYNE_E_UNSUPPORTED, "the device-side turn needs the mapped staging block");
  if (s->dq_launched != s->dq_collected) return fail(c, PAYNE_E_INVALID, "payne_ns_queue_dev_init with queues in flight");
  const int nd = s->sd.ndim, K = s->k_max;
  if (nlive + K > 2048) return fail(c, PAYNE_E_UNSUPPORTED, "nlive + queue size > 2048");
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (prev != c->device) HIPCHK(c, hipSetDevice(c->device));
  if (nlive != s->lv_n) {
    for (int b = 0; b < 2; ++b) {
      void* p = nullptr;
      HIPCHK(c, hipMalloc(&p, (size_t)nlive * nd * 8)); s->owned.push_back(p); s->lv_u[b] = static_cast<double*>(p);
      HIPCHK(c, hipMalloc(&p, (size_t)nlive * nd * 8)); s->owned.push_back(p); s->lv_v[b] = static_cast<double*>(p);
      HIPCHK(c, hipMalloc(&p, (size_t)nlive * 8)); s->owned.push_back(p); s->lv_l[b] = static_cast<double*>(p);
    }
    s->lv_n = nlive;
  }
  if (!s->dyn) { void* p = nullptr; HIPCHK(c, hipMalloc(&p, 4 * 8)); s->owned.push_back(p); s->dyn = static_cast<double*>(p); }
  const size_t nblk = q_doubles(K, nd) + 16;
  for (int b = 0; b < 2; ++b) {
    if (!s->dq_host[b]) {
      HIPCHK(c, hipHostMalloc((void**)&s->dq_host[b], nblk * 8, hipHostMallocMapped));
      void* dp = nullptr;
      HIPCHK(c, hipHostGetDevicePointer(&dp, s->dq_host[b], 0));
      s->dq_host_dev[b] = static_cast<double*>(dp);
      s->dax_n = (int)((size_t)PAYNE_MAX_ELL * (2 * nd * nd + nd));
      HIPCHK(c, hipHostMalloc((void**)&s->dax_host[b], (size_t)s->dax_n * 8, hipHostMallocMapped));
      HIPCHK(c, hipHostGetDevicePointer(&dp, s->dax_host[b], 0));
      s->dax_host_dev[b] = static_cast<double*>(dp);
    }
    reinterpret_cast<volatile unsigned long long*>(s->dq_host[b] + q_doubles(K, nd) + 8)[0] = 0ull;
    s->dq_seq[b] = 0;
  }
  HIPCHK(c, hipMemcpy(s->lv_u[0], live_u, (size_t)nlive * nd * 8, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(s->lv_v[0], live_v, (size_t)nlive * nd * 8, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(s->lv_l[0], live_logl, (size_t)nlive * 8, hipMemcpyHostToDevice));
  const double d2[4] = {scale, loglstar, 0.0, 0.0};
  HIPCHK(c, hipMemcpy(s->dyn, d2, sizeof(d2), hipMemcpyHostToDevice));
  // (a function attribute belongs to the device it was set on: asked here, with the sampler's device current, not once per process)
  s->turn_lds_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(payne_ns_turn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)kTurnRowsLdsMax) == hipSuccess;
  s->lv_cur = 0; s->lv_sorted = false; s->dq_launched = 0; s->dq_collected = 0; s->dq_exported = 0; s->dq_n_ell = 0;
  return PAYNE_OK;
}
extern "C" int payne_ns_queue_dev_launch(payne_sampler* s, int K, const double* axes_unit, int n_ell, const double* ctr, const double* ainv,
                                         int walks, unsigned long long seed, int merge, void* stream) {
  if (!s) return PAYNE_E_INVALID;
  payne_ctx* c = s->ctx;
  if (s->lv_n <= 0 || !s->dyn) return fail(c, PAYNE_E_INVALID, "payne_ns_queue_dev_launch before payne_ns_queue_dev_init");
  if (K <= 0 || K > s->k_max || walks <= 0) return fail(c, PAYNE_E_INVALID, "bad payne_ns_queue_dev_launch arguments");
  if (s->dq_launched - s->dq_collected >= 2) return fail(c, PAYNE_E_INVALID, "two queues already in flight");
  if (s->dq_launched != s->dq_collected && K != s->dq_K) return fail(c, PAYNE_E_INVALID, "queue size changed with a queue in flight");
  if (s->queue_open) return fail(c, PAYNE_E_INVALID, "a host-turn queue is in flight");
  if (axes_unit) {
    if (n_ell < 1 || n_ell > PAYNE_MAX_ELL || (n_ell > 1 && (!ctr || !ainv))) return fail(c, PAYNE_E_INVALID, "bad ellipsoid list");
  } else {
    n_ell = s->dq_n_ell;
    if (n_ell < 1) return fail(c, PAYNE_E_INVALID, "no bound on the device yet");
  }
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (prev != c->device) HIPCHK(c, hipSetDevice(c->device));
  const int nd = s->sd.ndim, nl = s->lv_n, b = s->dq_launched & 1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t nq_d = (size_t)K * (2 * nd + 1), n_cnt = ((size_t)3 * K + 1) / 2, n_ax = (size_t)n_ell * nd * nd;
  const size_t n_as = n_ell > 1 ? (size_t)n_ell * nd + n_ax : 0;
  double* du = s->q_dev;
  double* dv = du + (size_t)K * nd;
  double* dl = dv + (size_t)K * nd;
  int* dna = reinterpret_cast<int*>(dl + K);
  int* dnc = dna + K;
  int* dnr = dnc + K;
  double* dax = dl + K + n_cnt;
  int* dell = n_ell > 1 ? reinterpret_cast<int*>(dax + n_ax + n_as) : nullptr;
  if (axes_unit) {                                          // a new bound: up through its own mapped block (two, used in turn)
    double* hax = s->dax_host[b];
    std::memcpy(hax, axes_unit, n_ax * 8);
    if (n_ell > 1) {
      std::memcpy(hax + n_ax, ctr, (size_t)n_ell * nd * 8);
      std::memcpy(hax + n_ax + (size_t)n_ell * nd, ainv, n_ax * 8);
    }
    s->dq_n_ell = n_ell;
  }
  TurnArgs ta{};
  ta.ax_src = s->dyn; ta.ax_dst = nullptr; ta.ax_n = 0;
  if (axes_unit) { ta.ax_src = s->dax_host_dev[b]; ta.ax_dst = dax; ta.ax_n = (int)(n_ax + n_as); }   // (up with the turn kernel)
  const int cur = s->lv_cur, nxt = merge ? cur ^ 1 : cur;
  ta.lu = s->lv_u[cur]; ta.lv = s->lv_v[cur]; ta.ll = s->lv_l[cur];
  ta.ou = s->lv_u[nxt]; ta.ov = s->lv_v[nxt]; ta.ol = s->lv_l[nxt];
  ta.nlive = nl; ta.nd = nd; ta.K = K; ta.merge = merge ? 1 : 0;
  int n2 = 2;
  while (n2 < nl + K) n2 <<= 1;
  ta.n2 = n2;
  ta.cu = du; ta.cv = dv; ta.cl = dl; ta.na = dna; ta.nc = dnc; ta.nr = dnr;
  ta.dyn = s->dyn; ta.scale0 = 0.0; ta.lstar0 = 0.0; ta.seed = seed;
  if (s->dq_launched > s->dq_exported) {                    // the queue before this one: its results leave with this turn
    const int pb = s->dq_exported & 1;
    ta.exp_dst = s->dq_host_dev[pb]; ta.exp_n = (int)(nq_d + n_cnt);
    ta.exp_flag = reinterpret_cast<unsigned long long*>(s->dq_host_dev[pb] + q_doubles(s->k_max, nd) + 8);
    ta.exp_seq = ++s->dq_seq[pb];
    ++s->dq_exported;
  }
  if (!merge) {                                             // the values payne_ns_queue_dev_init left stay
    double d2[2];
    HIPCHK(c, hipMemcpy(d2, s->dyn, sizeof(d2), hipMemcpyDeviceToHost));
    ta.scale0 = d2[0]; ta.lstar0 = d2[1];
  }
  size_t rows_bytes = (size_t)nl * nd * 16;
  if (rows_bytes > kTurnRowsLdsMax || !s->turn_lds_ok) rows_bytes = 0;
  ta.rows_lds = rows_bytes ? 1 : 0;
  ta.live_sorted = s->lv_sorted ? 1 : 0;
  if (merge) s->lv_sorted = true;
  hipLaunchKernelGGL(payne_ns_turn_kernel, dim3(1), dim3(1024), rows_bytes, st, ta);
  s->lv_cur = nxt;
  rwalk_begin_impl(s, du, dv, dl, K, dax, dell, 0.0, 0.0, walks, seed, dna, dnc, dnr, stream);
  s->walk.dyn = s->dyn;
  if (n_ell > 1) { s->walk.as_ctr = dax + n_ax; s->walk.as_ainv = dax + n_ax + (size_t)n_ell * nd; s->walk.ell_out = dell; s->walk.n_ell = n_ell; }
  int rc = PAYNE_OK;
  for (int w = 0; !rc && w <= walks; ++w) rc = payne_rwalk_step(s, w);
  if (rc) return rc;
  // (its results: with the NEXT queue's turn, or by payne_ns_queue_dev_collect when none has been launched by then)
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, PAYNE_E_HIP, std::string("device-turn launch: ") + hipGetErrorString(e));
  ++s->dq_launched; s->dq_K = K; s->queue_stream = stream;
  return PAYNE_OK;
}
extern "C" int payne_ns_queue_dev_collect(payne_sampler* s, double* qu, double* qv, double* ql, int* qnc, int* nq, long long* stats,
                                          double* dyn_used) {
  if (!s) return PAYNE_E_INVALID;
  payne_ctx* c = s->ctx;
  if (s->dq_collected >= s->dq_launched) return fail(c, PAYNE_E_INVALID, "payne_ns_queue_dev_collect without a queue in flight");
  if (!qu || !qv || !ql || !qnc || !nq || !stats) return fail(c, PAYNE_E_INVALID, "bad payne_ns_queue_dev_collect arguments");
  const int b = s->dq_collected & 1, K = s->dq_K, nd = s->sd.ndim;
  if (s->dq_exported <= s->dq_collected) {                  // the newest queue, no turn behind it: a transfer of its own
    const size_t nq_d = (size_t)K * (2 * nd + 1), n_cnt = ((size_t)3 * K + 1) / 2;
    ++s->dq_seq[b];
    hipLaunchKernelGGL(payne_stage_out_kernel, dim3(s->q_arrivals ? 8 : 1), dim3(1024), 0, reinterpret_cast<hipStream_t>(s->queue_stream),
                       s->dq_host_dev[b], s->q_dev, nq_d + n_cnt,
                       reinterpret_cast<unsigned long long*>(s->dq_host_dev[b] + q_doubles(s->k_max, nd) + 8), s->dq_seq[b], s->dyn, 2, s->q_arrivals);
    ++s->dq_exported;
  }
  volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(s->dq_host[b] + q_doubles(s->k_max, nd) + 8);
  if (!wait_word(flag, s->dq_seq[b], 10.0)) {
    HIPCHK(c, hipStreamSynchronize(reinterpret_cast<hipStream_t>(s->queue_stream)));
    if (*flag != s->dq_seq[b]) return fail(c, PAYNE_E_HIP, "the queue's completion word never arrived");
  }
  queue_extract(s->dq_host[b], K, nd, qu, qv, ql, qnc, nq, stats);
  if (dyn_used) {
    const size_t nq_d = (size_t)K * (2 * nd + 1), n_cnt = ((size_t)3 * K + 1) / 2;
    dyn_used[0] = s->dq_host[b][nq_d + n_cnt]; dyn_used[1] = s->dq_host[b][nq_d + n_cnt + 1];
  }
  ++s->dq_collected;
  return PAYNE_OK;
}
// One turn of the sampler's pipelined loop in ONE call: collect the queue in flight (_end), adapt the step scale to its acceptance
// (dynesty's rule, as thepayne_amd/sampler/nested.py applies it), predict the live set and threshold its consumption will leave
// (payne_ns_peek) and launch the next queue from there (_begin) -- between a queue's last transfer and the next one's first
// kernel the GPU idles, and every microsecond of interpreter there is one of those.
extern "C" int payne_ns_rwalk_queue_turn(payne_sampler* s, double* qu, double* qv, double* ql, int* qnc, int* nq, long long* stats,
                                         const double* live_u, const double* live_v, const double* live_logl, int nlive, int K,
                                         const double* axes_unit, int n_ell, const double* ctr, const double* ainv, double* scale,
                                         double* loglstar, int walks, unsigned long long seed, int* n_dead) {
  if (!s) return PAYNE_E_INVALID;
  if (!scale || !loglstar || !n_dead || !live_u || !live_v || !live_logl || nlive <= 0) return fail(s->ctx, PAYNE_E_INVALID, "bad payne_ns_rwalk_queue_turn arguments");
  void* stream = s->queue_stream;
  int rc = payne_ns_rwalk_queue_end(s, qu, qv, ql, qnc, nq, stats);
  if (rc) return rc;
  const int nd = s->sd.ndim;
  {
    const long long denom = stats[1] + stats[2] > 1 ? stats[1] + stats[2] : 1;
    const double frac = (double)stats[0] / (double)denom;          // a redrawn (out-of-cube) proposal counts as a rejection
    double sc = *scale * exp((frac - 0.5) / nd / 0.5);
    sc = sc > 1e-4 ? sc : 1e-4;
    *scale = sc < 4.0 ? sc : 4.0;
  }
  payne_ns::peek_index(nlive, live_logl, ql, *nq, s->pk_src, s->pk_l, s->pk_heap, loglstar, n_dead);
  return queue_begin_core(s, live_u, live_v, live_logl, nlive, K, axes_unit, n_ell, ctr, ainv, *scale, *loglstar, walks, seed, stream,
                          s->pk_src.data(), qu, qv, s->pk_l.data());
}
extern "C" int payne_ns_rwalk_queue(payne_sampler* s, const double* live_u, const double* live_v, const double* live_logl,
                                    int nlive, int K, const double* axes_unit, int n_ell, const double* ctr, const double* ainv,
                                    double scale, double loglstar, int walks, unsigned long long seed, double* qu, double* qv,
                                    double* ql, int* qnc, int* nq, long long* stats, void* stream) {
  if (s && (!qu || !qv || !ql || !qnc || !nq || !stats)) return fail(s->ctx, PAYNE_E_INVALID, "bad payne_ns_rwalk_queue arguments");
  const int rc = payne_ns_rwalk_queue_begin(s, live_u, live_v, live_logl, nlive, K, axes_unit, n_ell, ctr, ainv, scale, loglstar,
                                            walks, seed, stream);
  return rc ? rc : payne_ns_rwalk_queue_end(s, qu, qv, ql, qnc, nq, stats);
}

extern "C" int payne_bc_batch(payne_ctx* c, const double* x, int B, double* bc, void* stream) {
  int rc = check_call(c, x, B, bc);
  if (rc) return rc;
  if (!c->has_phot) return fail(c, PAYNE_E_INVALID, "context has no photometric model");
  return run_sed(c, x, 6, 2, B, bc, reinterpret_cast<hipStream_t>(stream));
}

// ---- per-kernel timing ---------------------------------------------------------
extern "C" int payne_profile(payne_ctx* c, int enable) {
  if (!c) return PAYNE_E_INVALID;
  c->prof = enable != 0;
  if (enable) {
    c->prof_used = 0;
    for (int k = 0; k < 4; ++k) { c->prof_ms[k] = 0.0; c->prof_n[k] = 0; }
  }
  return PAYNE_OK;
}

extern "C" int payne_profile_read(payne_ctx* c, int kind, double* total_ms, long long* launches) {
  if (!c || kind < 0 || kind > 3) return PAYNE_E_INVALID;
  for (size_t i = 0; i < c->prof_used; ++i) {
    auto& r = c->prof_pool[i];
    float ms = 0.f;
    HIPCHK(c, hipEventSynchronize(r.e1));
    HIPCHK(c, hipEventElapsedTime(&ms, r.e0, r.e1));
    c->prof_ms[r.kind] += ms;
    c->prof_n[r.kind] += 1;
  }
  c->prof_used = 0;
  if (total_ms) *total_ms = c->prof_ms[kind];
  if (launches) *launches = c->prof_n[kind];
  return PAYNE_OK;
}

#ifdef PAYNE_STAMPS
// Diagnostic build only: cycle stamps of the first hidden-layer launch ([grid][16], slot 15 = grid size).
extern "C" int payne_diag_hidden_stamps(payne_ctx* c, const double* theta, int B, unsigned long long* host, int max_blocks) {
  int rc = check_call(c, theta, B, host);
  if (rc) return rc;
  unsigned long long* d = nullptr;
  const size_t nb_ = (size_t)(max_blocks < 0 ? -max_blocks : max_blocks);
  HIPCHK(c, hipMalloc(&d, nb_ * 16 * 8));
  HIPCHK(c, hipMemset(d, 0, nb_ * 16 * 8));
  if (max_blocks < 0) { max_blocks = -max_blocks; g_dense_stamps = d; } else g_hidden_stamps = d;   // negative: the output layer
  bool sed = c->has_phot;
  rc = run_ann(c, theta, B, 2.355, nullptr, true, &sed);
  g_hidden_stamps = nullptr; g_dense_stamps = nullptr;
  HIPCHK(c, hipDeviceSynchronize());
  HIPCHK(c, hipMemcpy(host, d, (size_t)max_blocks * 16 * 8, hipMemcpyDeviceToHost));
  (void)hipFree(d);
  return rc;
}
// Diagnostic build only: one lnlike batch with per-phase cycle stamps of the post kernel.
// stamps: host [B][payne_diag_stamp_row()] (slot 0 = number of stamps, slots 1.. = s_memtime after each barrier).
extern "C" int payne_diag_stamp_row(void) { return kStampRow; }
extern "C" int payne_diag_post_stamps(payne_ctx* c, const double* theta, int B, unsigned long long* stamps_host) {
  int rc = check_call(c, theta, B, stamps_host);
  if (rc) return rc;
  unsigned long long* d = nullptr;
  double* lnl = nullptr;
  HIPCHK(c, hipMalloc(&d, (size_t)B * kStampRow * 8));
  HIPCHK(c, hipMalloc(&lnl, (size_t)B * 8));
  HIPCHK(c, hipMemset(d, 0, (size_t)B * kStampRow * 8));
  if ((rc = run_ann(c, theta, B, 2.355, nullptr))) return rc;
  PostArgs a{};
  a.theta = theta; a.ld_theta = c->ncols; a.instr_factor = 2.355; a.raw = c->raw; a.ld_raw = c->T.npix;
  if (c->raw_freq && c->freq_rs_now) { a.ld_raw = c->T.n1; a.ld_raw_alt = c->T.npix; a.rot_flag = c->rot_flag; a.rot_seq = c->rot_seq; }
  a.out_stage = -1; a.lnl = lnl; a.stamps = d; a.prep = c->prep_valid ? c->prep : nullptr;
  a.stamp_sparse = getenv("PAYNE_DIAG_SPARSE") ? 1 : 0;
  if (c->big_ws && c->big_chip) {
    const int grid = B < c->big_grid ? B : c->big_grid;
    hipLaunchKernelGGL(payne_post_chip_kernel, dim3(grid), dim3(kChipThreads), kChipLdsBytes, nullptr, c->T, a, c->big_ws, B);
  } else if (c->big_ws) {                                  // spectra larger than LDS (PAYNE_BIG_TILED=0: plain passes)
    const int grid = B < c->big_grid ? B : c->big_grid;
    const int tiled = c->big_tiled ? 1 : 0;
    const size_t lds = tiled ? 2 * (size_t)fft_tile_complex() * sizeof(c32) : 0;
    if (tiled) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(payne_post_big_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(payne_post_big_kernel, dim3(grid), dim3(kBigThreads), lds, nullptr, c->T, a, c->big_ws, B, tiled);
  } else {
    hipLaunchKernelGGL(c->post_fn, dim3(B), dim3(kPostThreads), c->post_lds, nullptr, c->T.twf, a.raw, a.prep, a.theta, a.rot_flag, a.mags,
                       post_lead_ints(a.ld_raw, a.ld_theta, a.n_filters, c->T.raw_freq), (unsigned)a.rot_seq, c->T, a);
  }
  HIPCHK(c, hipDeviceSynchronize());
  HIPCHK(c, hipMemcpy(stamps_host, d, (size_t)B * kStampRow * 8, hipMemcpyDeviceToHost));
  (void)hipFree(d); (void)hipFree(lnl);
  return PAYNE_OK;
}
#endif
