/* payne_hip.h -- C ABI of libpayne_hip.so: the MI355X (gfx950) implementation of
 * ThePayne's nested-sampling likelihood hot path, evaluated in batch.
 *
 * Each entry point names the reference interface (pacargile/ThePayne, paths relative
 * to the reference root) it stands in for.  The reference is pure Python with no FFI
 * of its own; this is the boundary a maintainer binds with ctypes.
 *
 * Conventions
 *  - plain C, no torch/HIP types in signatures (`stream` is a hipStream_t passed as
 *    void*; NULL = the default stream);
 *  - return 0 on success, a negative PAYNE_E_* code on failure; never throws;
 *    payne_last_error() gives the message;
 *  - pointers documented "device" are fp32/fp64 arrays in the memory of the context's
 *    GPU, owned by the caller and alive for as long as the context uses them (weights,
 *    theta, outputs); pointers documented "host" are read during the call only;
 *  - batch calls enqueue work on `stream` and do not synchronise; the caller
 *    synchronises the stream before reading an output buffer;
 *  - a context is bound to one device and is not thread-safe (the reference's callers,
 *    dynesty's sampling loop, are single-threaded: Payne/fitting/fitstar.py:309-338);
 *  - NaN in -> NaN out at the same places as the reference.
 */
#ifndef PAYNE_HIP_H
#define PAYNE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define PAYNE_ABI_VERSION 2

#define PAYNE_OK 0
#define PAYNE_E_INVALID (-1)     /* bad argument / descriptor */
#define PAYNE_E_UNSUPPORTED (-2) /* valid in the reference, not (yet) by this library */
#define PAYNE_E_HIP (-3)         /* HIP runtime error */
#define PAYNE_E_BATCH (-4)       /* B exceeds opts.b_max */
#define PAYNE_E_SIGMA (-5)       /* payne_smooth_direct: target sigma below the input's (the reference raises ValueError) */

/* activation codes */
#define PAYNE_ACT_NONE 0
#define PAYNE_ACT_LRELU 1   /* z*(z>0) + 0.01*z*(z<0)  Payne/predict/ystpred.py:41-45 */
#define PAYNE_ACT_SIGMOID 2 /* torch.sigmoid            Payne/train/NNmodels.py:155-160 */

#define PAYNE_MAX_LAYERS 8
#define PAYNE_MAX_LABELS 5
#define PAYNE_MAX_POLY 12

/* One dense layer y = act(W x + b); W is [n_out][n_in] row-major fp32 exactly as the
 * reference stores it (w_array_k / model/linK.weight / model/features.K.weight). */
typedef struct payne_layer {
  const float* w; /* device [n_out*n_in] */
  const float* b; /* device [n_out] */
  int n_in, n_out;
  int act; /* activation applied to this layer's OUTPUT */
} payne_layer;

/* The spectral emulator: replaces ystpred.Net (Payne/predict/ystpred.py:18-58) and
 * predictspec.ANN + NNmodels.{LinNet,SMLP} (Payne/predict/predictspec.py:29-74,
 * Payne/train/NNmodels.py:92-168).  YST1: 3 layers lrelu,lrelu,none; SMLP: 4 layers
 * lrelu x3,none; LinNet: 6 layers sigmoid x5,none.  Input encoding for all of them:
 * (x - xmin)/(xmax - xmin) - 0.5. */
typedef struct payne_model_desc {
  int n_layers; /* >= 2 */
  payne_layer layers[PAYNE_MAX_LAYERS];
  int n_labels;             /* 4: Teff,logg,[Fe/H],[a/Fe]; 5: + vmic */
  const double* xmin;       /* host [n_labels] (Teff in K: apply ystpred.py:76-79 first) */
  const double* xmax;       /* host [n_labels] */
  int npix;                 /* == layers[n_layers-1].n_out */
  const double* wavelength; /* host [npix], strictly increasing, Angstrom */
  double resolution;        /* sigma-based R of the ANN (file key 'resolution') */
} payne_model_desc;

/* The observed spectrum: fitargs['obs_wave_fit','obs_flux_fit','obs_eflux_fit']
 * (Payne/fitting/fitstar.py:71-98).  flux/eflux may be NULL (predict-only context). */
typedef struct payne_obs_desc {
  int nobs;
  const double* wave;  /* host [nobs] */
  const double* flux;  /* host [nobs] or NULL */
  const double* eflux; /* host [nobs] or NULL */
} payne_obs_desc;

/* Photometric emulator: the stacked per-filter nets of photANN.fastANN
 * (Payne/predict/photANN.py:95-131) + highAv table (Payne/predict/highred.py:4-25)
 * + observed magnitudes fitargs['obs_phot'] (Payne/fitting/likelihood.py:109-112). */
typedef struct payne_phot_desc {
  int n_filters, hidden;
  const float* w1; /* device [F][H][6] */
  const float* b1; /* device [F][H]    */
  const float* w2; /* device [F][H][H] */
  const float* b2; /* device [F][H]    */
  const float* w3; /* device [F][H]    */
  const float* b3; /* device [F]       */
  const double* xmin;    /* host [6] */
  const double* xmax;    /* host [6] */
  const double* hiav;    /* host [F][5] = a1,b1,a2,b2,c2 (NaN rows allowed) or NULL */
  const double* obs_mag; /* host [F] or NULL */
  const double* obs_err; /* host [F] or NULL */
} payne_phot_desc;

typedef struct payne_opts {
  int b_max;     /* largest batch any call will pass (workspaces are sized for it) */
  int npoly;     /* Chebyshev blaze coefficients pc_0.. in theta (0 = modpoly off) */
  int photscale; /* 1: phot block carries log(A) (genphot_scaled); 0: log(R), Dist (genphot) */
  unsigned variant; /* 0 = the default kernels; PAYNE_V_* bits select the other code paths that ship */
} payne_opts;

/* Kernel variants (payne_opts.variant).  Every one computes the same results; they exist because the default of
 * each pair only covers some shapes (the other is what differently shaped nets / spectra run), and the parity
 * tests run the reference vectors through each of them. */
#define PAYNE_V_OUT_GENERIC 1u   /* output layer: register-staged GEMM (what nets with unequal hidden widths use) */
#define PAYNE_V_POST_GENERIC 2u  /* post kernel: runtime FFT geometry (what spectra other than 1k/2k/4k/8k use) */
#define PAYNE_V_TW_GLOBAL 4u     /* post kernel: twiddles read from L2 instead of LDS (what 8k spectra use) */
#define PAYNE_V_POST_FULL 8u     /* likelihood through the full post kernel instead of its likelihood-only build */
#define PAYNE_V_NO_PREP 16u      /* per-candidate records computed inside the post kernel (what 2-layer nets use) */
#define PAYNE_V_SELECT_MEDIAN 64u /* continuum / LSF medians by radix selection (what rows too long for an LDS sort use) */
#define PAYNE_V_NO_WALK_TAIL 512u /* the sampler's chain step as a launch of its own between two likelihood batches (what
                                  * contexts without a likelihood-only post kernel use) instead of at the post kernel's tail */
#define PAYNE_V_OUT_BK64 1024u   /* output layer: 64-deep stages (half as many barrier steps) when the padded width allows */
#define PAYNE_V_OUT_ROLLED 2048u /* output layer: the loop over k-steps as other hidden widths than 300 run it (not unrolled) */
#define PAYNE_V_OUT_F32 4096u    /* output layer: the fp32 matrix instruction (v_mfma_f32_32x32x2_f32) instead of six bf16 products of
                                  * operands split in three (fp32-accurate either way; what launches with several tiles per CU use) */
#define PAYNE_V_SED_OWN_LAUNCH 8192u /* joint likelihoods: the photometric nets as a launch of their own (one wave per candidate and
                                    * filter: what payne_sed_batch and nets wider than 64 use) instead of extra workgroups of the
                                    * hidden-layer launch */
#define PAYNE_V_OUT_SMALL_TILES 16384u /* output layer with many tiles per CU (C5): 64 x 128 tiles, two workgroups per CU (what batches that
                                       * are not whole 128 x 256 tiles use) instead of persistent workgroups on 128 x 256 tiles */
#define PAYNE_V_BIG_WORKSPACE 65536u /* 65 536-point spectra: both convolution stages through the global workspace (the four-step transform: what
                                      * other lengths above 16 384 use) instead of on the compute unit */
#define PAYNE_V_NO_WALK_SPEC 131072u /* the sampler's next proposal drawn at the post kernel's tail, after the likelihood it waits for (what fits
                                     * with more than 16 sampled dimensions or 32 theta columns use), instead of made ahead for both outcomes by
                                     * idle workgroups of the hidden-layer launch */
#define PAYNE_V_ROWS_PIXEL 262144u /* the output layer writes pixels and the post kernel transforms them itself (what runs with a continuum
                                    * network, with vsini maps that are not the identity, and for spectra other than 1k/2k/4k/8k);
                                    * default where it applies: the output layer's weights carry the first stage's forward transform */
#define PAYNE_V_OUT_PLANES 524288u /* output layer: the weights read as three bf16 planes split at payne_ctx_create (what nets of other widths than
                                    300 and batches with more tiles than compute units use) instead of fp32 weights split on their way into LDS */
#define PAYNE_V_OUT_BF16X3 1048576u /* output layer: operands split in three bf16 parts, six products (what batches with more tiles than
                                     compute units and nets whose last hidden layer could not be calibrated use) instead of two fp16 parts, three products */
#define PAYNE_V_HID_F32 2097152u  /* hidden layers: the fp32 matrix instruction for the second layer too (what later layers of deeper nets and
                                     widths other than 289..304 use) instead of products of fp16 pairs */
#define PAYNE_V_HID_CHAIN 4194304u /* deeper nets: the hidden layers past the second in ONE launch with hand-offs inside a 32-candidate row
                                      block (agent-scope release / acquire: measured 8 us a hop against 5.7 us a launch -- kept as a tested
                                      variant, not a default) */
#define PAYNE_V_HID_WAVES4 8388608u /* sigmoid nets: the first launch's tiles by four waves (what leaky-ReLU nets and launches that carry
                                      photometric tiles use) instead of eight */
#define PAYNE_V_OUT_WHOLE_TILE 16777216u /* output layer, one tile a compute unit: the tile's k-loop in one pass, all its rows stored at the end
                                           (what batches with more tiles than compute units use) instead of two halves, the first one's rows
                                           leaving under the second one's products */
#define PAYNE_V_LSF_GLOBAL 128u  /* LSF broadening with its buffers in global memory (what spectra > 8192 px use) */

typedef struct payne_ctx payne_ctx;

int payne_version(void);

/* Build a context on `device` (HIP ordinal).  model may be NULL for a photometry-only
 * fit, obs NULL if no observed grid is bound yet, phot NULL without photometry.
 * Replaces likelihood.__init__ -> GenMod._initspecnn/_initphotnn
 * (Payne/fitting/likelihood.py:7-40, Payne/fitting/genmod.py:15-43). */
int payne_ctx_create(const payne_model_desc* model, const payne_obs_desc* obs,
                     const payne_phot_desc* phot, const payne_opts* opts, int device,
                     payne_ctx** out);

/* Re-bind the observed grid (the `outwave` of getspec / a new spectrum). */
int payne_ctx_set_obs(payne_ctx* ctx, const payne_obs_desc* obs);

/* Bind (or, with NULL, remove) the continuum network of ystpred.PayneSpecPredict(Cnnpath=...)
 * (Payne/predict/ystpred.py:81-85): every spectrum the context produces from then on is the spectral
 * ANN's output times the continuum network's, converted F_nu -> F_lambda, normalised by its
 * NaN-ignoring median and interpolated onto the spectral ANN grid (NaN outside; ystpred.py:191-209).
 * Same descriptor as the spectral model (`resolution` unused); n_labels must match; any npix. */
int payne_ctx_set_continuum(payne_ctx* ctx, const payne_model_desc* cont);

/* LSF-vector instrumental broadening: getspec(inst_R=<array>, outwave=...) (Payne/predict/ystpred.py:248-269
 * -> smoothspec(smoothtype='lsf') -> smooth_lsf_fft, Payne/utils/smoothing.py:125-151, 482-586).
 * lsf: HOST fp64 [n], the Gaussian dispersion (same units as the wavelengths) at every pixel of the bound
 * observed grid (n must equal its length).  While a vector is set, theta's Inst_R column is ignored by
 * payne_lnlike_batch and stages 2/3 of payne_predict_batch; results are never NaN inside the model's range
 * (np.interp clamps).  NULL removes it; payne_ctx_set_obs removes it too.  Any spectrum length; the FFT length the
 * reference derives from the vector (smoothing.py:533-538) must not exceed the model's own (pow2ceil(npix)): NaN then. */
int payne_ctx_set_lsf(payne_ctx* ctx, const double* lsf, int n);
/* The same with the dispersion vector given on wavelengths of its own (HOST fp64 [n], strictly increasing) instead of on the
 * bound observed grid: smoothspec(wave, spec, resolution=<vector on wave>, outwave=<another grid>, smoothtype='lsf')
 * (Payne/utils/smoothing.py:125-151: the vector is a function of the INPUT wavelengths there; getspec interpolates it from the
 * output grid, ystpred.py:255-259).  The dispersion at a model pixel is np.interp of the vector (clamped outside). */
int payne_ctx_set_lsf_on(payne_ctx* ctx, const double* lsf_wave, const double* lsf, int n);

void payne_ctx_destroy(payne_ctx* ctx);

/* Message for the last failure on ctx (ctx == NULL: last create failure). */
const char* payne_last_error(const payne_ctx* ctx);

/* Number of fp64 columns of one theta row: 8 + npoly + 4.
 *   0 Teff  1 log(g)  2 [Fe/H]  3 [a/Fe]  4 Vrad  5 Vrot  6 Vmic  7 Inst_R
 *   8.. pc_0..pc_{npoly-1}
 *   then  log(A) | log(R),  Dist,  Av,  Rv
 * i.e. `specpars` followed by `photpars` of likelihood.lnlikefn
 * (Payne/fitting/likelihood.py:50-72); NaN = "absent" exactly as there. */
int payne_theta_cols(const payne_ctx* ctx);

/* lnL for B parameter vectors: likelihood.lnlike over a batch
 * (Payne/fitting/likelihood.py:84-117 -> genmod.py:58-108 -> ystpred.py:119-277).
 * Inst_R is the sampled FWHM-based value; the 2.355 factor of genmod.py:82-85 is
 * applied inside.  theta: device fp64 [B][payne_theta_cols]; lnl: device fp64 [B]. */
int payne_lnlike_batch(payne_ctx* ctx, const double* theta, int B, double* lnl, void* stream);

/* Model spectra for B parameter vectors.
 *   stage 0: raw ANN output on the ANN grid          (predictspec, ystpred.py:84-99)
 *   stage 1: after rotational broadening, ANN grid   (ystpred.py:211-224)
 *   stage 2: getspec on the bound observed grid      (ystpred.py:226-277)
 *   stage 3: genspec = stage 2 x Chebyshev blaze     (genmod.py:103-106)
 *   stage 4: the continuum network's own output      (predictcont, ystpred.py:101-117); ld_out >= its npix
 * With a continuum network bound (payne_ctx_set_continuum) stages 1-3 and the likelihood include the
 * normalised continuum (ystpred.py:191-209); stage 0 stays the spectral network's own output.
 * flags bit 0 (PAYNE_F_FWHM_R): theta[7] is FWHM-based (genspec semantics, x2.355);
 * otherwise it is the sigma-based R handed to getspec.
 * out: device fp32 [B][ld_out]; ld_out >= npix (stages 0,1) or nobs (stages 2,3). */
#define PAYNE_F_FWHM_R 1u
#define PAYNE_STAGE_CONT 4
int payne_predict_batch(payne_ctx* ctx, const double* theta, int B, int stage, unsigned flags,
                        float* out, int ld_out, void* stream);

/* The broadening stages on caller-supplied spectra: PayneSpecPredict.smoothspec (Payne/predict/ystpred.py:279-281 ->
 * Payne/utils/smoothing.py:19-169, the 'vsini' / 'vel' / 'R' / 'lsf' FFT branches).  spectra: device fp32
 * [B][ld_spec], full flux on the context's model wavelength grid; theta, stage (1..3), flags and out as for
 * payne_predict_batch -- the ANN forward pass is simply replaced by `spectra`.  Stage 4 (PAYNE_SMOOTH_VSINI_TO_OBS):
 * rotational broadening only, interpolated from the stage's own resampled grid onto the bound observed grid with NaN outside
 * (smoothspec(smoothtype='vsini', outwave=...), smoothing.py:293-312); out is [B][nobs]. */
#define PAYNE_SMOOTH_VSINI_TO_OBS 4
int payne_smooth_batch(payne_ctx* ctx, const float* spectra, int ld_spec, const double* theta, int B, int stage,
                       unsigned flags, float* out, int ld_out, void* stream);

/* smoothspec's branches that are not on the sampler's path, on caller-supplied HOST arrays, synchronous (no context: an
 * analysis helper of PayneSpecPredict.smoothspec, Payne/predict/ystpred.py:279-281 -> Payne/utils/smoothing.py):
 *   PAYNE_SMOOTH_VEL_DIRECT   smooth_vel       (smoothing.py:171-210)  fftsmooth=False, smoothtype 'vel' / 'R'
 *   PAYNE_SMOOTH_WAVE_DIRECT  smooth_wave      (:339-393)              fftsmooth=False, smoothtype 'lambda'; sigma scalar or [n]
 *   PAYNE_SMOOTH_LSF_DIRECT   smooth_lsf       (:435-480)              fftsmooth=False, smoothtype 'lsf'; sigma [nout] (or scalar)
 *   PAYNE_SMOOTH_WAVE_FFT     smooth_wave_fft  (:395-433)              fftsmooth=True,  smoothtype 'lambda'
 *   PAYNE_SMOOTH_INTERP       np.interp(outwave, wave, spec)           smooth_lsf with neither sigma nor lsf (:464-465)
 * wave, spec [n]: the input AFTER smoothspec's mask and nan_to_num (:131-138); sigma in the units the reference's function
 * takes (km/s for VEL_DIRECT, Angstrom otherwise); inres / in_vel / nsigma: its keywords (defaults 0 / 0 / 10).
 * Returns PAYNE_E_SIGMA where the reference raises ValueError (smooth_wave: target sigma below the input's, :381-383);
 * PAYNE_E_INVALID is a malformed call (null pointers, n < 2, a sigma vector of the wrong length). */
#define PAYNE_SMOOTH_VEL_DIRECT 0
#define PAYNE_SMOOTH_WAVE_DIRECT 1
#define PAYNE_SMOOTH_LSF_DIRECT 2
#define PAYNE_SMOOTH_WAVE_FFT 3
#define PAYNE_SMOOTH_INTERP 4
int payne_smooth_direct(int device, int kind, const double* wave, const double* spec, int n, const double* outwave, int nout,
                        const double* sigma, int nsig, double inres, int in_vel, double nsigma, double* out);

/* Magnitudes for B parameter vectors: FastPayneSEDPredict.sed
 * (Payne/predict/predictsed.py:75-103).  pars: device fp64 [B][9] =
 * logt, logg, feh, afe, av, rv, logl, dist, logA  (NaN = kwarg absent; the
 * (logl, dist) form wins over logA as in the reference).  mags: device fp64 [B][F]. */
int payne_sed_batch(payne_ctx* ctx, const double* pars, int B, double* mags, void* stream);

/* Bolometric corrections for B label vectors: fastANN.eval
 * (Payne/predict/photANN.py:125-131).  x: device fp64 [B][6] = Teff, logg, feh, afe,
 * av, rv; bc: device fp64 [B][F]. */
int payne_bc_batch(payne_ctx* ctx, const double* x, int B, double* bc, void* stream);

/* ---- device-side sampler step ------------------------------------
 * The reference evaluates prior.priortrans(u) and lnprobfn(v) one vector at a time on the host
 * (Payne/fitting/prior.py:126-272, Payne/fitting/fitstar.py:647-659; 334 us per priortrans call).
 * A batched sampler would be host-bound by that, so the unit-cube -> parameter transform, the
 * additive ln-priors, the assembly of theta rows and the random-walk proposal loop run on the
 * GPU; the host only launches and reads back the survivors. */
#define PAYNE_MAX_DIM 24
#define PAYNE_MAX_FIXED 16
#define PAYNE_PRIOR_UNIFORM 0    /* p = lo, hi                 (max-min)*u+min          prior.py:153-157 */
#define PAYNE_PRIOR_GAUSSIAN 1   /* p = mu, sigma              norm.ppf                 prior.py:158-159 */
#define PAYNE_PRIOR_TGAUSSIAN 2  /* p = lo, hi, mu, sigma      truncnorm.ppf, inf -> hi prior.py:161-166 */
#define PAYNE_PRIOR_EXP 3        /* p = loc, scale             expon.ppf                prior.py:167-168 */
#define PAYNE_PRIOR_TEXP 4       /* p = lo, hi, scale          truncexpon.ppf, inf -> hi prior.py:169-174 */
#define PAYNE_PRIOR_LOGUNIFORM 5 /* p = lo, hi */
#define PAYNE_PRIOR_TABLE 6      /* p[0] * np.interp(u, adv.tab_cdf, adv.tab_val): Dist under a GAL prior = 1000 * gal_ppf(u)
                                  * (prior.py:231-234 -> advancedpriors.py:665-670); one such dimension per sampler */

typedef struct payne_prior_dim {
  int kind;      /* PAYNE_PRIOR_* */
  int theta_col; /* column of the theta row this sampled dimension fills (payne_theta_cols layout) */
  double p[4];
  int has_gauss; /* additive ln-prior -0.5((v-mu)/sigma)^2      prior.py:388-390 */
  int has_box;   /* -inf outside [box_lo, box_hi]               prior.py:391-394 */
  double g_mu, g_sigma, box_lo, box_hi;
} payne_prior_dim;

/* The priors on DERIVED quantities prior.lnpriorfn adds (Payne/fitting/prior.py:286-336, :425-465 via
 * Payne/fitting/advancedpriors.py:93-137, :691-733): everything below is off when zero-initialised.
 * A parameter they need is named by the sampled dimension that carries it (dim_* >= 0) or, when it is fixed or absent
 * (dim_* < 0), by its value (NaN = absent). */
typedef struct payne_adv_priors {
  int imf;   /* + imf_lnprior(mass), Kroupa (alpha 1.3 / 2.3, break 0.5, -inf at <= 0.08), mass = 10^(logg + 2 logR - 4.437) */
  int vrot;  /* + vrot_lnprior(Vrot, mass, eep = 350, logg); mass = 10^(logg + 2 logR) when both are finite and
              * vrot_mass_one == 0, else 1 (prior.py:320-334: a photscale fit has no radius) */
  int vrot_mass_one;
  int dim_logg, dim_logr, dim_vrot;
  double val_logg, val_logr, val_vrot;
  int plx_dim;          /* dimension holding Dist for the derived 'Parallax' = 1000 / Dist (prior.py:449-451); < 0: none */
  int plx_has_gauss, plx_has_box;
  double plx_mu, plx_sigma, plx_lo, plx_hi;
  const double* tab_cdf; /* HOST [tab_n], non-decreasing in [0, 1]: abscissae of PAYNE_PRIOR_TABLE (copied at create) */
  const double* tab_val; /* HOST [tab_n] */
  int tab_n;
} payne_adv_priors;

typedef struct payne_sampler_desc {
  int ndim; /* <= PAYNE_MAX_DIM */
  payne_prior_dim dims[PAYNE_MAX_DIM];
  int n_fixed; /* 'fixed' parameters merged into every theta row (likelihood.py:47-48) */
  int fixed_col[PAYNE_MAX_FIXED];
  double fixed_val[PAYNE_MAX_FIXED];
  payne_adv_priors adv;
} payne_sampler_desc;

typedef struct payne_sampler payne_sampler;

int payne_sampler_create(payne_ctx* ctx, const payne_sampler_desc* desc, int k_max, payne_sampler** out);
void payne_sampler_destroy(payne_sampler* s);

/* v = priortrans(u) for K unit-cube vectors.  u, v: device fp64 [K][ndim]. */
int payne_prior_transform_batch(payne_sampler* s, const double* u, int K, double* v, void* stream);

/* v = priortrans(u); lnprob = lnprior(v) + lnlike(v)  (lnprobfn over a batch).
 * u, v: device fp64 [K][ndim]; lnprob: device fp64 [K]. */
int payne_lnprob_u_batch(payne_sampler* s, const double* u, int K, double* v, double* lnprob, void* stream);

/* `walks` Metropolis steps for K lock-step chains under the constraint lnprob > loglstar
 * (the 'rwalk' proposal of the nested sampler): u' = u + scale * (axes . z), z uniform in the
 * unit ball; a step is accepted iff u' is inside the unit cube and lnprob(u') > loglstar.
 * u, v: device fp64 [K][ndim] in/out (chain positions); lnprob: device fp64 [K] in/out;
 * axes: HOST fp64 [ndim][ndim] row-major; nacc, ncall: device int32 [K] (accepted steps,
 * likelihood calls) overwritten (zeroed by the walk's first step).  One small kernel + one payne_lnlike_batch for the first
 * step; the steps after it ride in the likelihood batch's own launches (payne_sampler_counters). */
int payne_rwalk_batch(payne_sampler* s, double* u, double* v, double* lnprob, int K, const double* axes,
                      double scale, double loglstar, int walks, unsigned long long seed, int* nacc, int* ncall,
                      void* stream);

/* ---- host-side nested-sampling bookkeeping (no GPU work) ---------------------------------
 * Replaces the per-iteration body of dynesty's sampling loop as the reference drives it
 * (Payne/fitting/fitstar.py:332-338: one 15-tuple per dead point) for a QUEUE of proposals
 * evaluated in one GPU batch: worst live point, trapezoid evidence update, replacement by the
 * next queued proposal with lnprob > threshold.  All arrays are the caller's. */
typedef struct payne_ns_state {
  int nlive, ndim;
  long long it;            /* iteration number of the next dead point (starts at 1) */
  long long pending_nc;    /* likelihood calls spent since the last accepted replacement */
  double logz, logzvar, h, logvol, loglstar;   /* start: -1e300, 0, 0, 0, -1e300 */
} payne_ns_state;

typedef struct payne_ns_dead {          /* one row per dead point, capacity `cap` rows */
  int* worst;       double* u;          /* [cap], [cap][ndim] */
  double* v;        double* logl;       /* [cap][ndim], [cap] */
  double* logvol;   double* logwt;
  double* logz;     double* logzvar;
  double* h;        int* nc;
  int* worst_it;    double* delta_logz;
} payne_ns_dead;

#define PAYNE_NS_QUEUE_EMPTY 0   /* every queued proposal was used or rejected */
#define PAYNE_NS_CONVERGED   1   /* ln(1 + L_max X / Z) < dlogz */
#define PAYNE_NS_LIMIT       2   /* max_emit or the record capacity reached */
#define PAYNE_NS_LOGL_MAX    3   /* worst live lnprob >= logl_max */

/* live_*: [nlive][ndim] / [nlive], updated in place.  q*: the proposal queue in order
 * ([nq][ndim], [nq]; qnc = likelihood calls behind each proposal).  Returns the number of dead
 * points written to `out` (>= 0) or a negative error code; *consumed = queue entries used up,
 * *stop = one of PAYNE_NS_*. */
int payne_ns_consume(payne_ns_state* s, double* live_u, double* live_v, double* live_logl, int* live_it,
                     const double* qu, const double* qv, const double* ql, const int* qnc, int nq,
                     double dlogz, long long max_emit, double logl_max, payne_ns_dead* out, int cap,
                     int* consumed, int* stop);

/* The live set and the threshold as payne_ns_consume WILL leave them once it has walked the whole queue (replacements only:
 * no evidence arithmetic, no records, no stop condition but the queue's end), into copies out_* of the live arrays;
 * *loglstar is written only when *n_dead > 0.  The batched sampler starts its next queue of proposals from this state and
 * consumes the current one while the GPU walks (no reference counterpart: dynesty proposes one point at a time). */
int payne_ns_peek(int nlive, int ndim, const double* live_u, const double* live_v, const double* live_logl, const double* qu,
                  const double* qv, const double* ql, int nq, double* out_u, double* out_v, double* out_logl, double* loglstar,
                  int* n_dead);

/* Bounding ellipsoid(s) of n live points u[n][ndim] (unit cube): dynesty's bound='single' (multi = 0) or
 * 'multi' (recursive 2-means split while the children hold less than half the parent's volume, at most
 * max_ell <= PAYNE_MAX_ELL pieces).  Outputs, one entry per ellipsoid {ctr + axes z, |z| <= 1}: ctr [.][ndim],
 * axes / axes_unit (the cluster's Cholesky factor scaled to hold its points, resp. by sqrt(ndim+2)) / ainv
 * (inverse of axes) [.][ndim][ndim] row-major, logvol [.] (up to the unit ball's volume); *n_ell = count. */
int payne_ns_bound(const double* u, int n, int ndim, double enlarge, int multi, int max_ell, double* ctr,
                   double* axes, double* axes_unit, double* ainv, double* logvol, int* n_ell);

/* Text rows of the output table (fitstar.py:345-371: "Iter <pars> log(lk) log(vol) log(wt) h nc log(z)
 * delta(log(z))", every value as Python's str() writes it): vals host fp64 [m][ncol], is_int[c] marks integer
 * columns; each value is followed by a blank, each row by a newline.  Returns bytes written (< 0: error; 48
 * bytes per value always suffice). */
long long payne_format_rows(const double* vals, int m, int ncol, const int* is_int, char* out, long long cap);

/* payne_rwalk_batch in two parts: `begin` takes the same arguments and enqueues the set-up, `step(s, w)` for
 * w = 0 .. walks enqueues one step (settle proposal w-1, draw and evaluate proposal w; the last only settles).
 * A caller driving several samplers (one context and one stream each) interleaves their steps from one host
 * thread to keep two independent batches in flight. */
int payne_rwalk_begin(payne_sampler* s, double* u, double* v, double* lnprob, int K, const double* axes,
                      double scale, double loglstar, int walks, unsigned long long seed, int* nacc, int* ncall,
                      void* stream);
int payne_rwalk_step(payne_sampler* s, int w);

/* `begin` for a multi-ellipsoid bound (dynesty bound='multi' as fitstar.py:314 passes it): axes is HOST fp64
 * [n_ell][ndim][ndim], ell is HOST int32 [K] naming the ellipsoid whose axes shape chain k's steps (NULL with
 * n_ell == 1 is payne_rwalk_begin). */
#define PAYNE_MAX_ELL 32
int payne_rwalk_begin_ell(payne_sampler* s, double* u, double* v, double* lnprob, int K, const double* axes,
                          int n_ell, const int* ell, double scale, double loglstar, int walks,
                          unsigned long long seed, int* nacc, int* ncall, void* stream);

/* One queue of random-walk proposals of the batched static sampler, start to finish (the body of the sampler's "fill the
 * queue" step: what dynesty does per proposal in sample_rwalk + the pool's queue, Payne/fitting/fitstar.py:309-338): K
 * chains start from live points drawn at random (splitmix of `seed`); with n_ell > 1 each steps in the metric of an
 * ellipsoid that holds its start point (ctr [n_ell][ndim] / ainv [n_ell][ndim][ndim] of payne_ns_bound; else the nearest);
 * upload, `walks` Metropolis steps under lnprob > loglstar on the device (payne_rwalk_begin_ell / _step; a proposal that
 * leaves the unit cube is redrawn without a likelihood call, as dynesty does), download, and the chains that moved at
 * least once become the queue: qu, qv [nq][ndim], ql [nq] (NaN -> -inf), qnc [nq] (likelihood calls behind each), HOST
 * arrays of capacity K owned by the caller.  stats[0..3] = accepted steps, likelihood calls, redrawn proposals, calls of the
 * chains that never moved.  live_*: HOST; axes_unit: HOST [n_ell][ndim][ndim].  Synchronises `stream`. */
int payne_ns_rwalk_queue(payne_sampler* s, const double* live_u, const double* live_v, const double* live_logl, int nlive,
                         int K, const double* axes_unit, int n_ell, const double* ctr, const double* ainv, double scale,
                         double loglstar, int walks, unsigned long long seed, double* qu, double* qv, double* ql, int* qnc,
                         int* nq, long long* stats, void* stream);

/* The same in two parts: _begin returns when everything is enqueued on `stream` (nothing of its host arguments is read
 * afterwards), _end waits for the stream and writes the queue.  Between the two the host is free: the batched sampler computes
 * the bounding ellipsoids of its NEXT update there (thepayne_amd/sampler/nested.py, overlap_bound). */
int payne_ns_rwalk_queue_begin(payne_sampler* s, const double* live_u, const double* live_v, const double* live_logl, int nlive,
                               int K, const double* axes_unit, int n_ell, const double* ctr, const double* ainv, double scale,
                               double loglstar, int walks, unsigned long long seed, void* stream);
int payne_ns_rwalk_queue_end(payne_sampler* s, double* qu, double* qv, double* ql, int* qnc, int* nq, long long* stats);

/* One turn of a loop that keeps a queue in flight: _end of the queue in flight (-> qu .. stats), the step scale adapted to its
 * acceptance (*scale in/out: scale * exp((acc / (calls + redrawn) - 0.5) / ndim / 0.5), clamped to [1e-4, 4]), payne_ns_peek of
 * that queue against the live arrays (*loglstar in: the current threshold; out: the one the queue's consumption will leave;
 * *n_dead: the dead points it will give), and _begin of the next queue from the predicted state, on the stream of the queue
 * collected.  The caller consumes the collected queue (payne_ns_consume) while the GPU walks. */
int payne_ns_rwalk_queue_turn(payne_sampler* s, double* qu, double* qv, double* ql, int* qnc, int* nq, long long* stats,
                              const double* live_u, const double* live_v, const double* live_logl, int nlive, int K,
                              const double* axes_unit, int n_ell, const double* ctr, const double* ainv, double* scale,
                              double* loglstar, int walks, unsigned long long seed, int* n_dead);

/* How the chain steps of this sampler ran so far: out[0] at the tail of the likelihood-only post kernel (the workgroup of
 * candidate k settles chain k's proposal as soon as it has the likelihood and takes the next one from the two that idle workgroups
 * of the batch's hidden-layer launch made ahead, one per outcome -- or draws it there: PAYNE_V_NO_WALK_SPEC), out[1] as launches of
 * their own (the first step of every walk; every step under PAYNE_V_NO_WALK_TAIL, with an LSF, or when the spectrum length
 * has no likelihood-only kernel).  Measurement / test aid, no reference counterpart. */
int payne_sampler_counters(const payne_sampler* s, long long out[2]);

/* The proposal queue's TURN on the device.  payne_ns_rwalk_queue_turn collects a queue, adapts the scale, predicts the live set and
 * launches the next queue from the host -- the GPU idles meanwhile (~100 us of a 1 ms cycle at C2).  Here the live set lives on the
 * device: one workgroup merges the finished queue's proposals into it (the nlive largest of live points and proposals: what consuming
 * them in order leaves, thresholds only rise), adapts the scale by dynesty's rule, takes the new threshold and draws every chain's
 * start point, and the next queue is ENQUEUED before the current one has finished.  The host consumes each queue (payne_ns_consume) for
 * the evidence in its own time; its live SET is the device's, its slot order is not (start points differ from the host-turn loop's:
 * the same sampler statistically, not to the bit).  dynesty has no counterpart (one proposal at a time).
 *   _init    uploads the live set and the scale / threshold to start from (no queue may be in flight);
 *   _launch  enqueues [new bound, if one is given] + turn (merge = 0: start from the uploaded set as it is) + walks + 1 steps + the
 *            results' transfer; at most two queues in flight; axes_unit = NULL keeps the bound already on the device;
 *   _collect waits for the OLDEST queue in flight and returns it as payne_ns_rwalk_queue_end does, with the scale and threshold
 *            it ran under in dyn_used[2]. */
int payne_ns_queue_dev_init(payne_sampler* s, const double* live_u, const double* live_v, const double* live_logl, int nlive,
                            double scale, double loglstar);
int payne_ns_queue_dev_launch(payne_sampler* s, int K, const double* axes_unit, int n_ell, const double* ctr, const double* ainv,
                              int walks, unsigned long long seed, int merge, void* stream);
int payne_ns_queue_dev_collect(payne_sampler* s, double* qu, double* qv, double* ql, int* qnc, int* nq, long long* stats,
                               double* dyn_used);

/* Kernel family names (for profiler filters): 0 dense layer, 1 post, 2 sed. */
const char* payne_kernel_name(int which);

/* The kernel (with its template arguments) the context's last batch call launched for one kind -- 0 output dense layer,
 * 1 post kernel, 2 sed kernel, 3 hidden dense layers (the last launch of that kind in the call) --, "" if none yet: which
 * code path a net of a given depth / a spectrum of a given length takes.  kind 4: "frequency" when the output layer handed the
 * post kernel rows already transformed (its weights restated once at payne_ctx_create: 1k/2k/4k/8k spectra on a geometric grid,
 * no continuum network, not PAYNE_V_ROWS_PIXEL), "pixels" otherwise.  Test and measurement aid, no reference counterpart. */
const char* payne_last_kernel(const payne_ctx* ctx, int kind);

/* out[i] = the activation `act` (PAYNE_ACT_*) of z[i], computed by the function the dense layers' epilogues call (device pointers,
 * n values, enqueued on `stream`).  Test aid, no reference counterpart: the sigmoid of NNmodels.py:92-121 is not the library's
 * expf + quotient here (dense_kernels.hpp sigmoid_f32), and its accuracy over the whole fp32 range is checked through this. */
int payne_activation_batch(const float* z, int n, int act, float* out, void* stream);

/* Per-kernel timing with HIP events recorded on the launch stream around every kernel
 * the batch calls enqueue (measurement aid for bench.py; no reference counterpart).
 * payne_profile(ctx, 1) clears the totals and starts recording, (ctx, 0) stops.
 * payne_profile_read waits for the recorded events and returns the accumulated
 * device time and launch count of one kind:
 *   0 output dense layer | 1 post kernel | 2 sed kernel | 3 hidden dense layers. */
int payne_profile(payne_ctx* ctx, int enable);
int payne_profile_read(payne_ctx* ctx, int kind, double* total_ms, long long* launches);

#ifdef __cplusplus
}
#endif
#endif /* PAYNE_HIP_H */
