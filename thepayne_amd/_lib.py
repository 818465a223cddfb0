"""ctypes binding of libpayne_hip.so (include/payne_hip.h).

The product path has no CPU fallback: if the HIP library is missing or does
not export the ABI, importing the engine raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpayne_hip.so")

PAYNE_MAX_LAYERS = 8
PAYNE_MAX_POLY = 12
ACT_NONE, ACT_LRELU, ACT_SIGMOID = 0, 1, 2
F_FWHM_R = 1
SMOOTH_VEL_DIRECT, SMOOTH_WAVE_DIRECT, SMOOTH_LSF_DIRECT, SMOOTH_WAVE_FFT, SMOOTH_INTERP = range(5)
ABI_VERSION = 2
E_INVALID, E_UNSUPPORTED, E_HIP, E_BATCH, E_SIGMA = -1, -2, -3, -4, -5
V_OUT_GENERIC, V_POST_GENERIC, V_TW_GLOBAL, V_POST_FULL, V_NO_PREP, V_SELECT_MEDIAN, V_LSF_GLOBAL, V_NO_WALK_TAIL, V_OUT_BK64, V_OUT_ROLLED, V_OUT_F32, V_SED_OWN_LAUNCH, V_BIG_WORKSPACE = 1, 2, 4, 8, 16, 64, 128, 512, 1024, 2048, 4096, 8192, 65536
V_OUT_SMALL_TILES = 16384
V_NO_WALK_SPEC = 131072
V_ROWS_PIXEL = 262144
V_OUT_PLANES = 524288
V_OUT_BF16X3 = 1048576
V_HID_F32 = 2097152
V_HID_CHAIN = 4194304
V_HID_WAVES4 = 8388608
V_OUT_WHOLE_TILE = 16777216

SYMBOLS = ["payne_version", "payne_ctx_create", "payne_ctx_set_obs", "payne_ctx_set_continuum", "payne_ctx_set_lsf", "payne_ctx_set_lsf_on", "payne_ctx_destroy", "payne_last_error",
           "payne_theta_cols", "payne_lnlike_batch", "payne_predict_batch", "payne_smooth_batch", "payne_smooth_direct", "payne_sed_batch", "payne_bc_batch", "payne_kernel_name", "payne_last_kernel", "payne_activation_batch",
           "payne_profile", "payne_profile_read",
           "payne_sampler_create", "payne_sampler_destroy", "payne_prior_transform_batch", "payne_lnprob_u_batch",
           "payne_rwalk_batch", "payne_rwalk_begin", "payne_rwalk_begin_ell", "payne_rwalk_step", "payne_sampler_counters", "payne_ns_rwalk_queue", "payne_ns_rwalk_queue_begin", "payne_ns_rwalk_queue_end", "payne_ns_rwalk_queue_turn", "payne_ns_consume", "payne_ns_peek", "payne_ns_bound", "payne_format_rows",
           "payne_ns_queue_dev_init", "payne_ns_queue_dev_launch", "payne_ns_queue_dev_collect"]

PAYNE_MAX_DIM, PAYNE_MAX_FIXED = 24, 16
PRIOR_UNIFORM, PRIOR_GAUSSIAN, PRIOR_TGAUSSIAN, PRIOR_EXP, PRIOR_TEXP, PRIOR_LOGUNIFORM, PRIOR_TABLE = range(7)

_dp = C.POINTER(C.c_double)


class Layer(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("n_in", C.c_int), ("n_out", C.c_int), ("act", C.c_int)]


class ModelDesc(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("layers", Layer * PAYNE_MAX_LAYERS), ("n_labels", C.c_int),
                ("xmin", _dp), ("xmax", _dp), ("npix", C.c_int), ("wavelength", _dp), ("resolution", C.c_double)]


class ObsDesc(C.Structure):
    _fields_ = [("nobs", C.c_int), ("wave", _dp), ("flux", _dp), ("eflux", _dp)]


class PhotDesc(C.Structure):
    _fields_ = [("n_filters", C.c_int), ("hidden", C.c_int),
                ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("w3", C.c_void_p), ("b3", C.c_void_p),
                ("xmin", _dp), ("xmax", _dp), ("hiav", _dp), ("obs_mag", _dp), ("obs_err", _dp)]


class Opts(C.Structure):
    _fields_ = [("b_max", C.c_int), ("npoly", C.c_int), ("photscale", C.c_int), ("variant", C.c_uint)]


class PriorDim(C.Structure):
    _fields_ = [("kind", C.c_int), ("theta_col", C.c_int), ("p", C.c_double * 4),
                ("has_gauss", C.c_int), ("has_box", C.c_int),
                ("g_mu", C.c_double), ("g_sigma", C.c_double), ("box_lo", C.c_double), ("box_hi", C.c_double)]


class AdvPriors(C.Structure):
    _fields_ = [("imf", C.c_int), ("vrot", C.c_int), ("vrot_mass_one", C.c_int),
                ("dim_logg", C.c_int), ("dim_logr", C.c_int), ("dim_vrot", C.c_int),
                ("val_logg", C.c_double), ("val_logr", C.c_double), ("val_vrot", C.c_double),
                ("plx_dim", C.c_int), ("plx_has_gauss", C.c_int), ("plx_has_box", C.c_int),
                ("plx_mu", C.c_double), ("plx_sigma", C.c_double), ("plx_lo", C.c_double), ("plx_hi", C.c_double),
                ("tab_cdf", C.c_void_p), ("tab_val", C.c_void_p), ("tab_n", C.c_int)]


class SamplerDesc(C.Structure):
    _fields_ = [("ndim", C.c_int), ("dims", PriorDim * 24), ("n_fixed", C.c_int),
                ("fixed_col", C.c_int * 16), ("fixed_val", C.c_double * 16), ("adv", AdvPriors)]


class NsState(C.Structure):
    _fields_ = [("nlive", C.c_int), ("ndim", C.c_int), ("it", C.c_longlong), ("pending_nc", C.c_longlong),
                ("logz", C.c_double), ("logzvar", C.c_double), ("h", C.c_double), ("logvol", C.c_double),
                ("loglstar", C.c_double)]


class NsDead(C.Structure):
    # (plain addresses: ndarray.ctypes.data_as costs tens of microseconds per array)
    _fields_ = [(k, C.c_void_p) for k in ("worst", "u", "v", "logl", "logvol", "logwt", "logz", "logzvar", "h", "nc",
                                          "worst_it", "delta_logz")]


NS_QUEUE_EMPTY, NS_CONVERGED, NS_LIMIT, NS_LOGL_MAX = range(4)


class PayneLibraryError(RuntimeError):
    pass


_lib = None


def load(path=None):
    """Load (once) and type the library.  torch is imported first so that this
    process holds a single HIP runtime (libamdhip64.so.7) shared with the
    tensors whose pointers we pass."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (must precede the dlopen; see docstring)
    path = path or os.environ.get("PAYNE_HIP_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise PayneLibraryError(
            "%s not found: build it with `python -m thepayne_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback." % path)
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise PayneLibraryError("cannot load %s: %s" % (path, e))
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise PayneLibraryError("%s lacks symbols %s" % (path, missing))
    ctxp = C.c_void_p
    lib.payne_version.restype = C.c_int
    lib.payne_ctx_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(ObsDesc), C.POINTER(PhotDesc),
                                     C.POINTER(Opts), C.c_int, C.POINTER(ctxp)]
    lib.payne_ctx_create.restype = C.c_int
    lib.payne_ctx_set_obs.argtypes = [ctxp, C.POINTER(ObsDesc)]
    lib.payne_ctx_set_obs.restype = C.c_int
    lib.payne_ctx_set_continuum.argtypes = [ctxp, C.POINTER(ModelDesc)]
    lib.payne_ctx_set_continuum.restype = C.c_int
    lib.payne_ctx_set_lsf_on.argtypes = [ctxp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
    lib.payne_ctx_set_lsf_on.restype = C.c_int
    lib.payne_ctx_set_lsf.argtypes = [ctxp, C.POINTER(C.c_double), C.c_int]
    lib.payne_ctx_set_lsf.restype = C.c_int
    lib.payne_smooth_batch.argtypes = [ctxp, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_uint, C.c_void_p, C.c_int,
                                       C.c_void_p]
    lib.payne_smooth_batch.restype = C.c_int
    lib.payne_smooth_direct.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                        C.c_double, C.c_int, C.c_double, C.c_void_p]
    lib.payne_smooth_direct.restype = C.c_int
    lib.payne_ctx_destroy.argtypes = [ctxp]
    lib.payne_ctx_destroy.restype = None
    lib.payne_last_error.argtypes = [ctxp]
    lib.payne_last_error.restype = C.c_char_p
    lib.payne_theta_cols.argtypes = [ctxp]
    lib.payne_theta_cols.restype = C.c_int
    lib.payne_lnlike_batch.argtypes = [ctxp, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.payne_lnlike_batch.restype = C.c_int
    lib.payne_predict_batch.argtypes = [ctxp, C.c_void_p, C.c_int, C.c_int, C.c_uint, C.c_void_p, C.c_int, C.c_void_p]
    lib.payne_predict_batch.restype = C.c_int
    lib.payne_sed_batch.argtypes = [ctxp, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.payne_sed_batch.restype = C.c_int
    lib.payne_bc_batch.argtypes = [ctxp, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.payne_bc_batch.restype = C.c_int
    lib.payne_profile.argtypes = [ctxp, C.c_int]
    lib.payne_profile.restype = C.c_int
    lib.payne_profile_read.argtypes = [ctxp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
    lib.payne_profile_read.restype = C.c_int
    lib.payne_sampler_create.argtypes = [ctxp, C.POINTER(SamplerDesc), C.c_int, C.POINTER(ctxp)]
    lib.payne_sampler_create.restype = C.c_int
    lib.payne_sampler_destroy.argtypes = [ctxp]
    lib.payne_sampler_destroy.restype = None
    lib.payne_prior_transform_batch.argtypes = [ctxp, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.payne_prior_transform_batch.restype = C.c_int
    lib.payne_lnprob_u_batch.argtypes = [ctxp, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.payne_lnprob_u_batch.restype = C.c_int
    lib.payne_rwalk_batch.argtypes = [ctxp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_double,
                                      C.c_double, C.c_int, C.c_ulonglong, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.payne_rwalk_batch.restype = C.c_int
    lib.payne_rwalk_begin.argtypes = lib.payne_rwalk_batch.argtypes
    lib.payne_rwalk_begin.restype = C.c_int
    lib.payne_rwalk_begin_ell.argtypes = [ctxp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                          C.c_double, C.c_double, C.c_int, C.c_ulonglong, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.payne_rwalk_begin_ell.restype = C.c_int
    lib.payne_ns_rwalk_queue.argtypes = [ctxp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_ulonglong, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
    lib.payne_ns_rwalk_queue.restype = C.c_int
    lib.payne_ns_rwalk_queue_begin.argtypes = [ctxp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                               C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_ulonglong, C.c_void_p]
    lib.payne_ns_rwalk_queue_begin.restype = C.c_int
    lib.payne_ns_rwalk_queue_end.argtypes = [ctxp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
    lib.payne_ns_rwalk_queue_end.restype = C.c_int
    lib.payne_ns_rwalk_queue_turn.argtypes = [ctxp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                              C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_ulonglong,
                                              C.POINTER(C.c_int)]
    lib.payne_ns_rwalk_queue_turn.restype = C.c_int
    lib.payne_ns_queue_dev_init.argtypes = [ctxp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double]
    lib.payne_ns_queue_dev_init.restype = C.c_int
    lib.payne_ns_queue_dev_launch.argtypes = [ctxp, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_ulonglong, C.c_int, C.c_void_p]
    lib.payne_ns_queue_dev_launch.restype = C.c_int
    lib.payne_ns_queue_dev_collect.argtypes = [ctxp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
    lib.payne_ns_queue_dev_collect.restype = C.c_int
    lib.payne_sampler_counters.argtypes = [ctxp, C.POINTER(C.c_longlong)]
    lib.payne_sampler_counters.restype = C.c_int
    lib.payne_rwalk_step.argtypes = [ctxp, C.c_int]
    lib.payne_rwalk_step.restype = C.c_int
    lib.payne_kernel_name.argtypes = [C.c_int]
    lib.payne_kernel_name.restype = C.c_char_p
    lib.payne_last_kernel.argtypes = [ctxp, C.c_int]
    lib.payne_last_kernel.restype = C.c_char_p
    lib.payne_activation_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.payne_activation_batch.restype = C.c_int
    vp, ip = C.c_void_p, C.POINTER(C.c_int)
    lib.payne_ns_bound.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, vp, vp, vp, vp, vp, ip]
    lib.payne_ns_bound.restype = C.c_int
    lib.payne_format_rows.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.c_longlong]
    lib.payne_format_rows.restype = C.c_longlong
    lib.payne_ns_consume.argtypes = [C.POINTER(NsState), vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_double,
                                     C.c_longlong, C.c_double, C.POINTER(NsDead), C.c_int, ip, ip]
    lib.payne_ns_consume.restype = C.c_int
    lib.payne_ns_peek.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.POINTER(C.c_double), ip]
    lib.payne_ns_peek.restype = C.c_int
    if lib.payne_version() != ABI_VERSION:
        raise PayneLibraryError("ABI version %d != %d" % (lib.payne_version(), ABI_VERSION))
    _lib = lib
    return lib
