"""PayneEngine: Python owner of one ``payne_ctx`` (include/payne_hip.h).

PyTorch-ROCm tensors are used for device storage only (weights, theta, outputs);
all arithmetic runs in the hand-written HIP kernels behind the C ABI.  There is no
CPU path: without a GPU, or without libpayne_hip.so, construction raises.
"""
import ctypes as C
import json
import os

import numpy as np

from . import _lib, nnio

__all__ = ["PayneEngine", "highav_coefficients", "THETA_SPEC_COLS"]

THETA_SPEC_COLS = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Vmic', 'Inst_R']
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def highav_coefficients(filters):
    """[F,5] a1,b1,a2,b2,c2 of the Av>=5 approximation (table of
    Payne/predict/highred.py:28-169, shipped as data/highav_table.json);
    NaN rows for filters the table lacks (highred.py:12-17)."""
    with open(os.path.join(_DATA, "highav_table.json")) as fh:
        tab = json.load(fh)["filters"]
    out = np.full((len(filters), 5), np.nan)
    for i, f in enumerate(filters):
        if f in tab:
            out[i] = [np.nan if v is None else v for v in tab[f]]
    return out


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class PayneEngine(object):
    """Batched likelihood / prediction engine on one MI355X.

    spec_net : dict from nnio.load_spec_net / normalize_spec_net, or None
    obs      : (wave, flux, eflux) | (wave,) | None
    phot     : stacked photometric nets (nnio.load_phot_nets / synth.make_phot_nets) or None
    obs_phot : ordered {filter: (mag, err)} matching phot['filters'], or None
    variant  : payne_opts.variant (PAYNE_V_* bits of include/payne_hip.h; 0 = default kernels)
    """

    def __init__(self, spec_net=None, obs=None, phot=None, obs_phot=None, npoly=0, photscale=False,
                 b_max=512, device=None, variant=0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("PayneEngine needs a ROCm GPU (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.torch = torch
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        self.b_max = int(b_max)
        self.npoly = int(npoly)
        self.photscale = bool(photscale)
        self._keep = []          # device tensors whose memory the context references
        self._host_keep = []     # host arrays referenced by descriptors during a call
        self._ctx = C.c_void_p()
        self.spec_net = spec_net
        self.phot = phot
        self.npix = 0
        self.nobs = 0
        self.npix_cont = 0
        mdesc = odesc = pdesc = None
        if spec_net is not None:
            mdesc = self._model_desc(spec_net)
        if obs is not None:
            odesc = self._obs_desc(*obs)
        if phot is not None:
            pdesc = self._phot_desc(phot, obs_phot)
        opts = _lib.Opts(self.b_max, self.npoly, int(self.photscale), int(variant))
        rc = self.lib.payne_ctx_create(mdesc, odesc, pdesc, C.byref(opts), self.device.index, C.byref(self._ctx))
        self._release_host()
        if rc != 0:
            raise RuntimeError("payne_ctx_create failed (%d): %s" % (rc, self.lib.payne_last_error(None).decode()))
        self.ncols = self.lib.payne_theta_cols(self._ctx)
        self.phot_off = 8 + self.npoly
        # Who is using this context (fitting/genmod.py keeps idle contexts of finished fits for the next fit of the same networks:
        # the owner GenMod and every DeviceProposer built on it hold it; the last one to let go hands it to `_on_idle`).
        self._holders = 0
        self._on_idle = None

    def hold(self):
        self._holders += 1

    def drop(self):
        """One holder less; the last one hands the (still open) context to `_on_idle` -- or closes it when nobody wants it."""
        self._holders -= 1
        if self._holders <= 0 and self.is_open():
            cb, self._on_idle = self._on_idle, None
            if cb is None or not cb(self):
                self.close()

    def is_open(self):
        return getattr(self, "_ctx", None) is not None and bool(self._ctx.value)

    # -- descriptors -----------------------------------------------------------
    def _dev(self, a, dtype):
        t = self.torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.device).contiguous()
        self._keep.append(t)
        return t

    def _host(self, a):
        # Host arrays are read by the library during the create/set_obs call only, but a pointer
        # stored in a ctypes Structure field does NOT keep its numpy array alive: hold them here
        # until the call has returned (_release_host).
        a = np.ascontiguousarray(a, dtype=np.float64)
        self._host_keep.append(a)
        return a

    def _release_host(self):
        self._host_keep = []

    def _model_desc(self, net):
        d = _lib.ModelDesc()
        layers = net["layers"]
        if not (2 <= len(layers) <= _lib.PAYNE_MAX_LAYERS):
            raise ValueError("spectral net must have 2..8 layers")
        d.n_layers = len(layers)
        for i, (W, b, act) in enumerate(layers):
            Wt, bt = self._dev(W, self.torch.float32), self._dev(b, self.torch.float32)
            d.layers[i] = _lib.Layer(Wt.data_ptr(), bt.data_ptr(), W.shape[1], W.shape[0], act)
        d.n_labels = layers[0][0].shape[1]
        d.xmin = _dptr(self._host(net["xmin"]))
        d.xmax = _dptr(self._host(net["xmax"]))
        d.npix = layers[-1][0].shape[0]
        d.wavelength = _dptr(self._host(net["wavelength"]))
        d.resolution = float(net["resolution"])
        self.npix = d.npix
        self.n_labels = d.n_labels
        self.wavelength = np.asarray(net["wavelength"], dtype=np.float64)
        return d

    def _obs_desc(self, wave, flux=None, eflux=None):
        d = _lib.ObsDesc()
        wave = self._host(wave)
        d.nobs = len(wave)
        d.wave = _dptr(wave)
        if flux is not None:
            d.flux = _dptr(self._host(flux))
            d.eflux = _dptr(self._host(eflux))
        self.nobs = d.nobs
        self.obs_wave = wave.copy()
        return d

    def _phot_desc(self, phot, obs_phot):
        d = _lib.PhotDesc()
        F, H = phot["w1"].shape[0], phot["w1"].shape[1]
        d.n_filters, d.hidden = F, H
        f32 = self.torch.float32
        d.w1 = self._dev(phot["w1"], f32).data_ptr()
        d.b1 = self._dev(np.reshape(phot["b1"], (F, H)), f32).data_ptr()
        d.w2 = self._dev(phot["w2"], f32).data_ptr()
        d.b2 = self._dev(np.reshape(phot["b2"], (F, H)), f32).data_ptr()
        d.w3 = self._dev(np.reshape(phot["w3"], (F, H)), f32).data_ptr()
        d.b3 = self._dev(np.reshape(phot["b3"], (F,)), f32).data_ptr()
        d.xmin = _dptr(self._host(phot["xmin"]))
        d.xmax = _dptr(self._host(phot["xmax"]))
        hiav = phot.get("hiav")
        if hiav is None:
            hiav = highav_coefficients(phot["filters"])
        d.hiav = _dptr(self._host(hiav))
        if obs_phot is not None:
            if list(obs_phot.keys()) != list(phot["filters"]):
                raise ValueError("obs_phot keys must match the photometric nets' filter order")
            d.obs_mag = _dptr(self._host([v[0] for v in obs_phot.values()]))
            d.obs_err = _dptr(self._host([v[1] for v in obs_phot.values()]))
        self.n_filters = F
        self.filters = list(phot["filters"])
        return d

    # -- calls -----------------------------------------------------------------
    def _err(self, rc, what):
        raise RuntimeError("%s failed (%d): %s" % (what, rc, self.lib.payne_last_error(self._ctx).decode()))

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _theta(self, theta, ncols):
        torch = self.torch
        if isinstance(theta, torch.Tensor):
            t = theta.to(device=self.device, dtype=torch.float64).contiguous()
        else:
            t = torch.as_tensor(np.ascontiguousarray(theta, dtype=np.float64)).to(self.device)
        if t.dim() != 2 or t.shape[1] != ncols:
            raise ValueError("theta must be [B, %d], got %s" % (ncols, tuple(t.shape)))
        return t

    def set_obs(self, wave, flux=None, eflux=None):
        """Re-bind the observed grid (payne_ctx_set_obs)."""
        d = self._obs_desc(wave, flux, eflux)
        rc = self.lib.payne_ctx_set_obs(self._ctx, C.byref(d))
        self._release_host()
        if rc != 0:
            self._err(rc, "payne_ctx_set_obs")

    def smooth_batch(self, spectra, theta, stage=2, fwhm_R=False):
        """The broadening stages on caller-supplied spectra [B, npix] (full flux on the model grid):
        payne_smooth_batch.  Returns a fp32 device tensor [B, npix | nobs]."""
        t = self._theta(theta, self.ncols)
        B = t.shape[0]
        sp = self.torch.as_tensor(np.ascontiguousarray(spectra, dtype=np.float32)).to(self.device).reshape(B, self.npix)
        n_out = self.npix if stage < 2 else self.nobs          # (stage 4: rotational broadening onto the observed grid)
        out = self.torch.empty((B, n_out), dtype=self.torch.float32, device=self.device)
        for s in range(0, B, self.b_max):
            n = min(self.b_max, B - s)
            rc = self.lib.payne_smooth_batch(self._ctx, sp[s:s + n].data_ptr(), self.npix, t[s:s + n].data_ptr(), n, int(stage),
                                             _lib.F_FWHM_R if fwhm_R else 0, out[s:s + n].data_ptr(), n_out, self._stream())
            if rc != 0:
                self._err(rc, "payne_smooth_batch")
        return out

    def set_lsf(self, lsf, wave=None):
        """Bind an LSF vector (dispersion per pixel of the bound observed grid, or -- with `wave` -- at wavelengths of its
        own: payne_ctx_set_lsf_on) or, with None, remove it (payne_ctx_set_lsf).  While set, theta's Inst_R column is ignored."""
        if lsf is None:
            rc = self.lib.payne_ctx_set_lsf(self._ctx, None, 0)
        elif wave is not None:
            a = np.ascontiguousarray(lsf, dtype=np.float64)
            w = np.ascontiguousarray(wave, dtype=np.float64)
            if len(a) != len(w):
                raise ValueError("the LSF vector and its wavelengths differ in length")
            rc = self.lib.payne_ctx_set_lsf_on(self._ctx, w.ctypes.data_as(C.POINTER(C.c_double)), a.ctypes.data_as(C.POINTER(C.c_double)), len(a))
        else:
            a = np.ascontiguousarray(lsf, dtype=np.float64)
            rc = self.lib.payne_ctx_set_lsf(self._ctx, a.ctypes.data_as(C.POINTER(C.c_double)), len(a))
        if rc != 0:
            self._err(rc, "payne_ctx_set_lsf")

    def set_continuum(self, net):
        """Bind a continuum network (normalised like the spectral net by nnio) or, with None, remove it
        (payne_ctx_set_continuum)."""
        if net is None:
            rc = self.lib.payne_ctx_set_continuum(self._ctx, None)
            self.npix_cont = 0
        else:
            npix, nlab, wave = self.npix, self.n_labels, self.wavelength
            d = self._model_desc(net)                      # (overwrites the three attributes above)
            self.npix_cont = d.npix
            self.npix, self.n_labels, self.wavelength = npix, nlab, wave
            rc = self.lib.payne_ctx_set_continuum(self._ctx, C.byref(d))
        self._release_host()
        if rc != 0:
            self._err(rc, "payne_ctx_set_continuum")

    def make_theta(self, B):
        """A NaN-filled [B, ncols] device tensor (NaN = parameter absent)."""
        return self.torch.full((B, self.ncols), float("nan"), dtype=self.torch.float64, device=self.device)

    def lnlike_batch(self, theta, out=None):
        """lnL [B] (fp64 device tensor) for theta [B, ncols]; enqueues on the
        current torch stream and does not synchronise."""
        t = self._theta(theta, self.ncols)
        B = t.shape[0]
        if out is None:
            out = self.torch.empty(B, dtype=self.torch.float64, device=self.device)
        for s in range(0, B, self.b_max):
            n = min(self.b_max, B - s)
            rc = self.lib.payne_lnlike_batch(self._ctx, t[s:s + n].data_ptr(), n, out[s:s + n].data_ptr(), self._stream())
            if rc != 0:
                self._err(rc, "payne_lnlike_batch")
        return out

    def predict_batch(self, theta, stage=2, fwhm_R=False):
        """Model spectra [B, npix|nobs] fp32 device tensor (payne_predict_batch)."""
        t = self._theta(theta, self.ncols)
        B = t.shape[0]
        n_out = self.npix if stage < 2 else (self.npix_cont if stage == 4 else self.nobs)
        out = self.torch.empty((B, n_out), dtype=self.torch.float32, device=self.device)
        for s in range(0, B, self.b_max):
            n = min(self.b_max, B - s)
            rc = self.lib.payne_predict_batch(self._ctx, t[s:s + n].data_ptr(), n, int(stage),
                                              _lib.F_FWHM_R if fwhm_R else 0, out[s:s + n].data_ptr(), n_out,
                                              self._stream())
            if rc != 0:
                self._err(rc, "payne_predict_batch")
        return out

    def sed_batch(self, pars):
        """Magnitudes [B, F] fp64 for pars [B, 9] = logt,logg,feh,afe,av,rv,logl,dist,logA."""
        t = self._theta(pars, 9)
        B = t.shape[0]
        out = self.torch.empty((B, self.n_filters), dtype=self.torch.float64, device=self.device)
        for s in range(0, B, self.b_max):
            n = min(self.b_max, B - s)
            rc = self.lib.payne_sed_batch(self._ctx, t[s:s + n].data_ptr(), n, out[s:s + n].data_ptr(), self._stream())
            if rc != 0:
                self._err(rc, "payne_sed_batch")
        return out

    def bc_batch(self, x):
        """Bolometric corrections [B, F] fp64 for x [B, 6] = Teff,logg,feh,afe,av,rv."""
        t = self._theta(x, 6)
        B = t.shape[0]
        out = self.torch.empty((B, self.n_filters), dtype=self.torch.float64, device=self.device)
        for s in range(0, B, self.b_max):
            n = min(self.b_max, B - s)
            rc = self.lib.payne_bc_batch(self._ctx, t[s:s + n].data_ptr(), n, out[s:s + n].data_ptr(), self._stream())
            if rc != 0:
                self._err(rc, "payne_bc_batch")
        return out

    def profile(self, enable):
        """Start/stop HIP-event timing of every kernel the batch calls launch."""
        rc = self.lib.payne_profile(self._ctx, 1 if enable else 0)
        if rc != 0:
            self._err(rc, "payne_profile")

    def profile_read(self):
        """{kind: (total_ms, launches)} for kinds dense_out, post, sed, dense_hidden."""
        out = {}
        for k, name in enumerate(("dense_out", "post", "sed", "dense_hidden")):
            ms, n = C.c_double(0.0), C.c_longlong(0)
            rc = self.lib.payne_profile_read(self._ctx, k, C.byref(ms), C.byref(n))
            if rc != 0:
                self._err(rc, "payne_profile_read")
            out[name] = (ms.value, n.value)
        return out

    def kernels_used(self):
        """{'out' | 'post' | 'sed' | 'hidden': kernel name, 'rows': 'frequency' | 'pixels'} as launched by the last batch call
        (payne_last_kernel)."""
        return {name: self.lib.payne_last_kernel(self._ctx, k).decode().strip("()") for k, name in enumerate(("out", "post", "sed", "hidden", "rows"))}

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self.torch.cuda.synchronize(self.device)
            self.lib.payne_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
