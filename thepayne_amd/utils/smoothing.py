"""``Payne.utils.smoothing.smoothspec`` (Payne/utils/smoothing.py:19-169) as a free function, on the GPU.

The reference's ``getspec`` calls this function; here the broadening stages live inside the likelihood kernels and the
function form exists for callers that import it directly.  Every branch is the one ``PayneSpecPredict.smoothspec`` runs
(predict/_spec.py): the FFT branches through the likelihood's own kernels, ``fftsmooth=False`` and smoothtype 'lambda'
through payne_smooth_direct.  There is no CPU path."""
import numpy as np

from ..predict._spec import PayneSpecPredict

__all__ = ["smoothspec", "ckms", "sigma_to_fwhm"]

ckms = 2.998e5              # smoothing.py:16
sigma_to_fwhm = 2.355       # smoothing.py:17


class _Smoother(PayneSpecPredict):
    """smoothspec needs no network: only the device the work runs on."""

    def __init__(self, device=None):
        self._dev = device

    def _device_index(self):
        if self._dev is None:
            import torch
            if not torch.cuda.is_available():
                raise RuntimeError("smoothspec needs a ROCm GPU (there is no CPU fallback)")
            self._dev = torch.cuda.current_device()
        return self._dev


_default = _Smoother()


def smoothspec(wave, spec, resolution=None, outwave=None, smoothtype="vel", fftsmooth=True,
               min_wave_smooth=0, max_wave_smooth=np.inf, **kwargs):
    """Same signature and meaning as the reference's (smoothing.py:19-84)."""
    return _default.smoothspec(wave, spec, resolution, outwave=outwave, smoothtype=smoothtype, fftsmooth=fftsmooth,
                               min_wave_smooth=min_wave_smooth, max_wave_smooth=max_wave_smooth, **kwargs)
