"""Small host-side helpers of the fit (numpy), mirroring the two functions of
Payne/fitting/fitutils.py that the sampler path touches."""
import numpy as np
from numpy.polynomial.chebyshev import chebval

__all__ = ["polycalc", "airtovacuum", "vacuumtoair"]


def polycalc(coef, inwave):
    """Chebyshev blaze on the wavelength range rescaled to [-1, 1]
    (Payne/fitting/fitutils.py:11-20).  The batched likelihood evaluates the same
    series on the GPU (post_core.hpp, phase_obs); this host version backs the
    public helper."""
    inwave = np.asarray(inwave, dtype=np.float64)
    span = inwave - inwave.min()
    return chebval(2.0 * (span / span.max()) - 1.0, coef)


def airtovacuum(inwave):
    """Ciddor (1996) air -> vacuum (Payne/fitting/fitutils.py:22-37); applied once
    to the observed wavelengths when inputdict['spec']['convertair'] is set."""
    mu = np.asarray(inwave, dtype=np.float64) * 1e-4          # micron
    s2 = 1.0 / mu ** 2.0
    refr = 0.0 + (5.792105e-2 / (238.0185 - s2)) + (1.67917e-3 / (57.362 - s2))
    return (mu * (refr + 1)) * 1e4


def vacuumtoair(inwave):
    """Payne/fitting/fitutils.py:39-44."""
    inwave = np.asarray(inwave, dtype=np.float64)
    s2 = ((10 ** 4) / inwave) ** 2.0
    return inwave / (1.0 + 0.0000834254 + 0.02406147 / (130.0 - s2) + 0.00015998 / (38.9 - s2))
