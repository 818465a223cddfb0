"""High-extinction approximation: mirrors Payne/predict/highred.py (highAv)."""
import numpy as np

from ..engine import highav_coefficients


class highAv(object):
    def __init__(self, filters):
        self.Avlist = [list(r) for r in highav_coefficients(list(filters))]

    def getAvaprox(self, Av, Rv, pars):
        a1, b1, a2, b2, c2 = pars
        return a1 + b1 * Av * (a2 + b2 * Rv + c2 * Rv ** 2.0)

    def calc(self, BC0, Av, Rv):
        """BC(Av>=5) = BC(Av=0) - offset (highred.py:23-25); host helper -- the batched
        path applies the same formula inside payne_sed_kernel."""
        return np.array([b - self.getAvaprox(Av, Rv, p) for p, b in zip(self.Avlist, BC0)])
