"""The physically motivated priors the fit can switch on (priordict keys IMF, GAL, VROT, VTOT,
AngDia): the functions of Payne/fitting/advancedpriors.py that Payne/fitting/prior.py reaches --
``imf_lnprior`` (:93-137), the number-density part of ``gal_lnprior`` (:410-573) with
``logn_disk`` (:241-270) / ``logn_halo`` (:272-328) behind ``gal_ppf`` (:665-670),
``vrot_lnprior`` (:691-733), ``Vtot_lnprior`` (:736-756) and ``AngDia_lnprior`` (:759-773).

Host side, O(1) per candidate; evaluated over a batch at a time.  The galactic model uses the
observer-based Cartesian construction of the reference's default branch (`coords == []`), which
needs no astropy.
"""
import numpy as np

__all__ = ["AdvancedPriors"]


def _logsumexp3(a, b, c):
    m = np.maximum(np.maximum(a, b), c)
    return m + np.log(np.exp(a - m) + np.exp(b - m) + np.exp(c - m))


class AdvancedPriors(object):
    def __init__(self, **kwargs):
        self.l = kwargs.get('l', 0.0)
        self.b = kwargs.get('b', 0.0)
        lp, bp = np.deg2rad(self.l), np.deg2rad(self.b)
        self.Xp, self.Yp, self.Zp = np.cos(lp) * np.cos(bp), np.sin(lp) * np.cos(bp), np.sin(bp)
        self.sol_X, self.sol_Z = 8.3, -27.0 / 1000.0
        self.mindist = kwargs.get('mindist', 0.001)
        self.maxdist = kwargs.get('maxdist', 200.0)
        # tabulated distance prior on a log-spaced grid (kpc); its weights are the density values
        # themselves, with no dx factor, as the reference forms them (:62-64, quantiles.py)
        self.distarr = np.logspace(np.log10(self.mindist), np.log10(self.maxdist), 10000)
        self.distnormfactor = np.exp(self.gal_lnprior(self.distarr))
        cdf = np.cumsum(self.distnormfactor)[:-1]
        cdf = cdf / cdf[-1]
        self._cdf = np.append(0, cdf)
        self.angdia = kwargs.get('AngDia', [1.0, 1.0])

    # ---- initial mass function -----------------------------------------------------------
    def imf_lnprior(self, mgrid, alpha_low=1.3, alpha_high=2.3, mass_break=0.5):
        """Kroupa-like broken power law; -inf at or below the hydrogen-burning limit 0.08."""
        m = np.atleast_1d(np.asarray(mgrid, dtype=np.float64))
        lnprior = np.full(m.shape, -np.inf)
        low = (m <= mass_break) & (m > 0.08)
        lnprior[low] = -alpha_low * np.log(m[low])
        high = m > mass_break
        lnprior[high] = -alpha_high * np.log(m[high]) + (alpha_high - alpha_low) * np.log(mass_break)
        norm = mass_break ** (1. - alpha_low) / (alpha_high - 1.) + 0.08 ** (1. - alpha_low) / (alpha_low - 1.) \
            - mass_break ** (1. - alpha_low) / (alpha_low - 1.)
        return lnprior - np.log(norm)

    # ---- galactic number density along the line of sight -------------------------------------
    @staticmethod
    def logn_disk(R, Z, R_solar=8.2, Z_solar=0.025, R_scale=2.6, Z_scale=0.3):
        return -((R - R_solar) / R_scale + (np.abs(Z) - np.abs(Z_solar)) / Z_scale)

    @staticmethod
    def logn_halo(R, Z, R_solar=8.2, Z_solar=0.025, R_smooth=0.5, eta=4.2, q_ctr=0.2, q_inf=0.8, r_q=6.0):
        r = np.sqrt(R ** 2 + Z ** 2)
        q = q_inf - (q_inf - q_ctr) * np.exp(1. - np.sqrt(r ** 2 + r_q ** 2) / r_q)
        Reff = np.sqrt(R ** 2 + (Z / q) ** 2 + R_smooth ** 2)
        q_solar = q_inf - (q_inf - q_ctr) * np.exp(1. - np.sqrt(R_solar ** 2 + Z_solar ** 2 + r_q ** 2) / r_q)
        Reff_solar = np.sqrt(R_solar ** 2 + (Z_solar / q_solar) + R_smooth ** 2)       # (Z/q not squared: as the reference, :322)
        return -eta * np.log(Reff / Reff_solar)

    def gal_lnprior(self, dists, return_components=False):
        """ln of (thin disk + thick disk + halo) number density times r^2, distances in kpc."""
        dists = np.atleast_1d(np.asarray(dists, dtype=np.float64))
        if np.any(dists <= 0.0):
            return (-np.inf, {}) if return_components else -np.inf
        vol = 2. * np.log(dists + 1e-300)
        X, Y = dists * self.Xp - self.sol_X, dists * self.Yp
        Z = dists * self.Zp - self.sol_Z
        R = np.hypot(X, Y)
        thin = self.logn_disk(R, Z, R_scale=2.6, Z_scale=0.3) + vol
        thick = self.logn_disk(R, Z, R_scale=2.0, Z_scale=0.9) + vol + np.log(0.04)
        halo = self.logn_halo(R, Z) + vol + np.log(0.005)
        lnprior = _logsumexp3(thin, thick, halo)
        if return_components:
            return lnprior, {'number_density': [thin, thick, halo],
                             'lnprior': [thin - lnprior, thick - lnprior, halo - lnprior]}
        return lnprior

    def gal_ppf(self, u):
        """Unit interval -> distance (kpc): weighted quantile of the tabulated prior."""
        d = np.interp(np.atleast_1d(u), self._cdf, self.distarr)
        return d[0] if np.ndim(u) == 0 else d

    # ---- rotation, total velocity, angular diameter ---------------------------------------------
    @staticmethod
    def vrot_lnprior(vrot=1.5, mass=1.0, eep=350, logg=4.44, giant=None, dwarf=None):
        """Sigmoid penalty on fast rotation: a/(1 + n exp(-(vrot - c))) with (a, c, n) chosen by
        the Kraft break (mass > 1.25), giants (logg < 3.5 or eep > 450) or dwarfs."""
        giant = giant or {'a': -10.0, 'c': 7.0, 'n': 1.0}
        dwarf = dwarf or {'a': -10.0, 'c': 10.0, 'n': 0.4}
        vrot, mass, eep, logg = np.broadcast_arrays(np.asarray(vrot, float), np.asarray(mass, float),
                                                    np.asarray(eep, float), np.asarray(logg, float))
        hot = mass > 1.25
        gi = (~hot) & ((logg < 3.5) | (eep > 450))
        a = np.where(hot, -1.0, np.where(gi, giant['a'], dwarf['a']))
        c = np.where(hot, 100.0, np.where(gi, giant['c'], dwarf['c']))
        n = np.where(hot, 1.0, np.where(gi, giant['n'], dwarf['n']))
        return a / (1.0 + n * np.exp(-(vrot - c)))

    @staticmethod
    def Vtot_lnprior(vrad=0.0, mu=0.0, dist=1e+6):
        Vtot = np.sqrt(vrad ** 2.0 + (mu * 4.74 * dist) ** 2.0)
        return -10.0 / (1.0 * np.exp(-(Vtot - 600.0)))

    def AngDia_lnprior(self, rad=1.0, dist=1.0):
        d = dist * 4.435E+7
        pred = np.rad2deg(2.0 * np.arcsin(rad / d)) * 3600000.0
        return -0.5 * ((pred - self.angdia[0]) ** 2.0) / (self.angdia[1] ** 2.0)
