"""Prior transforms and additive ln-priors of the fit (host side, numpy/scipy).

Same constructor and methods as ``Payne.fitting.prior.prior``
(Payne/fitting/prior.py:5-465): ``priortrans(u)`` maps a unit-cube vector to
parameters in ``fitpars_i`` order, ``lnpriorfn(pars)`` adds the optional
'gaussian'/'uniform' priors.  New here: ``priortrans_batch`` / ``lnprior_batch``
over a [B, ndim] array, which is what the batched sampler calls once per step.

The physically motivated priors (priordict keys IMF, GAL, VROT, VTOT, AngDia;
fitting/advancedpriors.py) act exactly where the reference lets them: IMF and VROT add
to ``lnpriorfn`` (prior.py:288-336), GAL replaces the transform of ``Dist``
(prior.py:231-234).  VTOT sets ``pm_bool`` rather than ``vtot_bool`` in the reference
(prior.py:66-69), so its branch of ``lnpriorfn`` never runs, and ``AngDia`` is stored but
never evaluated (prior.py:119-120); both are accepted here with the same (absent) effect.
"""
import numpy as np
from scipy.stats import norm, truncnorm, expon, truncexpon

from .advancedpriors import AdvancedPriors

__all__ = ["prior"]

_SPEC_NAMES = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Inst_R', 'CarbonScale']
_ATM_NAMES = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]']
_ISO_NAMES = ['log(A)', 'log(R)', 'Av', 'Rv', 'Dist']
_KINDS = ('uniform', 'gaussian', 'tgaussian', 'exp', 'texp', 'loguniform')

# default box of every parameter (Payne/fitting/prior.py:95-110)
DEFAULT_RANGES = {
    'Teff': [3000.0, 17000.0], 'log(g)': [-1.0, 5.5], '[Fe/H]': [-4.0, 0.5], '[a/Fe]': [-0.2, 0.6],
    'Vrad': [-700.0, 700.0], 'Vrot': [0, 300.0], 'Inst_R': [10000.0, 60000.0],
    'log(A)': [-3.0, 7.0], 'log(R)': [-2.0, 3.0], 'Dist': [0.0, 100000.0],
    'Av': [0.0, 5.0], 'Rv': [2.0, 5.0], 'CarbonScale': [0.0, 2.0],
}


class prior(object):
    def __init__(self, fitargs, inpriordict, fitpars, runbools):
        self.fitargs = fitargs
        self.fixedpars = self.fitargs['fixedpars']
        self.fitpars_i = [pp for pp in fitpars[0] if fitpars[1][pp]]
        self.ndim = len(self.fitpars_i)
        self.spec_bool, self.phot_bool, self.modpoly_bool, self.photscale_bool = runbools[:4]
        self.imf_bool = self.gal_bool = self.vrot_bool = self.vtot_bool = False
        self.angdia_bool = 'AngDia' in inpriordict
        self.defaultpars = {k: list(v) for k, v in DEFAULT_RANGES.items()}
        self.priordict = {k: {} for k in _KINDS}
        self.additionalpriors = {}
        for name, spec in inpriordict.items():
            if name == 'blaze_coeff':
                self.polycoefarr = spec
            elif name == 'IMF':
                self.imf = spec['IMF_type']
                self.imf_bool = True
            elif name == 'GAL':
                self.gal_bool = True
                self.lb_coords = spec['lb_coords']
                if 'Dist' in inpriordict:
                    self.mindist, self.maxdist = inpriordict['Dist']['pv_uniform'][:2]
                else:
                    self.mindist, self.maxdist = 1.0, 200000.0
            elif name == 'VROT':
                self.vrot_bool = True
            elif name == 'VTOT':
                self.pmra, self.pmdec = spec['pmra'], spec['pmdec']
                self.pm_bool = True                                    # (not vtot_bool: prior.py:66-69)
            else:
                for kind, val in spec.items():
                    if kind.startswith('pv_') and kind[3:] in _KINDS:
                        self.priordict[kind[3:]][name] = val
                    else:
                        self.additionalpriors.setdefault(name, {})[kind] = val
        apargs = {}
        if self.gal_bool:
            apargs.update(l=self.lb_coords[0], b=self.lb_coords[1], mindist=self.mindist / 1000.0,
                          maxdist=self.maxdist / 1000.0)
        if self.angdia_bool:
            apargs['AngDia'] = inpriordict['AngDia']['gaussian']
        self.AP = AdvancedPriors(**apargs)
        self.advanced = self.imf_bool or self.gal_bool or self.vrot_bool
        # like the reference, a 'fixed' entry lands in additionalpriors (prior.py:84-88), which
        # switches off the early return of lnpriorfn; keep that behaviour.

    # ---- unit cube -> parameter ---------------------------------------------
    def _transform(self, name, u, kinds):
        """One parameter; ``u`` scalar or array.  Priority order and formulae of
        Payne/fitting/prior.py:151-178 (spec) / :236-270 (phot)."""
        pd = self.priordict
        for kind in kinds:
            if name not in pd[kind]:
                continue
            p = pd[kind][name]
            if kind == 'uniform':
                return (max(p) - min(p)) * u + min(p)
            if kind == 'gaussian':
                return norm.ppf(u, loc=p[0], scale=p[1])
            if kind == 'tgaussian':
                a, b = (p[0] - p[2]) / p[3], (p[1] - p[2]) / p[3]
                out = truncnorm.ppf(u, a, b, loc=p[2], scale=p[3])
                return np.where(out == np.inf, p[1], out) if np.ndim(out) else (p[1] if out == np.inf else out)
            if kind == 'exp':
                return expon.ppf(u, loc=p[0], scale=p[1])
            if kind == 'texp':
                b = (p[1] - p[0]) / p[2]
                out = truncexpon.ppf(u, b, loc=p[0], scale=p[2])
                return np.where(out == np.inf, p[1], out) if np.ndim(out) else (p[1] if out == np.inf else out)
            if kind == 'loguniform':
                # the reference calls an un-imported scipy `reciprocal` here (prior.py:266 -> NameError);
                # this is the distribution it names
                return np.exp(np.log(p[0]) + u * (np.log(p[1]) - np.log(p[0])))
        lo, hi = self.defaultpars[name]
        return (hi - lo) * u + lo

    def priortrans_spec(self, upars):
        out = {}
        for name in _SPEC_NAMES:
            if name in upars:
                out[name] = self._transform(name, upars[name], ('uniform', 'gaussian', 'tgaussian', 'exp', 'texp'))
        for name in upars:
            if 'pc' in name:                                           # prior.py:180-191
                if name == 'pc_0':
                    out[name] = (1.25 - 0.75) * upars[name] + 0.75
                else:
                    mu, sig = self.polycoefarr[int(name.split('_')[-1])][:2]
                    out[name] = ((mu + 5.0 * sig) - (mu - 5.0 * sig)) * upars[name] + (mu - 5.0 * sig)
        return out

    def priortrans_phot(self, upars):
        out = {}
        if not self.spec_bool:                                         # prior.py:199-229
            for name in _ATM_NAMES:
                if name in upars:
                    out[name] = self._transform(name, upars[name], ('uniform', 'gaussian', 'tgaussian', 'exp', 'texp'))
        iso = list(_ISO_NAMES)
        if self.gal_bool and 'Dist' in upars:                          # prior.py:231-234
            out['Dist'] = 1000.0 * self.AP.gal_ppf(upars['Dist'])
            iso.remove('Dist')
        for name in iso:
            if name in upars:
                # NB 'texp' for these parameters indexes a 4-vector with 3 shape arguments in the
                # reference (prior.py:259-261) and cannot run there; the 3-vector form is used.
                out[name] = self._transform(name, upars[name],
                                            ('uniform', 'gaussian', 'exp', 'tgaussian', 'texp', 'loguniform'))
        return out

    def _trans_dict(self, udict):
        out = {}
        if self.spec_bool:
            out.update(self.priortrans_spec(udict))
        if self.phot_bool:
            out.update(self.priortrans_phot(udict))
        return out

    def priortrans(self, upars):
        """u[ndim] in [0,1) -> list of parameters in fitpars_i order (prior.py:126-142)."""
        out = self._trans_dict({pp: vv for pp, vv in zip(self.fitpars_i, upars)})
        return [out[pp] for pp in self.fitpars_i]

    def priortrans_batch(self, U):
        """U[B, ndim] -> theta[B, ndim]; vectorised over the batch, same arithmetic."""
        U = np.asarray(U, dtype=np.float64)
        out = self._trans_dict({pp: U[:, i] for i, pp in enumerate(self.fitpars_i)})
        return np.column_stack([np.broadcast_to(out[pp], (U.shape[0],)) for pp in self.fitpars_i])

    # ---- additive ln-priors -----------------------------------------------------
    def lnpriorfn(self, pars):
        """prior.py:274-377: the IMF / VROT terms, plus 'gaussian' / 'uniform' entries if any
        were given (-inf outside a 'uniform' box)."""
        if isinstance(pars, list):
            parsdict = {pp: vv for pp, vv in zip(self.fitpars_i, pars)}
        else:
            parsdict = pars
        for kk in self.fixedpars.keys():
            parsdict[kk] = self.fixedpars[kk]
        total = self._advanced(parsdict) if (self.imf_bool or self.vrot_bool) else 0.0
        if len(self.additionalpriors) == 0:
            return total
        if self.spec_bool:
            total = total + self.lnprior_spec(parsdict)
        if self.phot_bool:
            total = total + self.lnprior_phot(parsdict)
        return total

    def _advanced(self, parsdict):
        """prior.py:286-336.  Masses come from log(g) and log(R) when the fit carries no
        'initial_Mass'; the two branches use different zero points, as the reference does."""
        adv = 0.0
        if self.imf_bool:
            if 'initial_Mass' not in parsdict:
                # (a fit without log(R) raises KeyError here, as the reference does)
                if np.isfinite(parsdict['log(g)']) and np.isfinite(parsdict['log(R)']):
                    mass = 10.0 ** (parsdict['log(g)'] + 2.0 * parsdict['log(R)'] - 4.437)
                else:
                    raise NameError("IMF prior: no mass can be formed from non-finite log(g) / log(R)")
            else:
                mass = parsdict['initial_Mass']
            adv += float(self.AP.imf_lnprior(mass)[0])
        if self.vrot_bool:
            mass, eep = parsdict.get('initial_Mass'), parsdict.get('EEP')
            if mass is None:
                if 'log(A)' in parsdict:
                    mass = 1.0
                elif np.isfinite(parsdict['log(g)']) and np.isfinite(parsdict['log(R)']):
                    mass = 10.0 ** (parsdict['log(g)'] + 2.0 * parsdict['log(R)'])
                else:
                    mass = 1.0
                eep = 350
            adv += float(self.AP.vrot_lnprior(vrot=parsdict['Vrot'], mass=mass, eep=eep, logg=parsdict['log(g)']))
        return adv

    def _apply_additional(self, names, values):
        lnp = 0.0
        for kk, spec in self.additionalpriors.items():
            if kk not in names or kk not in values:
                # (a label prior in a joint fit is applied once, by lnprior_spec; the reference
                # raises KeyError in lnprior_phot there, prior.py:431-435 vs :451)
                continue
            if 'gaussian' in spec:
                lnp += -0.5 * (((values[kk] - spec['gaussian'][0]) ** 2.0) / (spec['gaussian'][1] ** 2.0))
            if 'uniform' in spec:
                if (values[kk] < spec['uniform'][0]) or (values[kk] > spec['uniform'][1]):
                    return -np.inf
            if 'beta' in spec:
                raise IOError('Beta Prior not implimented yet!!!')
            if 'log-normal' in spec:
                raise IOError('Log-Normal Prior not implimented yet!!!')
        return lnp

    def lnprior_spec(self, pardict, verbose=True):
        """prior.py:379-402."""
        return self._apply_additional(_SPEC_NAMES, pardict)

    def lnprior_phot(self, pardict, verbose=True):
        """prior.py:425-465 (incl. the derived 'Parallax' = 1000/Dist).  As in the
        reference, atmosphere labels are only visible here in a photometry-only fit."""
        vals = {}
        if not self.spec_bool:
            for name in _ATM_NAMES:
                vals[name] = pardict[name]
        for name in ('log(R)', 'Dist', 'log(A)', 'Av'):
            if name in self.fitpars_i:
                vals[name] = pardict[name]
        if 'Dist' in self.fitpars_i:
            vals['Parallax'] = 1000.0 / pardict['Dist']
        return self._apply_additional(_ATM_NAMES + ['log(R)', 'Dist', 'log(A)', 'Av', 'Parallax'], vals)

    def lnprior_batch(self, theta):
        """theta[B, ndim] -> lnprior[B] (fp64)."""
        theta = np.asarray(theta, dtype=np.float64)
        if len(self.additionalpriors) == 0 and not (self.imf_bool or self.vrot_bool):
            return np.zeros(theta.shape[0])
        return np.array([self.lnpriorfn(list(t)) for t in theta], dtype=np.float64)
