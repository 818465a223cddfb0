"""Prior transforms and additive ln-priors of the fit (host side, numpy/scipy).

Same constructor and methods as ``Payne.fitting.prior.prior``
(Payne/fitting/prior.py:5-465): ``priortrans(u)`` maps a unit-cube vector to
parameters in ``fitpars_i`` order, ``lnpriorfn(pars)`` adds the optional
'gaussian'/'uniform' priors.  New here: ``priortrans_batch`` / ``lnprior_batch``
over a [B, ndim] array, which is what the batched sampler calls once per step.

Out of scope: the brutus-derived AdvancedPriors (IMF, GAL,
VROT, VTOT, AngDia) -- requesting one raises NotImplementedError instead of being
silently ignored.
"""
import numpy as np
from scipy.stats import norm, truncnorm, expon, truncexpon

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
        self.defaultpars = {k: list(v) for k, v in DEFAULT_RANGES.items()}
        self.priordict = {k: {} for k in _KINDS}
        self.additionalpriors = {}
        for name, spec in inpriordict.items():
            if name == 'blaze_coeff':
                self.polycoefarr = spec
            elif name in ('IMF', 'GAL', 'VROT', 'VTOT', 'AngDia'):
                raise NotImplementedError(
                    "the %r advanced prior (Payne/fitting/advancedpriors.py) is outside this build's "
                    "hot-path scope" % name)
            else:
                for kind, val in spec.items():
                    if kind.startswith('pv_') and kind[3:] in _KINDS:
                        self.priordict[kind[3:]][name] = val
                    elif kind == 'fixed':
                        self.additionalpriors.setdefault(name, {})[kind] = val
                    else:
                        self.additionalpriors.setdefault(name, {})[kind] = val
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
        for name in _ISO_NAMES:
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
        """prior.py:274-377 (advanced priors excluded): 0.0 unless 'gaussian' /
        'uniform' entries were given; -inf outside a 'uniform' box."""
        if isinstance(pars, list):
            parsdict = {pp: vv for pp, vv in zip(self.fitpars_i, pars)}
        else:
            parsdict = pars
        for kk in self.fixedpars.keys():
            parsdict[kk] = self.fixedpars[kk]
        if len(self.additionalpriors) == 0:
            return 0.0
        total = 0.0
        if self.spec_bool:
            total = total + self.lnprior_spec(parsdict)
        if self.phot_bool:
            total = total + self.lnprior_phot(parsdict)
        return total

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
        if len(self.additionalpriors) == 0:
            return np.zeros(theta.shape[0])
        return np.array([self.lnpriorfn(list(t)) for t in theta], dtype=np.float64)
