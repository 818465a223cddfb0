"""Random API calls through the HIP path against the oracle: the generator of
oracle/fuzz_vs_reference.py (which checks the oracle against the reference itself) pointed at
PayneSpecPredict.getspec / GenMod.genspec / likelihood.lnlikefn.  Tolerances of SURVEY 8(d):
|dflux| <= 1e-6 (2e-6 where the blaze multiplies), NaN patterns identical, |dlnL| <= 2e-6 |lnL| + 5e-3."""
import numpy as np
import pytest

import oracle as O
from thepayne_amd import synth, nnio
from helpers import lnl_tol

pytestmark = pytest.mark.gpu
SEED0 = int(__import__("os").environ.get("PAYNE_FUZZ_SEED", "0"))      # another stream of random calls: PAYNE_FUZZ_SEED=n


def _save(tmp_path, raw, name):
    path = str(tmp_path / name)
    nnio.save_npz(path, {k: (np.array([v]) if k == "resolution" else v) for k, v in raw.items() if k != "kind"})
    return path


@pytest.mark.parametrize("D,seed,cont", [(4, 0, False), (4, 1, False), (5, 2, False), (4, 3, True), (4, 4, False), (4, 5, False),
                                         (4, 6, False), (4, 7, False)])
def test_random_getspec_calls(tmp_path, D, seed, cont):
    from thepayne_amd.predict.ystpred import PayneSpecPredict
    rng = np.random.default_rng(100 + seed + 1000 * SEED0)
    net = synth.make_yst_net(npix=[512, 700, 1024, 600, 4096, 3000, 20000, 40000][seed], H=32, seed=20 + seed, D=D, line_depth=0.3)
    cnet = None
    if cont:                                           # continuum network on its own, coarser grid (Cnnpath)
        w = net["wavelength"]
        cnet = synth.make_cont_net(npix=311, lam_lo=w[0] - 1.0, lam_hi=w[-1] + 1.0)
        PP = PayneSpecPredict(nnpath=_save(tmp_path, net, "n.npz"), Cnnpath=_save(tmp_path, cnet, "c.npz"), NNtype='YST1')
    else:
        PP = PayneSpecPredict(nnpath=_save(tmp_path, net, "n.npz"), NNtype='YST1')
    wave = net["wavelength"]
    alias = {"Teff": ["Teff", "logt"], "logg": ["logg", "log(g)"], "feh": ["feh", "[Fe/H]"],
             "afe": ["afe", "aFe", "[a/Fe]", "[alpha/Fe]"]}
    worst = 0.0
    for it in range(120 if len(net["wavelength"]) < 10000 else 40):       # (the big sizes: the global-workspace kernel)
        kw, canon = {}, {}
        lab = dict(Teff=rng.uniform(4000, 7500), logg=rng.uniform(0.5, 5.2), feh=rng.uniform(-2, 0.4), afe=rng.uniform(-0.1, 0.5))
        for k, v in lab.items():
            if rng.uniform() < 0.15:
                continue
            name = alias[k][rng.integers(len(alias[k]))]
            kw[name] = np.log10(v) if name == "logt" else v
            canon[k] = 10.0 ** kw[name] if name == "logt" else v
        if D == 5 or rng.uniform() < 0.2:
            kw['vmic'] = rng.uniform(0.5, 2.5) if D == 5 else np.nan
        if rng.uniform() < 0.75:
            kw['rot_vel'] = [0.0, 1e-3, rng.uniform(0.2, 60.0)][rng.integers(3)]
        if rng.uniform() < 0.75:
            kw['rad_vel'] = [0.0, rng.uniform(-300, 300)][rng.integers(2)]
        nobs = int(rng.integers(50, 400))
        lo, hi = np.sort(rng.uniform(wave[0] - 2.0, wave[-1] + 2.0, 2))
        outwave = np.linspace(lo, max(hi, lo + 1.0), nobs) if rng.uniform() < 0.7 else None
        if outwave is not None:
            kw['outwave'] = outwave
        mode = rng.integers(7)
        if mode in (1, 3):
            kw['inst_R'] = float(rng.uniform(8000, 60000))
        elif mode == 2:
            kw['inst_R'] = [np.nan, 0.0, -5.0, float(net["resolution"]) * 1.2][rng.integers(4)]
        elif mode == 4 and outwave is not None:        # an LSF vector (any spectrum length: LDS form up to 8192 pixels, global form above)
            x = np.linspace(-0.5, 0.5, nobs)
            kw['inst_R'] = 0.08 * (1.0 + rng.uniform(-0.5, 0.5) * x + rng.uniform(0, 0.5) * x ** 2)
        canon.update({k: v for k, v in kw.items() if k in ('vmic', 'rot_vel', 'rad_vel', 'inst_R', 'outwave')})
        try:
            with np.errstate(all="ignore"):
                w_o, f_o = O.getspec(net, cnet=cnet, **canon)
        except ValueError:
            # an output grid that misses the shifted model entirely: numpy reduces an empty mask and raises
            # (the reference does); the documented deviation of the build is an all-NaN spectrum
            assert np.isnan(PP.getspec(**kw)[1]).all()
            continue
        w, f = PP.getspec(**kw)
        np.testing.assert_allclose(w, w_o, rtol=1e-15)
        assert np.array_equal(np.isnan(f), np.isnan(f_o)), (it, {k: v for k, v in kw.items() if np.ndim(v) == 0})
        ok = ~np.isnan(f_o)
        if ok.any():
            err = np.abs(f[ok] - f_o[ok]).max()
            worst = max(worst, err)
            tol = 2e-6 if isinstance(kw.get('inst_R'), np.ndarray) else 1e-6
            if 0.0 < kw.get('rot_vel', 0.0) < 0.01 and len(wave) > 10000:
                # a rotation far below one pixel: the reference's taper J1(u)/u - 3cos(u)/2u^2 + 3sin(u)/2u^3 cancels
                # catastrophically in its lowest bins (u_k ~ 1e-7 k at 32 768+ points: terms of 1e14 subtract to ~1, a
                # few per cent of noise in ITS lowest bins, 1e-6 .. 1e-5 in its spectrum); the kernel interpolates the
                # analytic value
                tol = 5e-5
            assert err <= tol, (it, err, {k: v for k, v in kw.items() if np.ndim(v) == 0})
    assert worst > 0.0


def test_random_likelihood_setups(tmp_path):
    from thepayne_amd.fitting.likelihood import likelihood
    rng = np.random.default_rng(77 + 1000 * SEED0)
    net = synth.make_yst_net(npix=512, H=32, seed=24, line_depth=0.3)
    path = _save(tmp_path, net, "like.npz")
    phot = synth.make_phot_nets()
    obs = synth.obs_grid(net["wavelength"], 300, inset=1.0)
    _, clean = O.getspec(net, Teff=5770.0, logg=4.44, feh=0.0, afe=0.0, rad_vel=10.0, rot_vel=3.0, inst_R=2.355 * 28800.0, outwave=obs)
    flux = clean + rng.normal(0, 0.01, len(obs))
    eflux = np.full(len(obs), 0.01)
    obs_phot = {f: [5.0 + 0.1 * i, 0.05] for i, f in enumerate(phot["filters"])}
    ALL = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Vmic', 'Inst_R', 'log(R)', 'Dist', 'log(A)', 'Av', 'Rv', 'CarbonScale']
    R = {'Teff': (4500, 7000), 'log(g)': (1.0, 5.0), '[Fe/H]': (-1.5, 0.4), '[a/Fe]': (-0.1, 0.5), 'Vrad': (-50, 50),
         'Vrot': (0, 30), 'Inst_R': (20000, 40000), 'log(R)': (-0.5, 1.0), 'Dist': (10, 3000), 'log(A)': (-2, 3),
         'Av': (0, 4.5), 'Rv': (2.2, 4.8)}
    ncmp = 0
    for it in range(24):
        spec = rng.uniform() < 0.8
        has_phot = (not spec) or rng.uniform() < 0.5
        photscale = bool(rng.uniform() < 0.5)
        modpoly = bool(spec and rng.uniform() < 0.4)
        on = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]']
        if spec:
            on += ['Vrad', 'Vrot', 'Inst_R']
        if has_phot:
            on += (['log(A)'] if photscale else ['log(R)', 'Dist']) + ['Av']
            if rng.uniform() < 0.4:
                on.append('Rv')
        names = list(ALL) + (['pc_0', 'pc_1', 'pc_2'] if modpoly else [])
        fixed = {}
        for name in rng.permutation(on)[:int(rng.integers(0, 3))]:
            if name != 'Av':
                fixed[str(name)] = float(rng.uniform(*R[str(name)]))
        fitpars = [names, {p: ((p in on and p not in fixed) or p.startswith('pc_')) for p in names}]
        fitargs = {'fixedpars': dict(fixed)}
        if spec:
            fitargs.update(obs_wave_fit=obs, obs_flux_fit=flux, obs_eflux_fit=eflux, specANNpath=path, NNtype='YST1')
        if has_phot:
            fitargs.update(photANNpath=phot, obs_phot=obs_phot)
        L = likelihood(fitargs, fitpars, [spec, has_phot, modpoly, photscale, False], b_max=64, verbose=False)
        OL = O.OracleLikelihood(net if spec else None, obs if spec else None, flux, eflux, L.fitpars_i, fixedpars=fixed,
                                modpoly=modpoly, phot=dict(phot, hiav=None) if has_phot else None,
                                obs_phot=obs_phot if has_phot else None, photscale=photscale, spec=spec)
        theta = np.array([[rng.uniform(0.95, 1.05) if p == 'pc_0' else rng.normal(0, 0.02) if p.startswith('pc_')
                           else rng.uniform(*R[p]) for p in L.fitpars_i] for _ in range(6)])
        try:
            ref = np.array([OL.lnlikefn(list(t)) for t in theta])
        except KeyError:
            # a fixed log(A) / log(R) / Dist: the reference decides the parametrisation by `in fitpars_i`
            # (likelihood.py:57-64) and raises KeyError; nothing to compare (the build evaluates such a fit)
            assert any(k in fixed for k in ('log(A)', 'log(R)', 'Dist'))
            L.GM.engine.close()
            continue
        got = L.lnlike_batch(theta)
        one = L.lnlikefn(list(theta[0]))
        assert np.array_equal(np.isnan(got), np.isnan(ref)), (it, L.fitpars_i, fixed, got, ref)
        ok = ~np.isnan(ref)
        assert np.all(np.abs(got[ok] - ref[ok]) <= lnl_tol(ref[ok])), (it, L.fitpars_i, fixed, got, ref)
        assert (np.isnan(one) and np.isnan(ref[0])) or abs(one - ref[0]) <= lnl_tol(ref[0])
        assert L.parsdict == {**dict(zip(L.fitpars_i, theta[0])), **fixed}
        ncmp += len(ref)
        L.GM.engine.close()
    assert ncmp >= 18 * 6


def test_random_sed_calls():
    """FastPayneSEDPredict.sed / GenMod.genphot(_scaled) on random arguments (both sides of av = 5, both
    magnitude formulae) against the oracle."""
    from thepayne_amd.predict.predictsed import FastPayneSEDPredict
    from thepayne_amd.fitting.genmod import GenMod
    rng = np.random.default_rng(55 + 1000 * SEED0)
    phot = synth.make_phot_nets()
    S = FastPayneSEDPredict(usebands=phot["filters"], nnpath=phot)
    oph = dict(phot)
    oph["hiav"] = np.array(S.HiAv.Avlist, dtype=float)
    GM = GenMod()
    GM._initphotnn(phot["filters"], nnpath=phot)
    for it in range(80):
        logt, logg = np.log10(rng.uniform(3500, 9000)), rng.uniform(0, 5)
        feh, afe = rng.uniform(-2, 0.5), rng.uniform(-0.2, 0.6)
        av = [rng.uniform(0, 4.99), 5.0, rng.uniform(5.0, 9.0)][rng.integers(3)]
        kw = dict(logt=logt, logg=logg, feh=feh, afe=afe, av=av)
        if rng.uniform() < 0.6:
            kw['rv'] = rng.uniform(2.2, 4.8)
        if rng.uniform() < 0.5:
            kw.update(logl=rng.uniform(-1, 2), dist=rng.uniform(5, 5000))
        else:
            kw['logA'] = rng.uniform(-2, 3)
        ref = O.sed_mags(oph, logt, logg, feh, afe, **{k: v for k, v in kw.items() if k in ('av', 'rv', 'logl', 'dist', 'logA')})
        got = S.sed(**kw)
        assert np.abs(np.asarray(got) - ref).max() < 1e-9, (it, kw)
        if av < 5.0:
            p = [10.0 ** logt, logg, feh, afe, rng.uniform(-0.5, 1.0), rng.uniform(10, 3000), av]
            ref = O.genphot(oph, p)
            got = GM.genphot(p)
            assert np.abs(np.array([got[f] for f in phot["filters"]]) - ref).max() < 1e-9
            p = [10.0 ** logt, logg, feh, afe, rng.uniform(-2, 3), av]
            ref = O.genphot_scaled(oph, p)
            got = GM.genphot_scaled(p)
            assert np.abs(np.array([got[f] for f in phot["filters"]]) - ref).max() < 1e-9


def test_random_prior_dictionaries_on_the_device(tmp_path):
    """Random priordicts (a random pv_* kind with random parameters per sampled dimension, random additional
    'gaussian' / 'uniform' terms): the device transform and ln-prior against the host prior object, which is
    pinned to the reference by tests/golden/g6_prior.npz."""
    from thepayne_amd.fitting.fitstar import lnprob_batch
    from thepayne_amd.sampler.device import DeviceProposer
    from test_api_gpu import _fit_objects
    from test_sampler_gpu import _clone_prior
    rng = np.random.default_rng(808 + 1000 * SEED0)
    centre = {'Teff': 5800.0, 'log(g)': 4.3, '[Fe/H]': -0.1, '[a/Fe]': 0.1, 'Vrad': 10.0, 'Vrot': 4.0, 'Inst_R': 29000.0,
              'log(A)': 0.2, 'log(R)': 0.1, 'Dist': 300.0, 'Av': 0.6}
    width = {'Teff': 400.0, 'log(g)': 0.3, '[Fe/H]': 0.2, '[a/Fe]': 0.1, 'Vrad': 3.0, 'Vrot': 2.0, 'Inst_R': 2000.0,
             'log(A)': 0.3, 'log(R)': 0.2, 'Dist': 100.0, 'Av': 0.3}
    for photscale in (True, False):
        L, P0, _ = _fit_objects(tmp_path, photscale=photscale)
        for trial in range(6):
            pd = {}
            for par in L.fitpars_i:
                c, w = centre[par], width[par]
                kinds = ['uniform', 'gaussian', 'tgaussian', 'default']
                if par in ('Vrot', 'Av', 'Dist'):
                    kinds += ['exp', 'texp']
                kind = kinds[rng.integers(len(kinds))]
                spec = {}
                if kind == 'uniform':
                    spec['pv_uniform'] = [c - 2 * w, c + 2 * w]
                elif kind == 'gaussian':
                    spec['pv_gaussian'] = [c + rng.normal(0, 0.2) * w, w * rng.uniform(0.3, 1.5)]
                elif kind == 'tgaussian':
                    spec['pv_tgaussian'] = [c - rng.uniform(0.5, 3) * w, c + rng.uniform(0.5, 3) * w, c, w * rng.uniform(0.3, 2)]
                elif kind == 'exp':
                    spec['pv_exp'] = [max(0.0, c - 2 * w), w]
                elif kind == 'texp':
                    spec['pv_texp'] = [max(0.0, c - 2 * w), c + 3 * w, w]
                if rng.uniform() < 0.3:
                    spec['gaussian'] = [c, w * rng.uniform(0.5, 2)]
                if rng.uniform() < 0.2:
                    spec['uniform'] = [c - 1.5 * w, c + 1.5 * w]
                if spec:
                    pd[par] = spec
            P = _clone_prior(P0, pd)
            prop = DeviceProposer(L, P, k_max=64)
            U = rng.uniform(size=(64, L.ndim))
            U[:4] = np.clip(U[:4], 1e-9, 1 - 1e-9)
            U[0, :] = 1e-7
            U[1, :] = 1 - 1e-7
            with np.errstate(all="ignore"):
                host_v = P.priortrans_batch(U)
                V = prop.prior_transform(U)
            np.testing.assert_allclose(V, host_v, rtol=2e-9, atol=1e-9, err_msg=str(pd))
            V2, lp = prop.lnprob_u(U)
            np.testing.assert_allclose(V2, host_v, rtol=2e-9, atol=1e-9)
            with np.errstate(all="ignore"):
                ref = lnprob_batch(V2, L, P)
            assert np.array_equal(np.isneginf(lp), np.isneginf(ref)) and np.array_equal(np.isnan(lp), np.isnan(ref))
            ok = np.isfinite(ref)
            assert np.all(np.abs(lp[ok] - ref[ok]) <= 1e-9 * np.abs(ref[ok]) + 1e-6), (pd, np.abs(lp[ok] - ref[ok]).max())
            prop.close()
        L.GM.engine.close()
