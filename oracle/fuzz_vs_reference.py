"""Differential check of the restatement against the reference itself, on random API calls.

Runs only where /root/reference is mounted (it imports the unmodified reference through
oracle/ref_shim.py); nothing on the GPU box or in tests/ uses it.  For N random draws of the
getspec / genspec / lnlikefn arguments -- label aliases and defaults, outwave given or not,
inst_R absent / scalar / NaN / 0 / negative / above the network's / a dispersion vector, rotation
and Doppler on or off, blaze polynomial -- the restatement must return what the reference returns
(same NaN pattern, values to 1e-12).

    python oracle/fuzz_vs_reference.py [N=300] [seed=0]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ref_shim as rs  # noqa: E402

rs.install()
sys.path.insert(0, "/root/reference")

from Payne.predict import ystpred                       # noqa: E402
from Payne.fitting.genmod import GenMod                 # noqa: E402

import oracle as O                                       # noqa: E402
from thepayne_amd import synth                          # noqa: E402


def same(a, b, what, tol=1e-12):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    if a.shape != b.shape:
        return "%s: shape %s vs %s" % (what, a.shape, b.shape)
    if not np.array_equal(np.isnan(a), np.isnan(b)):
        return "%s: NaN pattern differs (%d vs %d)" % (what, np.isnan(a).sum(), np.isnan(b).sum())
    ok = ~np.isnan(a)
    if ok.any() and np.abs(a[ok] - b[ok]).max() > tol:
        return "%s: max diff %.3g" % (what, np.abs(a[ok] - b[ok]).max())
    return None


ALL_PARS = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Vmic', 'Inst_R', 'log(R)', 'Dist', 'log(A)', 'Av', 'Rv',
            'CarbonScale']
RANGES = {'Teff': (4500, 7000), 'log(g)': (1.0, 5.0), '[Fe/H]': (-1.5, 0.4), '[a/Fe]': (-0.1, 0.5), 'Vrad': (-50, 50),
          'Vrot': (0, 30), 'Inst_R': (20000, 40000), 'log(R)': (-0.5, 1.0), 'Dist': (10, 3000), 'log(A)': (-2, 3),
          'Av': (0, 4.5), 'Rv': (2.2, 4.8)}


def fuzz_likelihood(n, rng):
    """likelihood.lnlikefn of the reference against OracleLikelihood.lnlikefn for random fit set-ups: spectrum
    and/or photometry (both parametrisations, Rv free or not), blaze polynomial, fixed parameters."""
    from Payne.fitting.likelihood import likelihood
    net = synth.make_yst_net(npix=512, H=32, seed=24, line_depth=0.3)
    rs.register_yst('/fuzz/like.h5', net)
    phot = synth.make_phot_nets()
    rs.register_phot('/fuzz/phot/', phot)
    obs = synth.obs_grid(net["wavelength"], 300, inset=1.0)
    _, clean = O.getspec(net, Teff=5770.0, logg=4.44, feh=0.0, afe=0.0, rad_vel=10.0, rot_vel=3.0, inst_R=2.355 * 28800.0, outwave=obs)
    flux = clean + rng.normal(0, 0.01, len(obs))
    eflux = np.full(len(obs), 0.01)
    obs_phot = {f: [5.0 + 0.1 * i, 0.05] for i, f in enumerate(phot["filters"])}
    bad = []
    fuzz_likelihood.compared = fuzz_likelihood.both_raised = 0
    for it in range(n):
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
        npoly = 3 if modpoly else 0
        names = list(ALL_PARS) + ['pc_%d' % i for i in range(npoly)]
        fixed = {}
        for name in rng.permutation(on)[:int(rng.integers(0, 3))]:
            if name in ('Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Av') and has_phot and name == 'Av':
                continue
            fixed[str(name)] = float(rng.uniform(*RANGES[str(name)]))
        fitpars = [names, {p: ((p in on and p not in fixed) or p.startswith('pc_')) for p in names}]
        fitargs = {'fixedpars': dict(fixed)}
        if spec:
            fitargs.update(obs_wave_fit=obs, obs_flux_fit=flux, obs_eflux_fit=eflux, specANNpath='/fuzz/like.h5', NNtype='YST1')
        if has_phot:
            fitargs.update(photANNpath='/fuzz/phot/', obs_phot=obs_phot)
        runbools = [spec, has_phot, modpoly, photscale, False]
        try:
            L = likelihood(fitargs, fitpars, runbools)
        except Exception as e:
            bad.append((it, "reference constructor raised %r" % (e,), dict(on=on, fixed=fixed)))
            continue
        OL = O.OracleLikelihood(net if spec else None, obs if spec else None, flux, eflux, L.fitpars_i, fixedpars=fixed,
                                modpoly=modpoly, phot=dict(phot, hiav=None) if has_phot else None,
                                obs_phot=obs_phot if has_phot else None, photscale=photscale, spec=spec)
        for _ in range(3):
            th = [float(rng.uniform(0.95, 1.05)) if p == 'pc_0' else float(rng.normal(0, 0.02)) if p.startswith('pc_')
                  else float(rng.uniform(*RANGES[p])) for p in L.fitpars_i]
            with np.errstate(all="ignore"):
                try:
                    ref = L.lnlikefn(list(th))
                except Exception as e:
                    try:
                        OL.lnlikefn(list(th))
                        bad.append((it, "reference raised %s, restatement did not" % type(e).__name__, dict(on=L.fitpars_i, fixed=fixed)))
                    except Exception:
                        fuzz_likelihood.both_raised += 1
                    continue
                got = OL.lnlikefn(list(th))
            fuzz_likelihood.compared += 1
            if not ((np.isnan(ref) and np.isnan(got)) or abs(ref - got) <= 1e-9 * max(1.0, abs(ref))):
                bad.append((it, "lnlike %r vs %r" % (got, ref), dict(on=L.fitpars_i, fixed=fixed, modpoly=modpoly, photscale=photscale)))
            if L.parsdict != OL.parsdict:
                bad.append((it, "parsdict differs", dict(ref=L.parsdict, got=OL.parsdict)))
    return bad


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad_l = fuzz_likelihood(max(20, n // 5), rng)
    print("%d likelihood set-ups (%d values compared, %d calls where both sides raise), %d disagreements"
          % (max(20, n // 5), fuzz_likelihood.compared, fuzz_likelihood.both_raised, len(bad_l)))
    for b in bad_l[:12]:
        print(b[0], b[1], b[2])
    nets = {}
    for D in (4, 5):
        net = synth.make_yst_net(npix=512, H=32, seed=20 + D, D=D, line_depth=0.3)
        rs.register_yst('/fuzz/yst%d.h5' % D, net)
        nets[D] = (net, ystpred.PayneSpecPredict(nnpath='/fuzz/yst%d.h5' % D, NNtype='YST1'))
    GM = GenMod()
    GM._initspecnn(nnpath='/fuzz/yst4.h5', NNtype='YST1')
    bad = []
    for it in range(n):
        D = 4 if rng.uniform() < 0.7 else 5
        net, PP = nets[D]
        wave = net["wavelength"]
        kw = {}
        lab = dict(Teff=rng.uniform(4000, 7500), logg=rng.uniform(0.5, 5.2), feh=rng.uniform(-2, 0.4), afe=rng.uniform(-0.1, 0.5))
        alias = {"Teff": ["Teff", "logt"], "logg": ["logg", "log(g)"], "feh": ["feh", "[Fe/H]"],
                 "afe": ["afe", "aFe", "[a/Fe]", "[alpha/Fe]"]}
        for k, v in lab.items():
            if rng.uniform() < 0.15:
                continue                                            # default label
            name = alias[k][rng.integers(len(alias[k]))]
            kw[name] = np.log10(v) if name == "logt" else v
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
        if mode == 1:
            kw['inst_R'] = float(rng.uniform(8000, 60000)) * 2.355 / 2.355
        elif mode == 2:
            kw['inst_R'] = [np.nan, 0.0, -5.0, float(net["resolution"]) * 1.2][rng.integers(4)]
        elif mode == 3:
            kw['inst_R'] = float(rng.uniform(20000, 40000))
        elif mode == 4:
            m = nobs if outwave is not None else len(wave)
            x = np.linspace(-0.5, 0.5, m)
            kw['inst_R'] = 0.08 * (1.0 + rng.uniform(-0.5, 0.5) * x + rng.uniform(0, 0.5) * x ** 2)
        try:
            with np.errstate(all="ignore"):
                w_ref, f_ref = PP.getspec(**kw)
        except Exception as e:                                       # the reference raises: so must the restatement
            try:
                with np.errstate(all="ignore"):
                    O.getspec(net, **{k: v for k, v in kw.items() if k in ('vmic', 'rot_vel', 'rad_vel', 'inst_R', 'outwave')})
                bad.append((it, "reference raised %s, restatement did not" % type(e).__name__, kw))
            except Exception:
                pass
            continue
        # label aliases / defaults are host logic of the build: its resolver against the reference's inputdict
        from thepayne_amd.predict._spec import PayneSpecPredict as Host
        t_, g_, f_, a_ = Host._labels_from_kwargs(kw)
        got = (t_, g_, f_, a_)
        want = tuple(PP.inputdict[k] for k in ('teff', 'logg', 'feh', 'afe'))
        if got != want:
            bad.append((it, "labels %r vs %r" % (got, want), kw))
        canon = {k: v for k, v in kw.items() if k in ('vmic', 'rot_vel', 'rad_vel', 'inst_R', 'outwave')}
        canon.update(Teff=t_, logg=g_, feh=f_, afe=a_)
        try:
            with np.errstate(all="ignore"):
                w_o, f_o = O.getspec(net, **canon)
        except Exception as e:
            bad.append((it, "restatement raised %r" % (e,), kw))
            continue
        msg = same(w_o, w_ref, "wave", 0.0) or same(f_o, f_ref, "flux")
        if msg:
            bad.append((it, "getspec " + msg, kw))
        # genspec through GenMod on the 4-label net: list interface, FWHM-based Inst_R, blaze
        if D == 4 and not isinstance(kw.get('inst_R', 1.0), np.ndarray):
            R = kw.get('inst_R', float(rng.uniform(20000, 40000)))
            pars = [lab["Teff"], lab["logg"], lab["feh"], lab["afe"], kw.get('rad_vel', 0.0), kw.get('rot_vel', 0.0), np.nan, float(R)]
            poly = rng.uniform() < 0.5
            if poly:
                pars = pars + [1.0 + 0.05 * rng.standard_normal(), 0.05 * rng.standard_normal(), 0.02 * rng.standard_normal()]
            try:
                with np.errstate(all="ignore"):
                    wr, fr = GM.genspec(pars, outwave=outwave, modpoly=poly)
            except Exception:
                continue
            with np.errstate(all="ignore"):
                wo, fo = O.genspec(net, pars, outwave=outwave, modpoly=poly)
            msg = same(wo, wr, "wave", 0.0) or same(fo, fr, "flux")
            if msg:
                bad.append((it, "genspec " + msg, dict(pars=pars, outwave=None if outwave is None else len(outwave), poly=poly)))
    print("%d random calls, %d disagreements" % (n, len(bad)))
    for b in bad[:20]:
        print(b[0], b[1], {k: (v if np.ndim(v) == 0 else "array[%d]" % len(v)) for k, v in b[2].items()})
    return 1 if (bad or bad_l) else 0


if __name__ == "__main__":
    sys.exit(main())
