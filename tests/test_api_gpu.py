"""The reference-shaped Python API (FitPayne / likelihood / GenMod / PayneSpecPredict /
FastPayneSEDPredict) driving the HIP engine, against the oracle.  Reads like a test the
reference would have for these classes."""
import numpy as np
import pytest

import oracle as O
from thepayne_amd import synth, nnio
from helpers import SPEC_PARS, yst_problem, lnl_tol, theta_full

pytestmark = pytest.mark.gpu
ALL_PARS = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Vmic', 'Inst_R',
            'log(R)', 'Dist', 'log(A)', 'Av', 'Rv', 'CarbonScale']


def _save_yst(tmp_path, raw, name="yst.npz"):
    path = str(tmp_path / name)
    nnio.save_npz(path, {k: (np.array([v]) if k == "resolution" else v) for k, v in raw.items() if k != "kind"})
    return path


def test_payne_spec_predict_yst(tmp_path):
    from thepayne_amd.predict.ystpred import PayneSpecPredict
    raw = synth.make_yst_net(npix=1024, H=64, seed=5, line_depth=0.3)
    PP = PayneSpecPredict(nnpath=_save_yst(tmp_path, raw), NNtype='YST1')
    assert np.allclose(PP.anns.wavelength, raw["wavelength"]) and PP.anns.resolution == raw["resolution"]
    lab = [5300.0, 4.1, -0.3, 0.15]
    assert np.abs(PP.predictspec(lab) - O.yst_forward(raw, lab)).max() < 1e-6
    obs = synth.obs_grid(raw["wavelength"], 700, inset=1.5)
    kw = dict(Teff=5300.0, logg=4.1, feh=-0.3, afe=0.15)
    for extra in (dict(rad_vel=12.0, rot_vel=4.0, inst_R=2.355 * 28000.0, outwave=obs),
                  dict(rad_vel=-20.0, rot_vel=0.0, inst_R=2.355 * 20000.0, outwave=obs),
                  dict(rad_vel=5.0, rot_vel=2.0, outwave=obs),                      # no inst_R: plain interpolation
                  dict(rot_vel=3.0),                                               # model grid, rotation only
                  dict(rad_vel=30.0, rot_vel=3.0)):                                # shifted model grid
        w, f = PP.getspec(**kw, **extra)
        wr, fr = O.getspec(raw, **kw, **extra)
        assert np.allclose(w, wr, rtol=0, atol=1e-9)
        assert np.array_equal(np.isnan(f), np.isnan(fr)) and np.nanmax(np.abs(f - fr)) < 1e-6
    # aliases and defaults (ystpred.py:132-165): solar defaults when labels are absent
    w1, f1 = PP.getspec(**{'logt': np.log10(5300.0), 'log(g)': 4.1, '[Fe/H]': -0.3, '[a/Fe]': 0.15}, rot_vel=3.0)
    w2, f2 = PP.getspec(**kw, rot_vel=3.0)
    assert np.abs(f1 - f2).max() < 2e-6
    _, fs = PP.getspec()
    assert np.abs(fs - O.yst_forward(raw, [5770.0, 4.44, 0.0, 0.0])).max() < 1e-6
    with pytest.raises(ValueError):
        PP.getspec(**kw, inst_R=np.full(1024, 0.1), outwave=obs)                   # LSF vector of the wrong length (np.interp raises)


def test_payne_spec_predict_linnet(tmp_path):
    from thepayne_amd.predict.predictspec import PayneSpecPredict
    raw = synth.make_torch_net("LinNet", npix=512, seed=7)
    path = str(tmp_path / "lin.npz")
    d = {("model/" + k if k.startswith("lin") else k): v for k, v in raw.items() if k != "kind"}
    d["wavelengths"] = d.pop("wavelength")
    nnio.save_npz(path, d)
    PP = PayneSpecPredict(nnpath=path, NNtype='LinNet')
    lab = [6100.0, 3.9, -0.8, 0.2]
    ref = O.torchnet_forward(raw, lab)
    got = PP.predictspec(lab)
    assert got.dtype == np.float32 and np.abs(got - ref).max() < 3e-6


def test_sed_predictors():
    from thepayne_amd.predict.predictsed import FastPayneSEDPredict, PayneSEDPredict
    phot = synth.make_phot_nets()
    S = FastPayneSEDPredict(usebands=phot["filters"], nnpath=phot)
    oph = dict(phot); oph["hiav"] = np.array(S.HiAv.Avlist, dtype=float)
    for av in (0.3, 4.999, 5.0, 7.5):
        m = S.sed(logt=np.log10(5800.0), logg=4.3, feh=-0.2, afe=0.1, av=av, rv=3.3, logl=0.1, dist=150.0)
        assert np.abs(m - O.sed_mags(oph, np.log10(5800.0), 4.3, -0.2, 0.1, av=av, rv=3.3, logl=0.1, dist=150.0)).max() < 1e-9
        m = S.sed(logt=np.log10(4800.0), logg=2.3, feh=-1.2, afe=0.3, av=av, logA=1.7)
        assert np.abs(m - O.sed_mags(oph, np.log10(4800.0), 2.3, -1.2, 0.3, av=av, logA=1.7)).max() < 1e-9
    assert len(S.sed(logt=3.76, logg=4.4, feh=0, afe=0, logA=0.0, band_indices=slice(0, 3))) == 3
    with pytest.raises(IOError):
        S.sed(logt=3.76, logg=4.4, feh=0, afe=0)
    x = [5800.0, 4.3, -0.2, 0.1, 0.4, 3.1]
    assert np.abs(S.anns.eval(x) - O.fastann_forward(phot, x)).max() < 1e-10
    P = PayneSEDPredict(usebands=phot["filters"], nnpath=phot)
    assert np.allclose(P.sed(logt=3.7, logg=4.0, feh=0, afe=0, av=0.2, rv=3.1, logA=1.0),
                       S.sed(logt=3.7, logg=4.0, feh=0, afe=0, av=0.2, rv=3.1, logA=1.0))


def _fit_objects(tmp_path, photscale=True, modpoly=False, variant=0, b_max=64):
    from thepayne_amd.fitting.likelihood import likelihood
    from thepayne_amd.fitting.prior import prior
    raw, obs, flux, eflux = yst_problem("small", H=64)
    phot = synth.make_phot_nets()
    obs_phot = {f: [5.0 + 0.1 * i, 0.05] for i, f in enumerate(phot["filters"])}
    on = SPEC_PARS + (['log(A)', 'Av'] if photscale else ['log(R)', 'Dist', 'Av'])
    names = list(ALL_PARS) + (['pc_0', 'pc_1', 'pc_2'] if modpoly else [])
    fitpars = [names, {p: (p in on or p.startswith('pc_')) for p in names}]
    fitargs = {'obs_wave_fit': obs, 'obs_flux_fit': flux, 'obs_eflux_fit': eflux,
               'specANNpath': _save_yst(tmp_path, raw), 'NNtype': 'YST1', 'fixedpars': {},
               'photANNpath': phot, 'obs_phot': obs_phot}
    runbools = [True, True, modpoly, photscale, False]
    pd = synth.demo_priordict()
    pd['log(A)'] = {'pv_uniform': [-1.0, 1.0]}
    pd['Av'] = {'pv_uniform': [0.0, 2.0]}
    pd['log(R)'] = {'pv_uniform': [-0.5, 0.5]}
    pd['Dist'] = {'pv_uniform': [10.0, 1000.0]}
    if modpoly:
        pd['blaze_coeff'] = [[0.0, 0.05], [0.0, 0.02], [0.0, 0.01]]
    L = likelihood(fitargs, fitpars, runbools, b_max=b_max, variant=variant)
    P = prior(fitargs, pd, fitpars, runbools)
    OL = O.OracleLikelihood(raw, obs, flux, eflux, L.fitpars_i, phot=dict(phot, hiav=None), obs_phot=obs_phot,
                            photscale=photscale, modpoly=modpoly)
    from thepayne_amd.engine import highav_coefficients
    OL.phot["hiav"] = highav_coefficients(phot["filters"])
    return L, P, OL


@pytest.mark.parametrize("photscale,modpoly", [(True, False), (False, False), (True, True)])
def test_likelihood_object_scalar_and_batch(tmp_path, photscale, modpoly):
    from thepayne_amd.fitting.fitstar import lnprobfn, lnprob_batch
    L, P, OL = _fit_objects(tmp_path, photscale, modpoly)
    assert L.ndim == len(L.fitpars_i) == P.ndim
    U = np.random.default_rng(4).uniform(size=(24, L.ndim))
    theta = P.priortrans_batch(U)
    ref = np.array([OL.lnlikefn(t) for t in theta])
    got = L.lnlike_batch(theta)
    assert np.all(np.abs(got - ref) <= lnl_tol(ref))
    one = L.lnlikefn(theta[3])
    assert isinstance(one, float) and abs(one - ref[3]) <= lnl_tol(ref[3])
    assert list(L.parsdict.keys())[:L.ndim] == L.fitpars_i and L.parsdict['Teff'] == theta[3, 0]
    assert abs(lnprobfn(theta[5], L, P) - ref[5]) <= lnl_tol(ref[5])
    assert np.all(np.abs(lnprob_batch(theta, L, P) - ref) <= lnl_tol(ref))
    # explicit specpars / photpars lists (likelihood.lnlike signature)
    sp, pp = OL.split([float(x) for x in theta[7]])
    assert abs(L.lnlike(specpars=sp, photpars=pp) - ref[7]) <= lnl_tol(ref[7])
    # GenMod pieces
    w, f = L.GM.genspec(sp, outwave=L.fitargs['obs_wave_fit'], modpoly=modpoly)
    _, fr = O.genspec(OL.net, sp, outwave=OL.obs_wave, modpoly=modpoly)
    assert np.nanmax(np.abs(f - fr)) < 2e-6
    mags = L.GM.genphot_scaled(pp) if photscale else L.GM.genphot(pp)
    mref = np.atleast_1d(O.genphot_scaled(OL.phot, pp) if photscale else O.genphot(OL.phot, pp))
    assert list(mags.keys()) == list(L.fitargs['obs_phot'].keys())
    assert np.abs(np.array(list(mags.values())) - mref).max() < 1e-9


def test_fixed_parameters_are_merged(tmp_path):
    from thepayne_amd.fitting.likelihood import likelihood
    raw, obs, flux, eflux = yst_problem("small", H=64)
    names = list(ALL_PARS)
    on = [p for p in SPEC_PARS if p not in ('[a/Fe]', 'Vrot')]
    fitpars = [names, {p: p in on for p in names}]
    fitargs = {'obs_wave_fit': obs, 'obs_flux_fit': flux, 'obs_eflux_fit': eflux,
               'specANNpath': _save_yst(tmp_path, raw), 'NNtype': 'YST1', 'fixedpars': {'[a/Fe]': 0.03, 'Vrot': 2.5}}
    L = likelihood(fitargs, fitpars, [True, False, False, False, False], b_max=8)
    OL = O.OracleLikelihood(raw, obs, flux, eflux, on, fixedpars=fitargs['fixedpars'])
    th = np.array([5600.0, 4.3, 0.02, 10.2, 28500.0])
    assert abs(L.lnlikefn(th) - OL.lnlikefn(th)) <= lnl_tol(OL.lnlikefn(th))
    assert L.parsdict['Vrot'] == 2.5


def test_fitpayne_run_end_to_end(tmp_path):
    """C1-style plumbing: the reference's inputdict through FitPayne.run with the batched
    sampler; the posterior must bracket the truth and the output file keep its format."""
    from thepayne_amd.fitting.fitstar import FitPayne
    raw, obs, flux, eflux = yst_problem("small", H=64, line_depth=0.3)
    inputdict = {
        'spec': {'obs_wave': obs, 'obs_flux': flux, 'obs_eflux': eflux, 'convertair': False},
        'specANNpath': _save_yst(tmp_path, raw), 'NNtype': 'YST1',
        'sampler': {'samplertype': 'Static', 'samplerbounds': 'multi', 'samplemethod': 'rwalk', 'npoints': 128,
                    'walks': 20, 'delta_logz_final': 0.5, 'bootstrap': 0, 'flushnum': 200, 'seed': 3},
        'priordict': synth.demo_priordict(),
        'output': str(tmp_path / 'fit.dat'),
    }
    sampler = FitPayne().run(inputdict=inputdict, verbose=False)
    r = sampler.results
    w = sampler.posterior_weights()
    mean = (w[:, None] * r.samples).sum(0)
    std = np.sqrt((w[:, None] * (r.samples - mean) ** 2).sum(0))
    T = synth.TRUTH
    truth = np.array([T["Teff"], T["logg"], T["feh"], T["afe"], T["vrad"], T["vrot"], T["inst_R"]])
    assert np.all(np.abs(mean - truth) < 5 * std + 1e-3 * np.abs(truth)), (mean, std, truth)
    lines = open(inputdict['output']).read().splitlines()
    assert lines[0].split()[:2] == ['Iter', 'Teff'] and lines[0].split()[-7:] == ['log(lk)', 'log(vol)', 'log(wt)', 'h', 'nc', 'log(z)', 'delta(log(z))']
    assert len(lines) == 1 + r.niter and len(lines[1].split()) == 1 + 7 + 7


def test_continuum_network(tmp_path, golden):
    """PayneSpecPredict(Cnnpath=...): predictcont and getspec with the continuum product
    (ystpred.py:81-85, 101-117, 191-209) against vectors frozen from the reference, two continuum grids."""
    from thepayne_amd.predict.ystpred import PayneSpecPredict
    g = golden("g8_continuum")
    raw = synth.make_yst_net(npix=1024, H=64, seed=5, line_depth=0.3)
    spath = _save_yst(tmp_path, raw)
    for tag, (lo, hi, npc) in {"full": (5140.0, 5190.0, 600), "short": (5140.0, 5170.0, 333)}.items():
        cnet = synth.make_cont_net(npix=npc, lam_lo=lo, lam_hi=hi)
        PP = PayneSpecPredict(nnpath=spath, Cnnpath=_save_yst(tmp_path, cnet, "cont_%s.npz" % tag), NNtype='YST1')
        assert np.allclose(PP.Canns.wavelength, cnet["wavelength"])
        for i, l in enumerate(g["labels"]):
            pc = PP.predictcont(list(l))
            assert np.abs(pc / g["cont_" + tag][i] - 1.0).max() < 2e-6            # fp32 network, values ~4e-5
            assert np.abs(PP.predictspec(list(l)) - O.yst_forward(raw, list(l))).max() < 1e-6   # no continuum in predictspec
            kw = dict(Teff=l[0], logg=l[1], feh=l[2], afe=l[3])
            w, nat = PP.getspec(**kw)
            ref = g["native_" + tag][i]
            assert np.array_equal(np.isnan(nat), np.isnan(ref)) and np.isnan(ref).any() == (tag == "short")
            assert np.nanmax(np.abs(nat - ref)) < 2e-6
            for j, (vrad, vrot, R) in enumerate(g["rows"]):
                _, f = PP.getspec(rad_vel=vrad, rot_vel=vrot, inst_R=2.355 * R, outwave=g["obs"], **kw)
                ref = g["final_" + tag][i, j]
                assert np.array_equal(np.isnan(f), np.isnan(ref)), (tag, i, j)
                assert np.nanmax(np.abs(f - ref)) < 2e-6, (tag, i, j)


def test_genmod_with_continuum_network(tmp_path):
    """GenMod._initspecnn(Cnnpath=...) (genmod.py:28-32): genspec and the batched chi^2 carry the continuum."""
    from thepayne_amd.fitting.genmod import GenMod
    raw, obs, flux, eflux = yst_problem("small", H=64)
    cnet = synth.make_cont_net(npix=500, lam_lo=raw["wavelength"][0] - 2.0, lam_hi=raw["wavelength"][-1] + 2.0, H=24)
    GM = GenMod(b_max=16)
    GM._initspecnn(nnpath=_save_yst(tmp_path, raw), NNtype='YST1', Cnnpath=_save_yst(tmp_path, cnet, "cont.npz"))
    GM.configure(obs=(obs, flux, eflux))
    rng = np.random.default_rng(3)
    th7 = synth.draw_candidates(8, seed=9)
    for t in th7[:3]:
        pars = [float(x) for x in t[:6]] + [np.nan, float(t[6])]
        w, f = GM.genspec(pars, outwave=obs)
        _, fr = O.genspec(raw, pars + [np.nan], outwave=obs, cnet=cnet)
        assert np.array_equal(np.isnan(f), np.isnan(fr)) and np.nanmax(np.abs(f - fr)) < 2e-6
    from helpers import theta_full
    got = GM.engine.lnlike_batch(theta_full(th7)).cpu().numpy()
    ref = np.array([-0.5 * O.chi2_spec(O.genspec(raw, [float(x) for x in t[:6]] + [np.nan, float(t[6]), np.nan], outwave=obs,
                                                 cnet=cnet)[1], flux, eflux) for t in th7])
    ok = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), ok) and np.all(np.abs(got[ok] - ref[ok]) <= lnl_tol(ref[ok]))


def test_lsf_vector_inst_R(tmp_path, golden):
    """getspec(inst_R=<dispersion per output pixel>) (ystpred.py:248-269 -> smooth_lsf_fft) against vectors
    frozen from the reference: three LSF shapes x Doppler / rotation, plus the model-grid (outwave=None) form."""
    from thepayne_amd.predict.ystpred import PayneSpecPredict
    g = golden("g9_lsf")
    raw = synth.make_yst_net(npix=1024, H=64, seed=5, line_depth=0.3)
    PP = PayneSpecPredict(nnpath=_save_yst(tmp_path, raw), NNtype='YST1')
    worst = 0.0
    for a, l in enumerate(g["labels"]):
        kw = dict(Teff=l[0], logg=l[1], feh=l[2], afe=l[3])
        for b, lsf in enumerate(g["lsfs"]):
            for c, (vrad, vrot) in enumerate(g["rows"]):
                w, f = PP.getspec(rad_vel=vrad, rot_vel=vrot, inst_R=lsf, outwave=g["obs"], **kw)
                assert np.array_equal(w, g["obs"]) and not np.isnan(f).any()
                worst = max(worst, np.abs(f - g["final"][a, b, c]).max())
    assert worst < 2e-6, worst
    w, f = PP.getspec(rad_vel=8.0, rot_vel=3.0, inst_R=g["lsf_native"], Teff=5300.0, logg=4.1, feh=-0.3, afe=0.15)
    assert np.allclose(w, g["native_wave"], rtol=1e-14) and np.abs(f - g["native"]).max() < 2e-6
    # a scalar call afterwards is unaffected (the vector is bound for the call only)
    _, fs = PP.getspec(rad_vel=12.0, rot_vel=4.0, inst_R=2.355 * 28000.0, outwave=g["obs"], Teff=5300.0, logg=4.1, feh=-0.3, afe=0.15)
    _, fr = O.getspec(raw, Teff=5300.0, logg=4.1, feh=-0.3, afe=0.15, rad_vel=12.0, rot_vel=4.0, inst_R=2.355 * 28000.0, outwave=g["obs"])
    assert np.nanmax(np.abs(fs - fr)) < 1e-6
    with pytest.raises(RuntimeError):
        PP.anns.engine.set_lsf(np.full(len(g["obs"]) - 1, 0.1))


def test_fixed_lsf_vector_in_the_likelihood(tmp_path):
    """A fixed array-valued Inst_R (fixedpars) reaches getspec as an LSF vector (genmod.py:82-85)."""
    from thepayne_amd.fitting.likelihood import likelihood
    raw, obs, flux, eflux = yst_problem("small", H=64)
    lsf = 0.075 * (1.0 + 0.3 * (obs - obs.mean()) / (obs.max() - obs.min()))
    on = [p for p in SPEC_PARS if p != 'Inst_R']
    fitpars = [list(ALL_PARS), {p: p in on for p in ALL_PARS}]
    fitargs = {'obs_wave_fit': obs, 'obs_flux_fit': flux, 'obs_eflux_fit': eflux,
               'specANNpath': _save_yst(tmp_path, raw), 'NNtype': 'YST1', 'fixedpars': {'Inst_R': lsf}}
    L = likelihood(fitargs, fitpars, [True, False, False, False, False], b_max=8)
    th = synth.draw_candidates(6, seed=4)[:, :6]
    got = L.lnlike_batch(th)
    ref = np.array([-0.5 * O.chi2_spec(O.genspec(raw, [float(x) for x in t] + [np.nan, lsf, np.nan], outwave=obs)[1], flux, eflux)
                    for t in th])
    assert np.all(np.isfinite(got)) and np.all(np.abs(got - ref) <= lnl_tol(ref))
    w, f = L.GM.genspec([float(x) for x in th[0]] + [np.nan, lsf], outwave=obs)
    assert np.abs(f - O.genspec(raw, [float(x) for x in th[0]] + [np.nan, lsf, np.nan], outwave=obs)[1]).max() < 2e-6


def test_smoothspec_on_arbitrary_spectra(tmp_path):
    """PayneSpecPredict.smoothspec(wave, spec, sigma, ...) (ystpred.py:279-281 -> smoothing.smoothspec) on a
    caller-supplied, linearly sampled spectrum: the 'vsini', 'R' (with and without inres), 'vel' and 'lsf' branches."""
    from thepayne_amd.predict.ystpred import PayneSpecPredict
    raw = synth.make_yst_net(npix=256, H=16, seed=2)
    PP = PayneSpecPredict(nnpath=_save_yst(tmp_path, raw), NNtype='YST1')
    rng = np.random.default_rng(12)
    wave = np.linspace(6500.0, 6620.0, 3000)                       # linear grid: the non-geometric code path
    spec = np.ones_like(wave)
    for c, d, s in zip(rng.uniform(6505, 6615, 60), rng.uniform(0.05, 0.6, 60), rng.uniform(0.06, 0.15, 60)):
        spec -= d * np.exp(-0.5 * ((wave - c) / s) ** 2)
    out = wave[200:-200:2].copy()
    got = PP.smoothspec(wave, spec, 12.0, outwave=None, smoothtype='vsini', fftsmooth=True, inres=0.0)
    ref = O.smooth_vsini(wave, spec, 12.0)                                                 # (NaN where the log grid ends short)
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.nanmax(np.abs(got - ref)) < 1e-6
    got = PP.smoothspec(wave, spec, 20000.0, outwave=out, smoothtype='R', fftsmooth=True, inres=60000.0)
    assert np.abs(got - O.smooth_R(wave, spec, 20000.0, out, 60000.0)).max() < 1e-6
    got = PP.smoothspec(wave, spec, 20000.0, outwave=None, smoothtype='R', inres=60000.0)   # back onto `wave` itself
    ref = O.smooth_R(wave, spec, 20000.0, None, 60000.0)
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.nanmax(np.abs(got - ref)) < 1e-6
    got = PP.smoothspec(wave, spec, 25000.0, outwave=out, smoothtype='R')                  # no inres: nothing subtracted
    assert np.abs(got - O.smooth_R(wave, spec, 25000.0, out, np.inf)).max() < 1e-6
    got = PP.smoothspec(wave, spec, 14.0, outwave=out)                                     # default 'vel': sigma in km/s
    assert np.abs(got - O.smooth_R(wave, spec, 2.998e5 / 14.0, out, np.inf)).max() < 1e-6
    lsf = 0.25 * (1.0 + 0.4 * (wave - wave.mean()) / 120.0)
    got = PP.smoothspec(wave, spec, lsf, outwave=wave, smoothtype='lsf')
    assert np.abs(got - O.smooth_lsf(wave, spec, lsf, wave)).max() < 2e-6
    got = PP.smoothspec(wave, spec, 0.5, outwave=out, smoothtype='lambda')                 # smooth_wave_fft (see the g13 test)
    assert np.nanmax(np.abs(got - O.smoothspec_offpath(wave, spec, 0.5, outwave=out, smoothtype='lambda'))) < 1e-6
    with pytest.raises(NotImplementedError):
        PP.smoothspec(wave, spec, 0.5, outwave=out, smoothtype='boxcar')


def test_genspec_on_any_grid_like_the_reference(tmp_path):
    """GenMod.genspec(pars, outwave=...) for grids other than the fit's, outwave=None with and without an
    instrumental stage, and the blaze polynomial on each (genmod.py:58-108)."""
    from thepayne_amd.fitting.genmod import GenMod
    raw, obs, flux, eflux = yst_problem("small", H=64, line_depth=0.3)
    GM = GenMod()
    GM._initspecnn(nnpath=_save_yst(tmp_path, raw), NNtype='YST1')
    GM.configure(obs=(obs, flux, eflux), npoly=3)
    base = [5600.0, 4.3, -0.2, 0.1, 12.0, 4.0, np.nan]
    coef = [1.02, 0.03, -0.01]
    other = np.linspace(obs[40], obs[-60], 333)
    cases = [(28000.0, obs, False), (28000.0, other, False), (28000.0, other, True), (31000.0, other[::2].copy(), True),
             (28000.0, None, False), (28000.0, None, True), (np.nan, None, False), (np.nan, None, True)]
    for R, grid, poly in cases:
        pars = base + [R] + (coef if poly else [])
        w, f = GM.genspec(pars, outwave=grid, modpoly=poly)
        wo, fo = O.genspec(raw, pars, outwave=grid, modpoly=poly)
        np.testing.assert_allclose(w, wo, rtol=1e-15)
        assert np.array_equal(np.isnan(f), np.isnan(fo))
        ok = ~np.isnan(fo)
        assert np.abs(f[ok] - fo[ok]).max() <= 2e-6, (R, None if grid is None else len(grid), poly)
    # the fit's own context still has its observed spectrum bound
    th = np.full((1, GM.engine.ncols), np.nan)
    th[0, :8] = base + [28000.0]
    th[0, 8:11] = coef
    lnl = GM.engine.lnlike_batch(th).cpu().numpy()[0]
    ref = -0.5 * O.chi2_spec(O.genspec(raw, base + [28000.0] + coef, outwave=obs, modpoly=True)[1], flux, eflux)
    assert abs(lnl - ref) <= lnl_tol(ref)


def test_long_rows_against_reference_golden(tmp_path, golden):
    """Sizes past the former 8192-pixel caps, against vectors frozen from the reference (g12): an LSF vector on a
    20 000-pixel model (payne_lsf_kernel<GLOBAL>: buffers in global memory, median by radix selection), a 9 001-pixel
    continuum network (median by selection), and a continuum network stored in kK, which the reference uses as stored
    (only the spectral net gets the Teff/1000 fix, ystpred.py:76-85)."""
    from thepayne_amd.predict.ystpred import PayneSpecPredict
    g = golden("g12_long_rows")
    net = synth.make_yst_net(npix=20000, H=16, seed=41, line_depth=0.3)
    PP = PayneSpecPredict(nnpath=_save_yst(tmp_path, net, "yst20k.npz"), NNtype='YST1', b_max=4)
    lab = g["lsf_label"]
    for b, lsf in enumerate(g["lsf_lsfs"]):
        for c, (vrad, vrot) in enumerate(g["lsf_rows"]):
            w, f = PP.getspec(rad_vel=vrad, rot_vel=vrot, inst_R=lsf, outwave=g["lsf_obs"], Teff=lab[0], logg=lab[1],
                              feh=lab[2], afe=lab[3])
            assert not np.isnan(f).any() and np.abs(f - g["lsf_final"][b, c]).max() < 2e-6, (b, c)
    raw = synth.make_yst_net(npix=1024, H=64, seed=5, line_depth=0.3)
    spath = _save_yst(tmp_path, raw)
    for tag, npc, kk in (("long", 9001, False), ("kk", 600, True)):
        cnet = synth.make_cont_net(npix=npc, lam_lo=5140.0, lam_hi=5190.0)
        if kk:
            cnet["x_min"][0] /= 1000.0
            cnet["x_max"][0] /= 1000.0
        PC = PayneSpecPredict(nnpath=spath, Cnnpath=_save_yst(tmp_path, cnet, "cont_%s.npz" % tag), NNtype='YST1')
        for i, l in enumerate(g["cont_labels"]):
            cref = g["cont_" + tag][i]                       # (the kK net's outputs cross zero: error relative to the row's scale)
            assert np.abs(PC.predictcont(list(l)) - cref).max() < 5e-6 * np.abs(cref).max(), tag
            for j, (vrad, vrot, R) in enumerate(g["cont_rows"]):
                _, f = PC.getspec(rad_vel=vrad, rot_vel=vrot, inst_R=2.355 * R, outwave=g["cont_obs"], Teff=l[0], logg=l[1],
                                  feh=l[2], afe=l[3])
                ref = g["final_" + tag][i, j]
                assert np.array_equal(np.isnan(f), np.isnan(ref)) and np.nanmax(np.abs(f - ref)) < 2e-6, (tag, i, j)


@pytest.mark.parametrize("variant", [64, 128], ids=["select_median", "lsf_global"])
def test_long_row_forms_on_the_short_goldens(tmp_path, golden, variant):
    """The forms long rows take (medians by radix selection, LSF buffers in global memory) forced onto the small
    problems of g8 / g9, whose every case the reference froze."""
    from thepayne_amd.predict.ystpred import PayneSpecPredict
    raw = synth.make_yst_net(npix=1024, H=64, seed=5, line_depth=0.3)
    spath = _save_yst(tmp_path, raw)
    if variant == 64:
        g = golden("g8_continuum")
        for tag, (lo, hi, npc) in {"full": (5140.0, 5190.0, 600), "short": (5140.0, 5170.0, 333)}.items():
            cnet = synth.make_cont_net(npix=npc, lam_lo=lo, lam_hi=hi)
            PP = PayneSpecPredict(nnpath=spath, Cnnpath=_save_yst(tmp_path, cnet, "c_%s.npz" % tag), NNtype='YST1', variant=variant)
            for i, l in enumerate(g["labels"]):
                for j, (vrad, vrot, R) in enumerate(g["rows"]):
                    _, f = PP.getspec(rad_vel=vrad, rot_vel=vrot, inst_R=2.355 * R, outwave=g["obs"], Teff=l[0], logg=l[1],
                                      feh=l[2], afe=l[3])
                    ref = g["final_" + tag][i, j]
                    assert np.array_equal(np.isnan(f), np.isnan(ref)) and np.nanmax(np.abs(f - ref)) < 2e-6, (tag, i, j)
    else:
        g = golden("g9_lsf")
        PP = PayneSpecPredict(nnpath=spath, NNtype='YST1', variant=variant)
        for a, l in enumerate(g["labels"]):
            for b, lsf in enumerate(g["lsfs"]):
                for c, (vrad, vrot) in enumerate(g["rows"]):
                    _, f = PP.getspec(rad_vel=vrad, rot_vel=vrot, inst_R=lsf, outwave=g["obs"], Teff=l[0], logg=l[1],
                                      feh=l[2], afe=l[3])
                    assert np.abs(f - g["final"][a, b, c]).max() < 2e-6, (a, b, c)


def test_tiny_rotation_corner_is_fixed_not_replicated(tmp_path, golden):
    """rot_vel = 1e-3 km/s on 32 768 pixels.  The reference evaluates the rotational taper
    j1(u)/u - 3 cos(u)/2u^2 + 3 sin(u)/2u^3 (smoothing.py:616-617) in fp64 where u ~ 1e-8 k: terms of 1e15 cancel to ~1,
    so ITS lowest Fourier bins carry noise of order 1e-2 and its spectrum differs from the exact convolution (which at
    1/3000 of a pixel is the identity to 1e-9) by up to a few 1e-5.  SURVEY appendix B's default is to replicate the
    reference's numerics; this corner is FIXED instead: the kernel interpolates the analytic
    taper.  The test pins both statements against the reference's own output (g12): the build agrees with the exact
    answer to 1e-6, and the reference's output is within 5e-5 of it -- the measured size of the reference's noise,
    which is what tests/test_fuzz_gpu.py allows in this corner."""
    from thepayne_amd.predict.ystpred import PayneSpecPredict
    g = golden("g12_long_rows")
    rnet = synth.make_yst_net(npix=32768, H=16, seed=43, line_depth=0.3)
    PP = PayneSpecPredict(nnpath=_save_yst(tmp_path, rnet, "yst32k.npz"), NNtype='YST1', b_max=2)
    rl = g["rot_label"]
    kw = dict(Teff=rl[0], logg=rl[1], feh=rl[2], afe=rl[3])
    _, exact = PP.getspec(**kw)                                   # no rotation: what a 1e-3 km/s kernel leaves unchanged
    _, got = PP.getspec(rot_vel=float(g["rot_values"][0]), **kw)
    ref = g["rot_after"][0]
    inner = slice(2, -2)                                          # (the edge rule copies pixels 1 and n-2 outwards)
    assert np.abs(got[inner] - exact[inner]).max() < 1e-6
    dev_ref = np.abs(ref[inner] - exact[inner]).max()
    assert 1e-7 < dev_ref < 5e-5, dev_ref                         # the reference's own noise, measured
    assert np.abs(got[inner] - ref[inner]).max() < 5e-5


def test_lsf_global_form_walks_the_batch_in_chunks(tmp_path):
    """The global-memory LSF form bounds its workspace by processing 256 candidates per launch: a batch of 300 must
    give, row for row, what the LDS form gives (same arithmetic apart from the median's algorithm)."""
    from thepayne_amd.engine import PayneEngine
    from thepayne_amd import nnio
    raw, obs, flux, eflux = yst_problem("small", H=64)
    lsf = 0.075 * (1.0 + 0.3 * (obs - obs.mean()) / (obs.max() - obs.min()))
    th = theta_full(synth.draw_candidates(300, seed=12))
    th[:, 7] = np.nan
    res = []
    for variant in (0, 128):
        eng = PayneEngine(nnio.normalize_spec_net(raw), obs=(obs, flux, eflux), b_max=300, variant=variant)
        eng.set_lsf(lsf)
        res.append(eng.lnlike_batch(th).cpu().numpy())
        eng.close()
    assert np.all(np.isfinite(res[0])) and np.allclose(res[0], res[1], rtol=1e-9, atol=1e-6)
    L = O.OracleLikelihood(raw, obs, flux, eflux, [p for p in SPEC_PARS if p != 'Inst_R'], fixedpars={'Inst_R': lsf})
    for k in (0, 255, 256, 299):                                   # both sides of the chunk boundary, against the oracle
        ref = L.lnlikefn(list(th[k, :6]))
        assert abs(res[1][k] - ref) <= lnl_tol(np.array([ref]))[0], k


def test_smoothspec_branches_off_the_sampler_path(tmp_path, golden):
    """PayneSpecPredict.smoothspec with fftsmooth=False (smooth_vel / smooth_wave / smooth_lsf) and smoothtype 'lambda'
    (smooth_wave_fft): payne_smooth_direct against vectors frozen from the reference's smoothspec (g13), through the
    class method and through the reference's module path Payne.utils.smoothing.smoothspec."""
    from thepayne_amd.predict.ystpred import PayneSpecPredict
    from Payne.utils.smoothing import smoothspec
    from test_oracle_golden import G13_CALLS, g13_call
    g = golden("g13_smoothspec")
    raw = synth.make_yst_net(npix=256, H=16, seed=3)
    PP = PayneSpecPredict(nnpath=_save_yst(tmp_path, raw), NNtype='YST1')
    for key in G13_CALLS:
        ref = g[key]
        for fn in (PP.smoothspec, smoothspec):
            got = g13_call(fn, g, key)
            assert np.array_equal(np.isnan(got), np.isnan(ref)), key
            tol = 1e-6 if "fft" in key else 1e-9                   # the FFT branch runs the likelihood's fp32 transform
            assert np.nanmax(np.abs(got - ref)) < tol, (key, np.nanmax(np.abs(got - ref)))
    with pytest.raises(ValueError):                                # smooth_wave: target sigma below the input's (:381-383)
        PP.smoothspec(g["wave"], g["spec"], 0.05, outwave=g["outwave"], smoothtype='lambda', fftsmooth=False, inres=0.08)
    # an output grid beyond the input's range leaves an empty mask: the quadrature branches return NaN (the reference's
    # trapz(.)/trapz(.) over an empty set), not the "sigma too low" error
    far = g["wave"][-1] * 1.5 + np.arange(5.0)
    with np.errstate(all="ignore"):
        assert np.isnan(PP.smoothspec(g["wave"], g["spec"], 0.3, outwave=far, smoothtype='lambda', fftsmooth=False)).all()
        assert np.isnan(PP.smoothspec(g["wave"], g["spec"], 30.0, outwave=far, smoothtype='vel', fftsmooth=False)).all()
    # the FFT velocity branches still go through the likelihood's kernels
    a = PP.smoothspec(g["wave"], np.nan_to_num(g["spec"], nan=1.0), 30000.0, outwave=g["outwave"], smoothtype='R')
    b = O.smooth_R(g["wave"], np.nan_to_num(g["spec"], nan=1.0), 30000.0, g["outwave"], np.inf)
    assert np.nanmax(np.abs(a - b)) < 1e-6


def test_smoothspec_fft_branches_with_any_arguments(tmp_path, golden):
    """The argument combinations of smoothspec's FFT branches that the sampler's path never uses and rounds 1-3 refused: 'vsini' onto
    another grid and with inres (payne_smooth_batch stage 4: the rotation stage's result interpolated from its own resampled grid),
    min_wave_smooth / max_wave_smooth (the input restricted, the result back on all of wave, NaN outside), an LSF vector given on
    the INPUT grid with another output grid (payne_ctx_set_lsf_on) -- against vectors frozen from the reference's smoothspec (g15)."""
    from thepayne_amd.predict.ystpred import PayneSpecPredict
    from Payne.utils.smoothing import smoothspec
    from test_oracle_golden import g15_calls
    g = golden("g15_smoothspec_fft")
    raw = synth.make_yst_net(npix=256, H=16, seed=3)
    PP = PayneSpecPredict(nnpath=_save_yst(tmp_path, raw), NNtype='YST1')
    for key, kw in g15_calls(g).items():
        kw = dict(kw)
        res = kw.pop("resolution")
        ref = g[key]
        for fn in (lambda: PP.smoothspec(g["wave"], g["spec"], res, **kw), lambda: smoothspec(g["wave"], g["spec"], res, **kw)):
            got = fn()
            assert got.shape == ref.shape and np.array_equal(np.isnan(got), np.isnan(ref)), (key, int(np.isnan(got).sum()), int(np.isnan(ref).sum()))
            tol = 2e-6 if key.startswith("lsf") else 1e-6                # (the LSF branch's stated allowance, as in test_fuzz_gpu)
            assert np.nanmax(np.abs(got - ref)) < tol, (key, np.nanmax(np.abs(got - ref)))
    with pytest.raises(ValueError):                                # limits that keep fewer than 8 input pixels
        PP.smoothspec(g["wave"], g["spec"], 1e6, smoothtype='R', min_wave_smooth=5170.0, max_wave_smooth=5170.2)   # (~5 pixels kept)


def test_two_documented_deviations_from_the_reference(tmp_path):
    """Where this build does NOT do what the reference does, each pinned by name:
    (1) an output grid that misses the Doppler-shifted model entirely: the reference's mask is empty and numpy raises
        ValueError on ``wave[mask].min()`` (Payne/utils/smoothing.py:653, zero-size reduction); a batched kernel cannot
        raise for one row, so the build returns an all-NaN spectrum (and a NaN likelihood);
    (2) an instrumental mask of fewer than 8 model pixels: the reference runs its FFT on 1-4 points (or fails inside
        np.interp); the build returns NaN for every pixel of that spectrum."""
    from thepayne_amd.predict.ystpred import PayneSpecPredict
    raw = synth.make_yst_net(npix=1024, H=32, seed=9, line_depth=0.3)
    PP = PayneSpecPredict(nnpath=_save_yst(tmp_path, raw), NNtype='YST1')
    kw = dict(Teff=5500.0, logg=4.2, feh=-0.1, afe=0.1, rad_vel=5.0, rot_vel=2.0)
    # (1) the output grid lies 40 Angstrom beyond the red end of the model
    far = np.linspace(raw["wavelength"][-1] + 40.0, raw["wavelength"][-1] + 45.0, 50)
    with pytest.raises(ValueError):
        O.getspec(raw, inst_R=2.355 * 28000.0, outwave=far, vmic=np.nan, **kw)             # the reference's behaviour
    _, f = PP.getspec(inst_R=2.355 * 28000.0, outwave=far, **kw)
    assert f.shape == far.shape and np.isnan(f).all()                                     # the build's
    # (2) three output pixels inside one model pixel, at a resolution whose +-20 sigma pad adds nothing: a 2-pixel mask
    w = raw["wavelength"]
    tight = w[500] + (w[501] - w[500]) * np.array([0.3, 0.5, 0.7])
    R_huge = 2.0e7                                                                         # pad 20/R = 1e-6: << one pixel
    n_mask = int(((w * (1 + 5.0 / 299792.458) > tight.min() * (1 - 20.0 / R_huge)) &
                  (w * (1 + 5.0 / 299792.458) < tight.max() * (1 + 20.0 / R_huge))).sum())
    assert n_mask < 8
    sharp = dict(raw); sharp["resolution"] = 1.0e9             # (a net that claims R = 1e9: R_huge is then a legal target)
    PS = PayneSpecPredict(nnpath=_save_yst(tmp_path, sharp, "sharp.npz"), NNtype='YST1')
    with np.errstate(all="ignore"), pytest.raises(ValueError):
        O.getspec(sharp, inst_R=R_huge, outwave=tight, vmic=np.nan, **kw)                  # the reference: numpy's irfft refuses
    _, f = PS.getspec(inst_R=R_huge, outwave=tight, **kw)
    assert np.isnan(f).all()
    _, ok = PS.getspec(inst_R=2.355 * 28000.0, outwave=tight, **kw)                       # an ordinary mask on the same net
    assert np.isfinite(ok).all()


def _integration_block():
    """The python block of INTEGRATION.md section 3, as text."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = text[text.index("## 3. The stub a maintainer adds"):text.index("## 4. Using the batch with a sampler")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert len(blocks) == 1
    return blocks[0]


def test_the_documented_binding_runs_as_written(golden, monkeypatch):
    """INTEGRATION.md section 3 is the reference-side evidence of the boundary: the ctypes stub a maintainer adds to
    Payne/fitting/likelihood.py.  It is executed here exactly as printed there, on a stand-in for the reference's likelihood object
    whose network carries the attribute names of Payne/predict/ystpred.py:22-38, and must reproduce the values frozen from the
    reference (g4: 512 prior draws at C2) at SURVEY 8(d)'s tolerance.  A renamed struct field, a changed argument order or a
    stale column map fails this test."""
    import os
    import types
    from thepayne_amd import _lib
    monkeypatch.setenv("PAYNE_HIP_LIB", _lib.LIB_PATH)
    ns = {}
    exec(compile(_integration_block(), "INTEGRATION.md#3", "exec"), ns)
    # the structures of the stub against the ones the product path binds (field names, order, types)
    for doc, own in (("payne_layer", _lib.Layer), ("payne_model_desc", _lib.ModelDesc), ("payne_obs_desc", _lib.ObsDesc),
                     ("payne_opts", _lib.Opts)):
        a, b = ns[doc], own
        assert [f[0] for f in a._fields_] == [f[0] for f in b._fields_], doc
        assert ctypes_sizeof(a) == ctypes_sizeof(b), doc
    g = golden("g4_lnlike_c2")
    cfg = synth.CONFIGS["C2"]
    raw = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], seed=0)
    anns = types.SimpleNamespace(**{k: raw[k] for k in ("w_array_0", "w_array_1", "w_array_2", "b_array_0", "b_array_1", "b_array_2")},
                                 xmin=raw["x_min"], xmax=raw["x_max"], wavelength=raw["wavelength"], resolution=raw["resolution"])

    class likelihood(ns["HipLikelihood"]):                    # what the reference's class looks like to the stub
        def __init__(self, fitargs, fitpars_i, fixedpars):
            self.GM = types.SimpleNamespace(PP=types.SimpleNamespace(anns=anns))
            self.fitpars_i, self.fixedpars = list(fitpars_i), dict(fixedpars)
            self.ndim = len(self.fitpars_i)
            self._hip_init(fitargs)

    fitargs = {'obs_wave_fit': g["obs_wave"], 'obs_flux_fit': g["obs_flux"], 'obs_eflux_fit': g["obs_eflux"]}
    L = likelihood(fitargs, SPEC_PARS, {})
    lnl = L.lnlike_batch(g["theta"])
    err = np.abs(lnl - g["lnlike"])
    assert np.all(err <= lnl_tol(g["lnlike"])), (err.max(), int(np.argmax(err)))
    # more rows than b_max, the scalar contract, parsdict as the reference leaves it
    big = np.tile(g["theta"], (2, 1))[:700]
    assert np.array_equal(L.lnlike_batch(big)[512:], lnl[:188])
    assert L.lnlikefn(g["theta"][3]) == lnl[3] and L.parsdict['Teff'] == g["theta"][3, 0]
    L._hip_close()
    # a fixed parameter is merged like likelihood.py:47-48
    free = [p for p in SPEC_PARS if p != 'Vrot']
    L2 = likelihood(fitargs, free, {'Vrot': float(g["theta"][5, 5])})
    row = np.delete(g["theta"][5], 5)
    assert abs(L2.lnlikefn(row) - g["lnlike"][5]) <= lnl_tol(g["lnlike"][5])
    L2._hip_close()
    # errors come back as the library's message
    class bad(likelihood):
        def _hip_init(self, fitargs):
            ns["HipLikelihood"]._hip_init(self, fitargs, b_max=0)
    with pytest.raises(RuntimeError, match="b_max"):
        bad(fitargs, SPEC_PARS, {})


def ctypes_sizeof(t):
    import ctypes
    return ctypes.sizeof(t)
