#!/usr/bin/env python
"""Freeze outputs of the UNMODIFIED reference as golden vectors.

Run in the survey container only (needs /root/reference):

    python oracle/gen_golden.py            # writes tests/golden/*.npz
                                           # and thepayne_amd/data/highav_table.json

Inputs are regenerated from seeds by ``thepayne_amd.synth`` on both sides, so
the fixtures hold seeds, theta vectors and the reference's OUTPUTS only (plus
the noisy observed spectra, so they do not drift with numpy's RNG).  The
reference source itself is never copied.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_shim as rs  # noqa: E402

rs.install()
from thepayne_amd import synth  # noqa: E402

import scipy  # noqa: E402
from Payne.predict import ystpred, predictspec, predictsed, highred  # noqa: E402
from Payne.fitting.likelihood import likelihood  # noqa: E402
from Payne.fitting.prior import prior  # noqa: E402
from Payne.fitting.fitstar import lnprobfn  # noqa: E402
from Payne.fitting.fitutils import polycalc, airtovacuum  # noqa: E402
from Payne.utils import smoothing  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
VERS = dict(numpy=np.__version__, scipy=scipy.__version__)

SPEC_PARS = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Inst_R']
ALL_PARS = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Vmic', 'Inst_R',
            'log(R)', 'Dist', 'log(A)', 'Av', 'Rv', 'CarbonScale']


def save(name, **arrs):
    arrs["versions"] = np.array(json.dumps(VERS))
    path = os.path.join(GOLD, name)
    np.savez_compressed(path, **arrs)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path + ".npz" if not path.endswith(".npz") else path) / 1e3))


def fitpars_for(on, npoly=0):
    names = list(ALL_PARS) + ['pc_%d' % i for i in range(npoly)]
    bools = {p: (p in on or p.startswith('pc_')) for p in names}
    return [names, bools]


# ------------------------------------------------------------------ G1
def g1_ann():
    labels = np.array([
        [5770, 4.44, 0.0, 0.0], [3500, 0.0, -2.5, -0.2], [8000, 5.5, 0.5, 0.6], [3500, 5.5, -2.5, 0.6],
        [8000, 0.0, 0.5, -0.2], [4500, 2.5, -1.0, 0.2], [6500, 4.0, -0.5, 0.0], [5000, 4.6, 0.2, 0.1],
        [7200, 3.9, -2.0, 0.4], [4100, 1.2, 0.3, -0.1], [5772, 4.438, 0.01, 0.02], [6000, 4.0, 0.0, 0.0],
        [3600, 5.0, -0.3, 0.3], [7900, 4.3, 0.45, 0.55], [5200, 3.3, -1.7, 0.25], [6800, 4.9, 0.1, -0.15]])
    out = {"labels": labels}
    net = synth.make_yst_net(npix=256, H=48, seed=3)
    rs.register_yst('/g1/yst.h5', net)
    PP = ystpred.PayneSpecPredict(nnpath='/g1/yst.h5', NNtype='YST1')
    out["yst"] = np.array([PP.predictspec(list(l)) for l in labels])
    # Teff/1000 convention of ystpred.py:76-79
    netk = synth.make_yst_net(npix=256, H=48, seed=3)
    netk["x_min"][0] /= 1000.0
    netk["x_max"][0] /= 1000.0
    rs.register_yst('/g1/ystk.h5', netk)
    PPk = ystpred.PayneSpecPredict(nnpath='/g1/ystk.h5', NNtype='YST1')
    out["yst_kfix"] = np.array([PPk.predictspec(list(l)) for l in labels])
    # 5-label (vmic) network
    net5 = synth.make_yst_net(npix=256, H=48, seed=4, D=5)
    rs.register_yst('/g1/yst5.h5', net5)
    PP5 = ystpred.PayneSpecPredict(nnpath='/g1/yst5.h5', NNtype='YST1')
    lab5 = np.hstack([labels, np.linspace(0.5, 2.5, len(labels))[:, None]])
    out["labels5"] = lab5
    out["yst5"] = np.array([PP5.predictspec(list(l)) for l in lab5])
    for kind in ("LinNet", "SMLP"):
        tn = synth.make_torch_net(kind, npix=256, seed=7)
        rs.register_torchnet('/g1/%s.h5' % kind, tn)
        P = predictspec.PayneSpecPredict(nnpath='/g1/%s.h5' % kind, NNtype=kind)
        out[kind.lower()] = np.array([P.predictspec(list(l)) for l in labels])
    save("g1_ann", **out)


# ------------------------------------------------------------------ G2/G3
def g2_getspec():
    net = synth.make_yst_net(npix=1024, H=64, seed=5, line_depth=0.3)
    rs.register_yst('/g2/yst.h5', net)
    PP = ystpred.PayneSpecPredict(nnpath='/g2/yst.h5', NNtype='YST1')
    obs = synth.obs_grid(net["wavelength"], 900, inset=1.0)
    lab = dict(Teff=5300.0, logg=4.1, feh=-0.3, afe=0.15)
    rows = []
    for vrad in (-300.0, -10.0, 0.0, 10.0, 300.0):
        for vrot in (0.0, 1e-3, 0.5, 5.0, 50.0):
            for R in (10000.0, 30000.0):
                rows.append((vrad, vrot, R))
    for vrad in (0.0, 10.0):
        for vrot in (0.0, 5.0):
            for R in (40000.0, np.nan):        # above the ANN's own R -> NaN ; NaN -> plain interp
                rows.append((vrad, vrot, R))
    rows = np.array(rows)
    final, after_rot, masks = [], {}, []
    raw = PP.predictspec([lab['Teff'], lab['logg'], lab['feh'], lab['afe']])
    for vrad, vrot, R in rows:
        with np.errstate(all="ignore"):
            w, f = PP.getspec(rad_vel=vrad, rot_vel=vrot, vmic=np.nan, inst_R=2.355 * R, outwave=obs, **lab)
        final.append(f)
        if vrot not in after_rot:
            w1, f1 = PP.getspec(rot_vel=vrot, **lab)
            after_rot[vrot] = f1
        # G3: the data-dependent mask of the R stage (smoothing.py:631-647)
        mw = net["wavelength"] * (1.0 + vrad / ystpred.speedoflight) if vrad != 0.0 else net["wavelength"]
        if np.isfinite(R):
            m = smoothing.mask_wave(mw, width=2.355 * R, outwave=obs)
            masks.append((int(np.argmax(m)), int(m.sum())))
        else:
            masks.append((-1, -1))
    vr = np.array(sorted(after_rot))
    save("g2_getspec", theta_rows=rows, labels=np.array([lab['Teff'], lab['logg'], lab['feh'], lab['afe']]),
         obs_wave=obs, raw=raw, vrot_values=vr, after_rot=np.array([after_rot[v] for v in vr]),
         final=np.array(final), mask_first_count=np.array(masks))


# ------------------------------------------------------------------ G4
def _spec_problem(cfg, seed_noise=0, H=300, modpoly=False):
    net = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=H, seed=0)
    path = '/g4/yst_%d.h5' % cfg["npix"]
    rs.register_yst(path, net)
    PP = ystpred.PayneSpecPredict(nnpath=path, NNtype='YST1')
    obs = synth.obs_grid(net["wavelength"], cfg["nobs"])
    T = synth.TRUTH
    _, clean = PP.getspec(Teff=T["Teff"], logg=T["logg"], feh=T["feh"], afe=T["afe"], rad_vel=T["vrad"],
                          rot_vel=T["vrot"], vmic=np.nan, inst_R=2.355 * T["inst_R"], outwave=obs)
    rng = np.random.default_rng(seed_noise)
    flux = clean + rng.normal(0, 0.01, len(obs))
    eflux = np.full(len(obs), 0.01)
    return net, path, obs, flux, eflux


def g4_lnlike():
    cfg = synth.CONFIGS["C2"]
    net, path, obs, flux, eflux = _spec_problem(cfg)
    fitargs = {'obs_wave_fit': obs, 'obs_flux_fit': flux, 'obs_eflux_fit': eflux,
               'specANNpath': path, 'NNtype': 'YST1', 'fixedpars': {}}
    fitpars = fitpars_for(SPEC_PARS)
    runbools = [True, False, False, False, False]
    L = likelihood(fitargs, fitpars, runbools)
    P = prior(fitargs, synth.demo_priordict(), fitpars, runbools)
    rng = np.random.default_rng(1)
    u = rng.uniform(size=(512, 7))
    theta = np.array([P.priortrans(ui) for ui in u])
    lnl = np.array([L.lnlikefn(t) for t in theta])
    lnp = np.array([lnprobfn(t, L, P) for t in theta])
    save("g4_lnlike_c2", u=u, theta=theta, lnlike=lnl, lnprob=lnp, obs_wave=obs, obs_flux=flux, obs_eflux=eflux)

    # modpoly (3 Chebyshev blaze coefficients) on a smaller problem
    cfg = synth.CONFIGS["small"]
    net, path, obs, flux, eflux = _spec_problem(cfg, H=64)
    fitargs = {'obs_wave_fit': obs, 'obs_flux_fit': flux, 'obs_eflux_fit': eflux,
               'specANNpath': path, 'NNtype': 'YST1', 'fixedpars': {}}
    fitpars = fitpars_for(SPEC_PARS, npoly=3)
    runbools = [True, False, True, False, False]
    pd = synth.demo_priordict()
    pd['blaze_coeff'] = [[0.0, 0.05], [0.0, 0.02], [0.0, 0.01]]
    L = likelihood(fitargs, fitpars, runbools)
    P = prior(fitargs, pd, fitpars, runbools)
    u = np.random.default_rng(2).uniform(size=(64, 10))
    theta = np.array([P.priortrans(ui) for ui in u])
    lnl = np.array([L.lnlikefn(t) for t in theta])
    save("g4_lnlike_modpoly", u=u, theta=theta, lnlike=lnl, obs_wave=obs, obs_flux=flux, obs_eflux=eflux)

    # joint spectrum + photometry (photscale: log(A), Av) and (log(R), Dist, Av)
    phot = synth.make_phot_nets()
    rs.register_phot('/g4/phot/', phot)
    obs_phot = {f: [5.0 + 0.1 * i, 0.05] for i, f in enumerate(phot["filters"])}
    for tag, photscale, on in (("scaled", True, SPEC_PARS + ['log(A)', 'Av']),
                               ("dist", False, SPEC_PARS + ['log(R)', 'Dist', 'Av'])):
        fitargs = {'obs_wave_fit': obs, 'obs_flux_fit': flux, 'obs_eflux_fit': eflux,
                   'specANNpath': path, 'NNtype': 'YST1', 'fixedpars': {},
                   'photANNpath': '/g4/phot/', 'obs_phot': obs_phot}
        fitpars = fitpars_for(on)
        runbools = [True, True, False, photscale, False]
        pd = synth.demo_priordict()
        pd['log(A)'] = {'pv_uniform': [-3.0, 7.0]}
        pd['Av'] = {'pv_uniform': [0.0, 7.0]}          # reaches the av >= 5 branch
        pd['log(R)'] = {'pv_uniform': [-0.5, 0.5]}
        pd['Dist'] = {'pv_uniform': [10.0, 1000.0]}
        L = likelihood(fitargs, fitpars, runbools)
        P = prior(fitargs, pd, fitpars, runbools)
        nd = len(L.fitpars_i)
        u = np.random.default_rng(3).uniform(size=(64, nd))
        theta = np.array([P.priortrans(ui) for ui in u])
        lnl = np.array([L.lnlikefn(t) for t in theta])
        save("g4_lnlike_joint_" + tag, u=u, theta=theta, lnlike=lnl, fitpars_i=np.array(L.fitpars_i),
             obs_wave=obs, obs_flux=flux, obs_eflux=eflux,
             obs_mag=np.array([v[0] for v in obs_phot.values()]), obs_magerr=np.array([v[1] for v in obs_phot.values()]))


# ------------------------------------------------------------------ G5
def g5_sed():
    phot = synth.make_phot_nets()
    rs.register_phot('/g5/phot/', phot)
    S = predictsed.FastPayneSEDPredict(usebands=phot["filters"], nnpath='/g5/phot/')
    rng = np.random.default_rng(5)
    n = 24
    pars = np.column_stack([
        np.log10(rng.uniform(3500, 9000, n)), rng.uniform(0, 5, n), rng.uniform(-2, 0.5, n), rng.uniform(-0.2, 0.6, n),
        np.concatenate([rng.uniform(0, 4.9, n - 8), rng.uniform(5.0, 9.0, 8)]),   # av, last 8 use the high-Av branch
        rng.uniform(2.5, 4.5, n), rng.uniform(-1, 1, n), rng.uniform(10, 5000, n), rng.uniform(-2, 3, n)])
    pars[n - 8, 4] = 5.0       # exactly on the branch boundary
    m_dist = np.array([S.sed(logt=p[0], logg=p[1], feh=p[2], afe=p[3], av=p[4], rv=p[5], logl=p[6], dist=p[7]) for p in pars])
    m_scal = np.array([S.sed(logt=p[0], logg=p[1], feh=p[2], afe=p[3], av=p[4], rv=p[5], logA=p[8]) for p in pars])
    bc = np.array([S.anns.eval([10.0 ** p[0], p[1], p[2], p[3], p[4], p[5]]) for p in pars])
    save("g5_sed", pars=pars, mags_dist=m_dist, mags_scaled=m_scal, bc=bc, hiav=np.array(S.HiAv.Avlist, dtype=float))
    # the high-Av coefficient table is DATA the product needs too
    H = highred.highAv(list(predictsed._ALLFILTERS))
    table = {f: [None if not np.isfinite(v) else float(v) for v in row]
             for f, row in zip(predictsed._ALLFILTERS, H.Avlist)}
    os.makedirs(os.path.join(ROOT, "thepayne_amd", "data"), exist_ok=True)
    with open(os.path.join(ROOT, "thepayne_amd", "data", "highav_table.json"), "w") as fh:
        json.dump({"columns": ["a1", "b1", "a2", "b2", "c2"], "filters": table}, fh, indent=0)


# ------------------------------------------------------------------ G6
def g6_prior():
    u = np.array([0.0, 1e-6, 0.01, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99, 1 - 1e-9, 1.0])
    out = {"u": u}
    fitargs = {'fixedpars': {}}
    kinds = {
        "uniform": {'pv_uniform': [4000.0, 8000.0]},
        "uniform_rev": {'pv_uniform': [8000.0, 4000.0]},
        "gaussian": {'pv_gaussian': [5770.0, 100.0]},
        "tgaussian": {'pv_tgaussian': [25000.0, 37000.0, 28800.0, 1000.0]},
        "exp": {'pv_exp': [0.0, 2.0]},
        "texp": {'pv_texp': [0.0, 10.0, 2.0]},
        "default": None,
    }
    for par in ('Teff', 'Inst_R', 'Vrot'):
        for kname, spec in kinds.items():
            pd = {} if spec is None else {par: spec}
            fitpars = fitpars_for([par])
            with np.errstate(all="ignore"):
                P = prior(fitargs, pd, fitpars, [True, False, False, False, False])
                out["spec_%s_%s" % (par, kname)] = np.array([P.priortrans([ui])[0] for ui in u], dtype=float)
    for par in ('log(A)', 'Av', 'Dist', 'log(R)', 'Rv'):
        for kname in ("uniform", "gaussian", "tgaussian", "exp", "default"):
            spec = kinds[kname]
            pd = {} if spec is None else {par: spec}
            fitpars = fitpars_for([par])
            with np.errstate(all="ignore"):
                P = prior(fitargs, pd, fitpars, [False, True, False, True, False])
                out["phot_%s_%s" % (par, kname)] = np.array([P.priortrans([ui])[0] for ui in u], dtype=float)
    # phot-only run transforms the four atmosphere labels too (prior.py:200-229)
    fitpars = fitpars_for(['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'log(A)', 'Av'])
    pd = {'Teff': {'pv_gaussian': [5770.0, 200.0]}, 'Av': {'pv_uniform': [0.0, 1.0]}}
    P = prior(fitargs, pd, fitpars, [False, True, False, True, False])
    U = np.random.default_rng(6).uniform(size=(16, 6))
    out["photonly_u"] = U
    out["photonly_theta"] = np.array([P.priortrans(ui) for ui in U], dtype=float)
    # blaze coefficients (prior.py:180-191)
    fitpars = fitpars_for(SPEC_PARS, npoly=4)
    pd = synth.demo_priordict()
    pd['blaze_coeff'] = [[0.0, 0.5], [0.1, 0.2], [-0.05, 0.1], [0.0, 0.01]]
    P = prior(fitargs, pd, fitpars, [True, False, True, False, False])
    U = np.random.default_rng(7).uniform(size=(16, 11))
    out["blaze_u"] = U
    out["blaze_theta"] = np.array([P.priortrans(ui) for ui in U], dtype=float)
    # additive ln-priors (prior.py:274-465).  NB a 'gaussian'/'uniform' prior on an
    # atmosphere label in a JOINT run raises KeyError in the reference
    # (lnprior_phot only fills pardict_i['Teff'...] when there is no spectrum,
    # prior.py:431-435 vs :451), so the joint case carries a phot prior only.
    fitpars = fitpars_for(SPEC_PARS)
    pd = synth.demo_priordict()
    pd['Teff']['gaussian'] = [5770.0, 50.0]
    pd['[Fe/H]']['uniform'] = [-0.05, 0.08]
    P = prior(fitargs, pd, fitpars, [True, False, False, False, False])
    U = np.random.default_rng(8).uniform(size=(32, 7))
    th = np.array([P.priortrans(ui) for ui in U], dtype=float)
    out["lnprior_spec_u"] = U
    out["lnprior_spec_theta"] = th
    out["lnprior_spec"] = np.array([P.lnpriorfn(list(t)) for t in th], dtype=float)
    fitpars = fitpars_for(SPEC_PARS + ['log(A)', 'Av'])
    pd = synth.demo_priordict()
    pd['Av'] = {'pv_uniform': [0.0, 1.0], 'gaussian': [0.1, 0.05], 'uniform': [0.02, 0.9]}
    pd['log(A)'] = {'pv_uniform': [-3.0, 7.0]}
    P = prior(fitargs, pd, fitpars, [True, True, False, True, False])
    U = np.random.default_rng(9).uniform(size=(32, 9))
    th = np.array([P.priortrans(ui) for ui in U], dtype=float)
    out["lnprior_joint_u"] = U
    out["lnprior_joint_theta"] = th
    out["lnprior_joint"] = np.array([P.lnpriorfn(list(t)) for t in th], dtype=float)
    save("g6_prior", **out)


# ------------------------------------------------------------------ G7
def g7_misc():
    w = np.linspace(5153.0, 5241.0, 500)
    coefs = np.array([[1.0, 0.0, 0.0], [0.9, 0.05, -0.02], [1.1, -0.3, 0.2], [0.75, 1.0, 1.0]])
    air = np.linspace(3800.0, 9000.0, 64)
    save("g7_misc", wave=w, coefs=coefs, poly=np.array([polycalc(c, w) for c in coefs]),
         air=air, vac=airtovacuum(air))


# ------------------------------------------------------------------ G8
def g8_continuum():
    """PayneSpecPredict with a continuum network (Cnnpath, ystpred.py:81-85, 191-209): predictcont and
    getspec for two continuum grids -- one covering the spectral ANN, one that leaves its red end NaN."""
    net = synth.make_yst_net(npix=1024, H=64, seed=5, line_depth=0.3)
    rs.register_yst('/g8/yst.h5', net)
    obs = synth.obs_grid(net["wavelength"], 700, inset=1.5)
    labs = np.array([[5300.0, 4.1, -0.3, 0.15], [6400.0, 3.2, -1.4, 0.4], [4300.0, 4.9, 0.3, -0.1]])
    rows = np.array([(0.0, 0.0, 25000.0), (12.0, 4.0, 28000.0), (-35.0, 9.0, 18000.0), (5.0, 2.0, np.nan)])
    out = dict(obs=obs, labels=labs, rows=rows)
    for tag, (lo, hi, npc) in {"full": (5140.0, 5190.0, 600), "short": (5140.0, 5170.0, 333)}.items():
        cnet = synth.make_cont_net(npix=npc, lam_lo=lo, lam_hi=hi)
        rs.register_yst('/g8/cont_%s.h5' % tag, cnet)
        PP = ystpred.PayneSpecPredict(nnpath='/g8/yst.h5', Cnnpath='/g8/cont_%s.h5' % tag, NNtype='YST1')
        out["cont_" + tag] = np.array([PP.predictcont(list(l)) for l in labs])
        fin, native = [], []
        for l in labs:
            kw = dict(Teff=l[0], logg=l[1], feh=l[2], afe=l[3])
            with np.errstate(all="ignore"):
                native.append(PP.getspec(**kw)[1])
                for vrad, vrot, R in rows:
                    fin.append(PP.getspec(rad_vel=vrad, rot_vel=vrot, vmic=np.nan, inst_R=2.355 * R, outwave=obs, **kw)[1])
        out["native_" + tag] = np.array(native)
        out["final_" + tag] = np.array(fin).reshape(len(labs), len(rows), len(obs))
    save("g8_continuum", **out)


# ------------------------------------------------------------------ G9
def g9_lsf():
    """getspec with an LSF vector for inst_R (ystpred.py:248-269 -> smoothspec 'lsf' -> smooth_lsf_fft):
    dispersion in AA per observed pixel, three shapes; also genspec / lnlike through GenMod with the array."""
    net = synth.make_yst_net(npix=1024, H=64, seed=5, line_depth=0.3)
    rs.register_yst('/g9/yst.h5', net)
    PP = ystpred.PayneSpecPredict(nnpath='/g9/yst.h5', NNtype='YST1')
    obs = synth.obs_grid(net["wavelength"], 700, inset=1.5)
    xo = (obs - obs.mean()) / (obs.max() - obs.min())
    lsfs = np.array([np.full(len(obs), 0.08), 0.07 * (1.0 + 0.5 * xo), 0.10 + 0.04 * xo ** 2 + 0.02 * xo])
    labs = np.array([[5300.0, 4.1, -0.3, 0.15], [6400.0, 3.2, -1.4, 0.4]])
    rows = np.array([(0.0, 0.0), (12.0, 4.0), (-35.0, 9.0), (250.0, 1.5)])
    fin = np.zeros((len(labs), len(lsfs), len(rows), len(obs)))
    for a, l in enumerate(labs):
        kw = dict(Teff=l[0], logg=l[1], feh=l[2], afe=l[3])
        for b, lsf in enumerate(lsfs):
            for c, (vrad, vrot) in enumerate(rows):
                with np.errstate(all="ignore"):
                    fin[a, b, c] = PP.getspec(rad_vel=vrad, rot_vel=vrot, vmic=np.nan, inst_R=lsf, outwave=obs, **kw)[1]
    # on the model grid itself (outwave=None): the vector must have the model's length
    lsf_native = 0.09 * (1.0 + 0.3 * (net["wavelength"] - net["wavelength"].mean()) / 20.0)
    with np.errstate(all="ignore"):
        native = PP.getspec(rad_vel=8.0, rot_vel=3.0, inst_R=lsf_native, Teff=5300.0, logg=4.1, feh=-0.3, afe=0.15)
    save("g9_lsf", obs=obs, lsfs=lsfs, labels=labs, rows=rows, final=fin, lsf_native=lsf_native,
         native_wave=native[0], native=native[1])


# ------------------------------------------------------------------ G10
def g10_advanced_priors():
    """The priordict keys IMF / VROT / GAL / VTOT / AngDia as prior.py applies them: IMF and VROT through
    lnpriorfn (prior.py:286-336), GAL through the transform of Dist (prior.py:231-234); VTOT and AngDia
    leave everything unchanged in the reference (flag never set / function never called)."""
    fitargs = {'fixedpars': {}}
    rng = np.random.default_rng(10)
    out = {}
    # joint fit with log(R) + Dist: IMF and VROT need log(g), log(R)
    names = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Inst_R', 'log(R)', 'Dist', 'Av']
    fitpars = fitpars_for(names)
    n = 48
    theta = np.column_stack([rng.uniform(4000, 7000, n), rng.uniform(0.5, 5.2, n), rng.uniform(-2, 0.4, n),
                             rng.uniform(-0.1, 0.5, n), rng.uniform(-50, 50, n), rng.uniform(0, 40, n),
                             rng.uniform(20000, 35000, n), rng.uniform(-1.3, 1.6, n), rng.uniform(50, 5000, n),
                             rng.uniform(0, 1, n)])
    out["theta"] = theta
    runb = [True, True, False, False, False]
    base = {'Dist': {'pv_uniform': [10.0, 20000.0]}}
    cases = {
        "imf": dict(base, IMF={'IMF_type': 'Kroupa'}),
        "vrot": dict(base, VROT={}),
        "imf_vrot_gauss": dict(base, IMF={'IMF_type': 'Kroupa'}, VROT={}, Vrad={'gaussian': [5.0, 30.0]}),
        "vtot_angdia": dict(base, VTOT={'pmra': 30.0, 'pmdec': -12.0}, AngDia={'gaussian': [0.5, 0.05]}),
    }
    for tag, pd in cases.items():
        P = prior(fitargs, pd, fitpars, runb)
        out["lnp_" + tag] = np.array([P.lnpriorfn(list(t)) for t in theta], dtype=float)
    # photscale fit (log(A)): VROT takes mass 1, eep 350
    names_a = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Inst_R', 'log(A)', 'Av']
    P = prior(fitargs, {'VROT': {}}, fitpars_for(names_a), [True, True, False, True, False])
    theta_a = np.delete(theta, 8, axis=1)
    out["theta_a"] = theta_a
    out["lnp_vrot_logA"] = np.array([P.lnpriorfn(list(t)) for t in theta_a], dtype=float)
    # GAL: Dist = 1000 * gal_ppf(u) for two sight lines and two distance ranges
    u = np.concatenate([[0.0, 1e-6, 1e-3], np.linspace(0.01, 0.99, 29), [1 - 1e-6, 1.0]])
    out["u"] = u
    gal = {"disk": ([90.0, 2.0], [10.0, 20000.0]), "pole": ([10.0, 80.0], [100.0, 100000.0]), "nodist": ([200.0, -35.0], None)}
    fp = fitpars_for(['Dist'])
    for tag, (lb, rng_d) in gal.items():
        pd = {'GAL': {'lb_coords': lb}}
        if rng_d is not None:
            pd['Dist'] = {'pv_uniform': rng_d}
        P = prior(fitargs, pd, fp, [False, True, False, False, False])
        out["gal_" + tag] = np.array([P.priortrans([ui])[0] for ui in u], dtype=float)
        out["gal_lb_" + tag] = np.array(lb)
    save("g10_advpriors", **out)


# ------------------------------------------------------------------ G11
def g11_native_grid():
    """getspec / genspec without an output grid (outwave=None): the spectrum comes back on the Doppler-shifted
    model grid, with the instrumental stage (smoothspec puts outwave = wave, smoothing.py:140-141; the end pixels
    fall outside the resampled grid -> NaN) and without it; genspec also with the blaze polynomial."""
    from Payne.fitting.genmod import GenMod
    net = synth.make_yst_net(npix=1024, H=64, seed=0, line_depth=0.3)
    rs.register_yst('/g11/yst.h5', net)
    GM = GenMod()
    GM._initspecnn(nnpath='/g11/yst.h5', NNtype='YST1')
    base = [5600.0, 4.3, -0.2, 0.1]
    rows = np.array([(12.0, 4.0, 28000.0), (0.0, 0.0, 25000.0), (-40.0, 7.5, 30000.0), (12.0, 4.0, np.nan), (0.0, 3.0, 0.0)])
    coef = [1.02, 0.03, -0.01]
    waves, plain, poly = [], [], []
    for vrad, vrot, R in rows:
        with np.errstate(all="ignore"):
            w, f = GM.genspec(base + [vrad, vrot, np.nan, float(R)], outwave=None)
            w2, f2 = GM.genspec(base + [vrad, vrot, np.nan, float(R)] + coef, outwave=None, modpoly=True)
        assert np.array_equal(w, w2)
        waves.append(w); plain.append(f); poly.append(f2)
    save("g11_native", base=np.array(base), rows=rows, coef=np.array(coef), wave=np.array(waves),
         plain=np.array(plain), poly=np.array(poly))


# ------------------------------------------------------------------ G12
def g12_long_rows():
    """The branches the build used to cap at 8192 pixels, and the corner its tolerance excepted, at sizes past those
    caps, as the REFERENCE computes them:
      * lsf20k: getspec with an LSF vector on a 20 000-pixel model (smooth_lsf_fft, smoothing.py:482-586);
      * cont_long / cont_kk: a continuum network of 9 001 pixels (median of a row too long for an LDS sort), and one
        whose x_min / x_max are stored in kK -- PayneSpecPredict rescales the SPECTRAL net only (ystpred.py:76-85);
      * rot_tiny: rot_vel = 1e-3 km/s on a 32 768-pixel grid, where the reference's fp64 taper expression
        j1(u)/u - 3 cos(u)/2u^2 + 3 sin(u)/2u^3 (smoothing.py:616-617) is itself dominated by cancellation noise."""
    out = {}
    # ---- LSF vector on a 20 000-pixel model
    net = synth.make_yst_net(npix=20000, H=16, seed=41, line_depth=0.3)
    rs.register_yst('/g12/yst20k.h5', net)
    PP = ystpred.PayneSpecPredict(nnpath='/g12/yst20k.h5', NNtype='YST1')
    obs = synth.obs_grid(net["wavelength"], 3000, inset=30.0)
    xo = (obs - obs.mean()) / (obs.max() - obs.min())
    lsfs = np.array([np.full(len(obs), 0.09), 0.08 * (1.0 + 0.4 * xo)])
    rows = np.array([(0.0, 0.0), (-21.0, 6.0)])
    lab = np.array([5600.0, 4.3, -0.2, 0.1])
    fin = np.zeros((len(lsfs), len(rows), len(obs)))
    for b, lsf in enumerate(lsfs):
        for c, (vrad, vrot) in enumerate(rows):
            with np.errstate(all="ignore"):
                fin[b, c] = PP.getspec(rad_vel=vrad, rot_vel=vrot, vmic=np.nan, inst_R=lsf, outwave=obs,
                                       Teff=lab[0], logg=lab[1], feh=lab[2], afe=lab[3])[1]
    out.update(lsf_obs=obs, lsf_lsfs=lsfs, lsf_rows=rows, lsf_label=lab, lsf_final=fin)
    # ---- continuum networks: a long one, and one stored in kK
    snet = synth.make_yst_net(npix=1024, H=64, seed=5, line_depth=0.3)
    rs.register_yst('/g12/yst.h5', snet)
    cobs = synth.obs_grid(snet["wavelength"], 700, inset=1.5)
    labs = np.array([[5300.0, 4.1, -0.3, 0.15], [6400.0, 3.2, -1.4, 0.4]])
    crows = np.array([(0.0, 0.0, 25000.0), (12.0, 4.0, 28000.0)])
    out.update(cont_obs=cobs, cont_labels=labs, cont_rows=crows)
    for tag, npc, kk in (("long", 9001, False), ("kk", 600, True)):
        cnet = synth.make_cont_net(npix=npc, lam_lo=5140.0, lam_hi=5190.0)
        if kk:
            cnet["x_min"][0] /= 1000.0
            cnet["x_max"][0] /= 1000.0
        rs.register_yst('/g12/cont_%s.h5' % tag, cnet)
        PC = ystpred.PayneSpecPredict(nnpath='/g12/yst.h5', Cnnpath='/g12/cont_%s.h5' % tag, NNtype='YST1')
        out["cont_" + tag] = np.array([PC.predictcont(list(l)) for l in labs])
        fin = []
        for l in labs:
            for vrad, vrot, R in crows:
                with np.errstate(all="ignore"):
                    fin.append(PC.getspec(rad_vel=vrad, rot_vel=vrot, vmic=np.nan, inst_R=2.355 * R, outwave=cobs,
                                          Teff=l[0], logg=l[1], feh=l[2], afe=l[3])[1])
        out["final_" + tag] = np.array(fin).reshape(len(labs), len(crows), len(cobs))
    # ---- tiny rotation on a long grid
    rnet = synth.make_yst_net(npix=32768, H=16, seed=43, line_depth=0.3)
    rs.register_yst('/g12/yst32k.h5', rnet)
    PR = ystpred.PayneSpecPredict(nnpath='/g12/yst32k.h5', NNtype='YST1')
    rl = np.array([5900.0, 4.0, 0.1, 0.0])
    kw = dict(Teff=rl[0], logg=rl[1], feh=rl[2], afe=rl[3])
    vals = np.array([1e-3])
    with np.errstate(all="ignore"):
        rot = np.array([PR.getspec(rot_vel=v, **kw)[1] for v in vals])
    out.update(rot_label=rl, rot_values=vals, rot_after=rot)
    save("g12_long_rows", **out)


# ------------------------------------------------------------------ G13
def g13_smoothspec_branches():
    """smoothspec's branches off the sampler's path, as the reference computes them (Payne/utils/smoothing.py): the direct
    quadratures (fftsmooth=False: smooth_vel for 'vel' / 'R', smooth_wave for 'lambda', smooth_lsf for 'lsf') and the
    wavelength-space FFT ('lambda', fftsmooth=True).  Input: a 700-pixel spectrum with two NaN pixels (nan_to_num)."""
    from Payne.utils.smoothing import smoothspec
    rng = np.random.default_rng(13)
    wave = 5150.0 * (1.0 + 1.0 / 70000.0) ** np.arange(700)
    spec = 1.0 - 0.3 * np.exp(-0.5 * ((wave - 5175.0) / 0.15) ** 2) - 0.2 * np.exp(-0.5 * ((wave - 5190.0) / 0.4) ** 2) \
        + 0.01 * rng.normal(size=700)
    spec[[100, 431]] = np.nan
    outwave = np.linspace(5160.0, 5195.0, 300)
    lsf_on_wave = 0.12 * (1.0 + 0.4 * (wave - wave.mean()) / (wave.max() - wave.min()))
    out = dict(wave=wave, spec=spec, outwave=outwave, lsf_on_wave=lsf_on_wave)
    with np.errstate(all="ignore"):
        out["vel_direct"] = smoothspec(wave, spec, 8.0, outwave=outwave, smoothtype='vel', fftsmooth=False)
        out["vel_direct_inres"] = smoothspec(wave, spec, 8.0, outwave=outwave, smoothtype='vel', fftsmooth=False, inres=3.0)
        out["vel_direct_nsig"] = smoothspec(wave, spec, 8.0, outwave=outwave, smoothtype='vel', fftsmooth=False, nsigma=-1)
        out["R_direct"] = smoothspec(wave, spec, 30000.0, outwave=outwave, smoothtype='R', fftsmooth=False, inres=90000.0)
        out["R_direct_native"] = smoothspec(wave, spec, 30000.0, smoothtype='R', fftsmooth=False)
        out["lambda_direct"] = smoothspec(wave, spec, 0.2, outwave=outwave, smoothtype='lambda', fftsmooth=False)
        out["lambda_direct_inres"] = smoothspec(wave, spec, 0.2, outwave=outwave, smoothtype='lambda', fftsmooth=False, inres=0.08)
        out["lambda_direct_invel"] = smoothspec(wave, spec, 0.2, outwave=outwave, smoothtype='lambda', fftsmooth=False,
                                                inres=60000.0, in_vel=True)
        out["lambda_fft"] = smoothspec(wave, spec, 0.2, outwave=outwave, smoothtype='lambda', fftsmooth=True)
        out["lambda_fft_inres"] = smoothspec(wave, spec, 0.2, outwave=outwave, smoothtype='lambda', fftsmooth=True, inres=0.08)
        out["lambda_fft_native"] = smoothspec(wave, spec, 0.2, smoothtype='lambda', fftsmooth=True)
        out["lsf_direct"] = smoothspec(wave, spec, lsf_on_wave, outwave=outwave, smoothtype='lsf', fftsmooth=False)
        out["lsf_direct_none"] = smoothspec(wave, spec, None, outwave=outwave, smoothtype='lsf', fftsmooth=False)
    save("g13_smoothspec", **out)


# ------------------------------------------------------------------ G14
def g14_torchnets_300():
    """The reference's DEFAULT network at real size (FitPayne: NNtype='LinNet', fitstar.py:81): LinNet D -> 300 x5 -> Npix, five
    sigmoids (train/NNmodels.py:140-168), and SMLP D -> 300 x3 -> Npix, LeakyReLU (:92-137), both evaluated by torch in fp32
    (predict/predictspec.py:61-74), on the C2 shape: 4096 model pixels, 3600 observed pixels.  Frozen: predictspec for four
    label vectors (fp32, as the reference returns it), and lnlikefn for 128 (LinNet) / 32 (SMLP) draws of the demo priors."""
    cfg = synth.CONFIGS["C2"]
    labels = np.array([[5770.0, 4.44, 0.0, 0.0], [4100.0, 4.9, -0.08, 0.07], [7900.0, 4.05, 0.09, -0.1], [6250.0, 5.3, 0.03, 0.02]])
    T = synth.TRUTH
    for kind, ndraw, seed in (("LinNet", 128, 21), ("SMLP", 32, 22)):
        net = synth.make_torch_net(kind, npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=(300, 300, 300), seed=seed)
        path = '/g14/%s300.h5' % kind
        rs.register_torchnet(path, net)
        PP = predictspec.PayneSpecPredict(nnpath=path, NNtype=kind)
        raw = np.array([PP.predictspec(list(l)) for l in labels])
        assert raw.dtype == np.float32
        obs = synth.obs_grid(net["wavelength"], cfg["nobs"])
        _, clean = PP.getspec(Teff=T["Teff"], logg=T["logg"], feh=T["feh"], afe=T["afe"], rad_vel=T["vrad"],
                              rot_vel=T["vrot"], vmic=np.nan, inst_R=2.355 * T["inst_R"], outwave=obs)
        flux = clean + np.random.default_rng(0).normal(0, 0.01, len(obs))
        eflux = np.full(len(obs), 0.01)
        fitargs = {'obs_wave_fit': obs, 'obs_flux_fit': flux, 'obs_eflux_fit': eflux,
                   'specANNpath': path, 'NNtype': kind, 'fixedpars': {}}
        fitpars = fitpars_for(SPEC_PARS)
        runbools = [True, False, False, False, False]
        L = likelihood(fitargs, fitpars, runbools)
        P = prior(fitargs, synth.demo_priordict(), fitpars, runbools)
        u = np.random.default_rng(14).uniform(size=(ndraw, 7))
        theta = np.array([P.priortrans(ui) for ui in u])
        with np.errstate(all="ignore"):
            lnl = np.array([L.lnlikefn(t) for t in theta])
        # getspec on the observed grid for the first four draws (fp64, as getspec returns it)
        spec = []
        for t in theta[:4]:
            with np.errstate(all="ignore"):
                spec.append(PP.getspec(Teff=t[0], logg=t[1], feh=t[2], afe=t[3], rad_vel=t[4], rot_vel=t[5], vmic=np.nan,
                                       inst_R=2.355 * t[6], outwave=obs)[1])
        save("g14_%s300" % kind.lower(), seed=np.array(seed), labels=labels, raw=raw, u=u, theta=theta, lnlike=lnl,
             getspec4=np.array(spec), obs_wave=obs, obs_flux=flux, obs_eflux=eflux)


# ------------------------------------------------------------------ G15
def g15_smoothspec_fft_corners():
    """smoothspec's FFT branches with the argument combinations the sampler's path never uses (Payne/utils/smoothing.py:19-169):
    'vsini' onto another grid and with inres; min_wave_smooth / max_wave_smooth (outwave=None: the input restricted, the result back
    on all of wave); 'lsf' with the vector on the input grid and another output grid.  Same 700-pixel spectrum as g13."""
    from Payne.utils.smoothing import smoothspec
    rng = np.random.default_rng(13)
    wave = 5150.0 * (1.0 + 1.0 / 70000.0) ** np.arange(700)
    spec = 1.0 - 0.3 * np.exp(-0.5 * ((wave - 5175.0) / 0.15) ** 2) - 0.2 * np.exp(-0.5 * ((wave - 5190.0) / 0.4) ** 2) \
        + 0.01 * rng.normal(size=700)
    spec[[100, 431]] = np.nan
    outwave = np.linspace(5160.0, 5195.0, 300)
    lsf_on_wave = 0.12 * (1.0 + 0.4 * (wave - wave.mean()) / (wave.max() - wave.min()))
    lim = dict(min_wave_smooth=5162.0, max_wave_smooth=5193.0)
    out = dict(wave=wave, spec=spec, outwave=outwave, lsf_on_wave=lsf_on_wave, limits=np.array([lim['min_wave_smooth'], lim['max_wave_smooth']]))
    with np.errstate(all="ignore"):
        out["vsini_out"] = smoothspec(wave, spec, 12.0, outwave=outwave, smoothtype='vsini')
        out["vsini_out_inres"] = smoothspec(wave, spec, 12.0, outwave=outwave, smoothtype='vsini', inres=5.0)
        out["vsini_native_inres"] = smoothspec(wave, spec, 12.0, smoothtype='vsini', inres=5.0)
        out["vsini_limits"] = smoothspec(wave, spec, 12.0, smoothtype='vsini', **lim)
        out["vel_limits"] = smoothspec(wave, spec, 8.0, smoothtype='vel', **lim)
        out["vel_limits_inres"] = smoothspec(wave, spec, 8.0, smoothtype='vel', inres=3.0, **lim)
        out["R_limits"] = smoothspec(wave, spec, 30000.0, smoothtype='R', inres=90000.0, **lim)
        out["lsf_out"] = smoothspec(wave, spec, lsf_on_wave, outwave=outwave, smoothtype='lsf')
        out["lsf_limits"] = smoothspec(wave, spec, lsf_on_wave, smoothtype='lsf', **lim)
    save("g15_smoothspec_fft", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15"]
    for k in which:
        {"g1": g1_ann, "g2": g2_getspec, "g4": g4_lnlike, "g5": g5_sed, "g6": g6_prior, "g7": g7_misc,
         "g8": g8_continuum, "g9": g9_lsf, "g10": g10_advanced_priors, "g11": g11_native_grid, "g12": g12_long_rows, "g13": g13_smoothspec_branches, "g14": g14_torchnets_300, "g15": g15_smoothspec_fft_corners}[k]()
