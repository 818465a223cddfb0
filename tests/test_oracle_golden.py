"""The numpy oracle against golden vectors frozen from the reference itself
(oracle/gen_golden.py).  fp64 both sides -> tolerance 1e-12 (YST1 path);
torch-fp32 nets (LinNet/SMLP) to 2e-6."""
import numpy as np
import pytest

import oracle as O
from thepayne_amd import synth

SPEC_PARS = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Inst_R']


def test_g1_yst_forward(golden):
    g = golden("g1_ann")
    net = synth.make_yst_net(npix=256, H=48, seed=3)
    got = np.array([O.yst_forward(net, l) for l in g["labels"]])
    np.testing.assert_allclose(got, g["yst"], rtol=0, atol=1e-12)
    # the Teff/1000 convention is repaired at load (ystpred.py:76-79) -> same outputs
    np.testing.assert_allclose(g["yst_kfix"], g["yst"], rtol=0, atol=1e-12)
    net5 = synth.make_yst_net(npix=256, H=48, seed=4, D=5)
    got5 = np.array([O.yst_forward(net5, l) for l in g["labels5"]])
    np.testing.assert_allclose(got5, g["yst5"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("kind", ["LinNet", "SMLP"])
def test_g1_torch_nets(golden, kind):
    g = golden("g1_ann")
    net = synth.make_torch_net(kind, npix=256, seed=7)
    got = np.array([O.torchnet_forward(net, l) for l in g["labels"]])
    assert got.dtype == np.float32
    np.testing.assert_allclose(got, g[kind.lower()], rtol=0, atol=2e-6)


@pytest.mark.parametrize("kind,seed", [("LinNet", 21), ("SMLP", 22)])
def test_g14_torch_nets_at_full_width(golden, kind, seed):
    """The reference's default network at real size (LinNet 5 x 300, fitstar.py:81; SMLP 3 x 300) on the C2 shape: the reference
    runs these in torch fp32, the restatement in numpy fp32 -- the two differ by the order of the 300-term sums: one fp32 ulp."""
    g = golden("g14_%s300" % kind.lower())
    cfg = synth.CONFIGS["C2"]
    net = synth.make_torch_net(kind, npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=(300, 300, 300), seed=seed)
    got = np.array([O.torchnet_forward(net, l) for l in g["labels"]])
    assert got.dtype == np.float32 and g["raw"].dtype == np.float32
    np.testing.assert_allclose(got, g["raw"], rtol=0, atol=4e-7)
    L = O.OracleLikelihood(net, g["obs_wave"], g["obs_flux"], g["obs_eflux"], SPEC_PARS)
    th = g["theta"][:24]
    with np.errstate(all="ignore"):
        lnl = np.array([L.lnlikefn(t) for t in th])
    ref = g["lnlike"][:24]
    assert np.all(np.abs(lnl - ref) <= 2e-6 * np.abs(ref) + 5e-3)
    _, f = O.getspec(net, Teff=th[0, 0], logg=th[0, 1], feh=th[0, 2], afe=th[0, 3], rad_vel=th[0, 4], rot_vel=th[0, 5],
                     vmic=np.nan, inst_R=2.355 * th[0, 6], outwave=g["obs_wave"])
    np.testing.assert_allclose(f, g["getspec4"][0], rtol=0, atol=4e-7)


def test_g2_getspec_stages_and_masks(golden):
    g = golden("g2_getspec")
    net = synth.make_yst_net(npix=1024, H=64, seed=5, line_depth=0.3)
    lab = dict(zip(("Teff", "logg", "feh", "afe"), g["labels"]))
    obs = g["obs_wave"]
    np.testing.assert_allclose(O.yst_forward(net, g["labels"]), g["raw"], atol=1e-12)
    for v, ref in zip(g["vrot_values"], g["after_rot"]):
        _, f = O.getspec(net, rot_vel=float(v), **lab)
        np.testing.assert_allclose(f, ref, rtol=0, atol=1e-12, equal_nan=True)
    n_nan_rows = 0
    for (vrad, vrot, R), ref, (first, count) in zip(g["theta_rows"], g["final"], g["mask_first_count"]):
        with np.errstate(all="ignore"):
            _, f = O.getspec(net, rad_vel=float(vrad), rot_vel=float(vrot), vmic=np.nan,
                             inst_R=2.355 * float(R), outwave=obs, **lab)
        assert np.array_equal(np.isnan(f), np.isnan(ref))
        np.testing.assert_allclose(f, ref, rtol=0, atol=1e-12, equal_nan=True)
        n_nan_rows += int(np.isnan(ref).any())
        if np.isfinite(R):
            mw = net["wavelength"] * (1.0 + vrad / O.C_KMS_DOPPLER) if vrad != 0 else net["wavelength"]
            m = O.mask_range(mw, 2.355 * R, obs)
            assert (int(np.argmax(m)), int(m.sum())) == (first, count)
    assert n_nan_rows >= 4          # the fixture really exercises the NaN contracts


def _c2_like(golden, name):
    g = golden(name)
    return g


def test_g4_lnlike_c2(golden):
    g = golden("g4_lnlike_c2")
    cfg = synth.CONFIGS["C2"]
    net = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], seed=0)
    L = O.OracleLikelihood(net, g["obs_wave"], g["obs_flux"], g["obs_eflux"], SPEC_PARS)
    idx = np.arange(0, 512, 4)                  # 128 of the 512 draws keeps the CPU suite quick
    got = np.array([L.lnlikefn(g["theta"][i]) for i in idx])
    np.testing.assert_allclose(got, g["lnlike"][idx], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(g["lnprob"], g["lnlike"], rtol=0, atol=0)   # demo priors add 0.0
    got_p = np.array([O.lnprobfn(g["theta"][i], L) for i in idx[:8]])
    np.testing.assert_allclose(got_p, g["lnprob"][idx[:8]], rtol=1e-12, atol=1e-9)


def test_g4_lnlike_modpoly(golden):
    g = golden("g4_lnlike_modpoly")
    cfg = synth.CONFIGS["small"]
    net = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=64, seed=0)
    L = O.OracleLikelihood(net, g["obs_wave"], g["obs_flux"], g["obs_eflux"],
                           SPEC_PARS + ["pc_0", "pc_1", "pc_2"], modpoly=True)
    got = np.array([L.lnlikefn(t) for t in g["theta"]])
    np.testing.assert_allclose(got, g["lnlike"], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("tag,photscale", [("scaled", True), ("dist", False)])
def test_g4_lnlike_joint(golden, tag, photscale):
    g = golden("g4_lnlike_joint_" + tag)
    cfg = synth.CONFIGS["small"]
    net = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=64, seed=0)
    phot = synth.make_phot_nets()
    phot["hiav"] = golden("g5_sed")["hiav"]
    obs_phot = {f: (m, e) for f, m, e in zip(phot["filters"], g["obs_mag"], g["obs_magerr"])}
    L = O.OracleLikelihood(net, g["obs_wave"], g["obs_flux"], g["obs_eflux"], [str(s) for s in g["fitpars_i"]],
                           phot=phot, obs_phot=obs_phot, photscale=photscale)
    got = np.array([L.lnlikefn(t) for t in g["theta"]])
    assert (g["theta"][:, list(g["fitpars_i"]).index("Av")] >= 5.0).any()
    np.testing.assert_allclose(got, g["lnlike"], rtol=1e-12, atol=1e-9)


def test_g5_sed(golden):
    g = golden("g5_sed")
    phot = synth.make_phot_nets()
    phot["hiav"] = g["hiav"]
    for p, md, ms, bc in zip(g["pars"], g["mags_dist"], g["mags_scaled"], g["bc"]):
        np.testing.assert_allclose(O.fastann_forward(phot, [10.0 ** p[0], p[1], p[2], p[3], p[4], p[5]]), bc, atol=1e-12)
        np.testing.assert_allclose(O.sed_mags(phot, p[0], p[1], p[2], p[3], av=p[4], rv=p[5], logl=p[6], dist=p[7]), md, atol=1e-11)
        np.testing.assert_allclose(O.sed_mags(phot, p[0], p[1], p[2], p[3], av=p[4], rv=p[5], logA=p[8]), ms, atol=1e-11)


def test_g7_polycalc(golden):
    g = golden("g7_misc")
    for c, ref in zip(g["coefs"], g["poly"]):
        np.testing.assert_allclose(O.polycalc(c, g["wave"]), ref, atol=1e-14)


def test_g8_continuum(golden):
    """Continuum network branch of getspec (ystpred.py:191-209) and predictcont, two continuum grids."""
    g = golden("g8_continuum")
    net = synth.make_yst_net(npix=1024, H=64, seed=5, line_depth=0.3)
    for tag, (lo, hi, npc) in {"full": (5140.0, 5190.0, 600), "short": (5140.0, 5170.0, 333)}.items():
        cnet = synth.make_cont_net(npix=npc, lam_lo=lo, lam_hi=hi)
        for i, l in enumerate(g["labels"]):
            np.testing.assert_allclose(O.ann_forward(cnet, list(l)), g["cont_" + tag][i], rtol=1e-12)
            kw = dict(Teff=l[0], logg=l[1], feh=l[2], afe=l[3], cnet=cnet)
            with np.errstate(all="ignore"):
                nat = O.getspec(net, **kw)[1]
            assert np.array_equal(np.isnan(nat), np.isnan(g["native_" + tag][i]))
            assert np.isnan(nat).any() == (tag == "short")
            np.testing.assert_allclose(nat, g["native_" + tag][i], rtol=1e-12, equal_nan=True)
            for j, (vrad, vrot, R) in enumerate(g["rows"]):
                with np.errstate(all="ignore"):
                    f = O.getspec(net, rad_vel=vrad, rot_vel=vrot, vmic=np.nan, inst_R=2.355 * R, outwave=g["obs"], **kw)[1]
                ref = g["final_" + tag][i, j]
                assert np.array_equal(np.isnan(f), np.isnan(ref))
                np.testing.assert_allclose(f, ref, rtol=1e-10, atol=1e-12, equal_nan=True)


def test_g9_lsf(golden):
    """LSF-vector inst_R (ystpred.py:248-269, smoothing.py:125-151, 482-586)."""
    g = golden("g9_lsf")
    net = synth.make_yst_net(npix=1024, H=64, seed=5, line_depth=0.3)
    for a, l in enumerate(g["labels"]):
        kw = dict(Teff=l[0], logg=l[1], feh=l[2], afe=l[3])
        for b, lsf in enumerate(g["lsfs"]):
            for c, (vrad, vrot) in enumerate(g["rows"]):
                with np.errstate(all="ignore"):
                    f = O.getspec(net, rad_vel=vrad, rot_vel=vrot, vmic=np.nan, inst_R=lsf, outwave=g["obs"], **kw)[1]
                assert not np.isnan(f).any()
                np.testing.assert_allclose(f, g["final"][a, b, c], rtol=1e-10, atol=1e-12)
    with np.errstate(all="ignore"):
        w, f = O.getspec(net, rad_vel=8.0, rot_vel=3.0, inst_R=g["lsf_native"], Teff=5300.0, logg=4.1, feh=-0.3, afe=0.15)
    np.testing.assert_allclose(w, g["native_wave"], rtol=1e-15)
    np.testing.assert_allclose(f, g["native"], rtol=1e-10, atol=1e-12)


def test_g11_native_grid(golden):
    """outwave=None: spectrum on the Doppler-shifted model grid, with and without the instrumental stage."""
    g = golden("g11_native")
    net = synth.make_yst_net(npix=1024, H=64, seed=0, line_depth=0.3)
    base, coef = list(g["base"]), list(g["coef"])
    for i, (vrad, vrot, R) in enumerate(g["rows"]):
        with np.errstate(all="ignore"):
            w, f = O.genspec(net, base + [vrad, vrot, np.nan, float(R)], outwave=None)
            w2, f2 = O.genspec(net, base + [vrad, vrot, np.nan, float(R)] + coef, outwave=None, modpoly=True)
        np.testing.assert_allclose(w, g["wave"][i], rtol=1e-15)
        for got, ref in ((f, g["plain"][i]), (f2, g["poly"][i])):
            assert np.array_equal(np.isnan(got), np.isnan(ref))
            ok = ~np.isnan(ref)
            np.testing.assert_allclose(got[ok], ref[ok], rtol=0, atol=1e-12)
    assert np.isnan(g["plain"][0]).sum() == 2          # the end pixels fall outside the resampled grid


def test_g12_long_rows(golden):
    """The oracle on the sizes the build used to refuse (LSF vector on 20 000 pixels, a 9 001-pixel continuum net, a
    continuum net stored in kK) and on the tiny-rotation corner at 32 768 pixels -- vectors from the reference."""
    g = golden("g12_long_rows")
    net = synth.make_yst_net(npix=20000, H=16, seed=41, line_depth=0.3)
    lab = g["lsf_label"]
    for b, lsf in enumerate(g["lsf_lsfs"]):
        for c, (vrad, vrot) in enumerate(g["lsf_rows"]):
            with np.errstate(all="ignore"):
                f = O.getspec(net, Teff=lab[0], logg=lab[1], feh=lab[2], afe=lab[3], rad_vel=vrad, rot_vel=vrot, vmic=np.nan,
                              inst_R=lsf, outwave=g["lsf_obs"])[1]
            np.testing.assert_allclose(f, g["lsf_final"][b, c], rtol=1e-11, atol=1e-12)
    snet = synth.make_yst_net(npix=1024, H=64, seed=5, line_depth=0.3)
    for tag, npc, kk in (("long", 9001, False), ("kk", 600, True)):
        cnet = synth.make_cont_net(npix=npc, lam_lo=5140.0, lam_hi=5190.0)
        if kk:                                            # the reference uses the continuum net as stored (ystpred.py:81-85)
            cnet["x_min"][0] /= 1000.0
            cnet["x_max"][0] /= 1000.0
        for i, l in enumerate(g["cont_labels"]):
            np.testing.assert_allclose(O.ann_forward(cnet, list(l)), g["cont_" + tag][i], rtol=1e-12)
            for j, (vrad, vrot, R) in enumerate(g["cont_rows"]):
                with np.errstate(all="ignore"):
                    f = O.getspec(snet, Teff=l[0], logg=l[1], feh=l[2], afe=l[3], rad_vel=vrad, rot_vel=vrot, vmic=np.nan,
                                  inst_R=2.355 * R, outwave=g["cont_obs"], cnet=cnet)[1]
                ref = g["final_" + tag][i, j]
                assert np.array_equal(np.isnan(f), np.isnan(ref))
                np.testing.assert_allclose(f, ref, rtol=1e-11, atol=1e-12, equal_nan=True)
    rnet = synth.make_yst_net(npix=32768, H=16, seed=43, line_depth=0.3)
    rl = g["rot_label"]
    with np.errstate(all="ignore"):
        f = O.getspec(rnet, Teff=rl[0], logg=rl[1], feh=rl[2], afe=rl[3], rot_vel=float(g["rot_values"][0]))[1]
    np.testing.assert_allclose(f, g["rot_after"][0], rtol=1e-12, atol=1e-13)


G13_CALLS = {
    "vel_direct": dict(resolution=8.0, smoothtype='vel', fftsmooth=False),
    "vel_direct_inres": dict(resolution=8.0, smoothtype='vel', fftsmooth=False, inres=3.0),
    "vel_direct_nsig": dict(resolution=8.0, smoothtype='vel', fftsmooth=False, nsigma=-1),
    "R_direct": dict(resolution=30000.0, smoothtype='R', fftsmooth=False, inres=90000.0),
    "R_direct_native": dict(resolution=30000.0, smoothtype='R', fftsmooth=False, native=True),
    "lambda_direct": dict(resolution=0.2, smoothtype='lambda', fftsmooth=False),
    "lambda_direct_inres": dict(resolution=0.2, smoothtype='lambda', fftsmooth=False, inres=0.08),
    "lambda_direct_invel": dict(resolution=0.2, smoothtype='lambda', fftsmooth=False, inres=60000.0, in_vel=True),
    "lambda_fft": dict(resolution=0.2, smoothtype='lambda', fftsmooth=True),
    "lambda_fft_inres": dict(resolution=0.2, smoothtype='lambda', fftsmooth=True, inres=0.08),
    "lambda_fft_native": dict(resolution=0.2, smoothtype='lambda', fftsmooth=True, native=True),
    "lsf_direct": dict(resolution="lsf_on_wave", smoothtype='lsf', fftsmooth=False),
    "lsf_direct_none": dict(resolution=None, smoothtype='lsf', fftsmooth=False),
}


def g13_call(fn, g, key):
    kw = dict(G13_CALLS[key])
    res = kw.pop("resolution")
    res = g["lsf_on_wave"] if isinstance(res, str) else res
    ow = None if kw.pop("native", False) else g["outwave"]
    return fn(g["wave"], g["spec"], res, outwave=ow, **kw)


def test_g13_smoothspec_branches_off_the_path(golden):
    """smooth_vel / smooth_wave / smooth_lsf / smooth_wave_fft as smoothspec reaches them, against the reference."""
    g = golden("g13_smoothspec")
    for key in G13_CALLS:
        got = g13_call(O.smoothspec_offpath, g, key)
        ref = g[key]
        assert np.array_equal(np.isnan(got), np.isnan(ref)), key
        np.testing.assert_allclose(got, ref, rtol=1e-11, atol=1e-12, equal_nan=True, err_msg=key)


def g15_calls(g):
    """The nine calls of gen_golden.g15_smoothspec_fft_corners: key -> keyword arguments of smoothspec(wave, spec, resolution, ...)."""
    ow, lsf = g["outwave"], g["lsf_on_wave"]
    lim = dict(min_wave_smooth=float(g["limits"][0]), max_wave_smooth=float(g["limits"][1]))
    return {"vsini_out": dict(resolution=12.0, outwave=ow, smoothtype='vsini'),
            "vsini_out_inres": dict(resolution=12.0, outwave=ow, smoothtype='vsini', inres=5.0),
            "vsini_native_inres": dict(resolution=12.0, smoothtype='vsini', inres=5.0),
            "vsini_limits": dict(resolution=12.0, smoothtype='vsini', **lim),
            "vel_limits": dict(resolution=8.0, smoothtype='vel', **lim),
            "vel_limits_inres": dict(resolution=8.0, smoothtype='vel', inres=3.0, **lim),
            "R_limits": dict(resolution=30000.0, smoothtype='R', inres=90000.0, **lim),
            "lsf_out": dict(resolution=lsf, outwave=ow, smoothtype='lsf'),
            "lsf_limits": dict(resolution=lsf, smoothtype='lsf', **lim)}


def test_g15_smoothspec_fft_branches_in_general(golden):
    """smoothspec's FFT branches with the argument combinations off the sampler's path ('vsini' onto another grid / with inres,
    min / max_wave_smooth, an LSF vector on the input grid with another output grid): the restatement against the reference."""
    g = golden("g15_smoothspec_fft")
    n_nan = 0
    for key, kw in g15_calls(g).items():
        with np.errstate(all="ignore"):
            got = O.smoothspec_fft(g["wave"], g["spec"], **kw)
        assert np.array_equal(np.isnan(got), np.isnan(g[key])), key
        np.testing.assert_allclose(got, g[key], rtol=0, atol=1e-12, equal_nan=True)
        n_nan += int(np.isnan(g[key]).sum())
    assert n_nan > 500                                            # the limited calls really leave NaN outside the kept range
