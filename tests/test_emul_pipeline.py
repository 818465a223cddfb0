"""The per-candidate pipeline source that runs on the GPU (thepayne_amd/csrc/post_core.hpp,
post_seq.hpp), executed on the host by tests/emul/cpu_emul.cpp, against the reference's
golden vectors and the oracle.  This is the CPU-side check of the kernel LOGIC (and the
target of sanitizer builds); the GPU tests check the kernel itself."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import oracle as O
from thepayne_amd import synth
from helpers import lnl_tol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dp, fp, ip = (ctypes.POINTER(t) for t in (ctypes.c_double, ctypes.c_float, ctypes.c_int))


@pytest.fixture(scope="module")
def emul():
    import importlib.util
    spec = importlib.util.spec_from_file_location("graft_entry", os.path.join(ROOT, "__graft_entry__.py"))
    ge = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ge)
    lib = ctypes.CDLL(ge.build_emul())

    def P(a, t):
        return a.ctypes.data_as(t) if a is not None else None

    def run(net, obs, flux, eflux, theta8, stage, npoly=0, pcs=None, factor=2.355, general=0, nthreads=512, prep=1):
        theta8 = np.atleast_2d(theta8)
        B, npix = len(theta8), len(net["wavelength"])
        raw = np.array([O.yst_forward(net, t[:4]) - 1.0 for t in theta8]).astype(np.float32)
        th = np.ascontiguousarray(theta8 if pcs is None else np.hstack([theta8, pcs]), dtype=np.float64)
        nout = npix if stage in (0, 1) else len(obs)
        out = np.zeros((B, nout), np.float32)
        chi2, info = np.zeros(B), np.zeros((B, 3), np.int32)
        rc = lib.payne_emul_post(P(net["wavelength"], dp), npix, ctypes.c_double(net["resolution"]), P(obs, dp),
                                 P(flux, dp), P(eflux, dp), len(obs), npoly, P(th, dp), th.shape[1], B,
                                 ctypes.c_double(factor), P(raw, fp), stage, P(out, fp), nout, P(chi2, dp), P(info, ip),
                                 nthreads, general, prep)
        assert rc == 0, rc        # -77/-78: the setup-time mask probe disagrees with the full mask count
        return out, chi2, info
    run.fast_windows = lib.payne_emul_fast_windows
    run.lib = lib
    return run


def _th8(th7):
    th7 = np.atleast_2d(th7)
    return np.column_stack([th7[:, :6], np.full(len(th7), np.nan), th7[:, 6]])


@pytest.mark.parametrize("general,prep", [(0, 1), (1, 1), (0, 0), (1, 0), (4, 1), (4, 0)])   # 4: rows in the frequency domain
def test_c2_lnlike_matches_reference_golden(emul, golden, general, prep):
    g = golden("g4_lnlike_c2")
    cfg = synth.CONFIGS["C2"]
    net = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], seed=0)
    idx = np.arange(0, 512, 16)
    before = emul.fast_windows()
    _, chi2, info = emul(net, g["obs_wave"], g["obs_flux"], g["obs_eflux"], _th8(g["theta"][idx]), -1, general=general,
                         prep=prep)
    # mask counts without the full scan: always with the ahead-of-kernel record, else on geometric grids (setup probe)
    assert emul.fast_windows() - before == (len(idx) if (prep or general in (0, 4)) else 0)
    ref = g["lnlike"][idx]
    assert np.all(np.abs(-0.5 * chi2 - ref) <= lnl_tol(ref))
    assert np.abs(-0.5 * chi2 - ref).max() < 1e-3        # in practice ~1e-4
    assert (info[:, 2] == 4096).all() and (info[:, 1] > 3800).all()


@pytest.mark.parametrize("general", [0, 1, 4])
def test_getspec_grid_matches_reference_golden(emul, golden, general):
    g = golden("g2_getspec")
    net = synth.make_yst_net(npix=1024, H=64, seed=5, line_depth=0.3)
    lab, rows = g["labels"], g["theta_rows"]
    th8 = np.array([[lab[0], lab[1], lab[2], lab[3], r[0], r[1], np.nan, r[2]] for r in rows])
    out, _, info = emul(net, g["obs_wave"], None, None, th8, 2, general=general)
    assert np.array_equal(np.isnan(out), np.isnan(g["final"]))
    assert np.nanmax(np.abs(out - g["final"])) <= 1e-6
    ok = np.isfinite(rows[:, 2])
    assert np.array_equal(info[ok, 0], g["mask_first_count"][ok, 0])
    assert np.array_equal(info[ok, 1], g["mask_first_count"][ok, 1])
    for v, ref in zip(g["vrot_values"], g["after_rot"]):
        t = th8[:1].copy(); t[0, 5] = v
        o, _, _ = emul(net, g["obs_wave"], None, None, t, 1, general=general)
        assert np.abs(o[0] - ref).max() <= 1e-6, v


def test_modpoly_matches_reference_golden(emul, golden):
    g = golden("g4_lnlike_modpoly")
    cfg = synth.CONFIGS["small"]
    net = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=64, seed=0)
    th = g["theta"]
    _, chi2, _ = emul(net, g["obs_wave"], g["obs_flux"], g["obs_eflux"], _th8(th[:, :7]), -1, npoly=3, pcs=th[:, 7:10])
    assert np.all(np.abs(-0.5 * chi2 - g["lnlike"]) <= lnl_tol(g["lnlike"]))


@pytest.mark.parametrize("npix,nobs,nthreads", [(300, 250, 64), (777, 600, 256), (2048, 1800, 128), (5000, 4000, 256), (40000, 30000, 512)])
def test_ragged_sizes_vs_oracle(emul, npix, nobs, nthreads):
    """npix not a power of two (the vsini grid is then a genuine resampling), few threads."""
    net = synth.make_yst_net(npix=npix, H=24, seed=21, line_depth=0.2)
    span = net["wavelength"][-1] - net["wavelength"][0]
    obs = synth.obs_grid(net["wavelength"], nobs, inset=0.06 * span)
    th7 = synth.draw_candidates(6 if npix < 20000 else 2, seed=npix)
    ref = np.array([O.genspec(net, list(_th8(t)[0]), outwave=obs)[1] for t in th7])
    out, _, _ = emul(net, obs, None, None, _th8(th7), 2, nthreads=nthreads)
    assert np.array_equal(np.isnan(out), np.isnan(ref))
    assert np.nanmax(np.abs(out - ref)) <= 1e-6


def test_nonuniform_ann_grid_vs_oracle(emul):
    """A non-geometric ANN wavelength grid takes the search path automatically."""
    net = synth.make_yst_net(npix=1500, H=24, seed=22, line_depth=0.2)
    rng = np.random.default_rng(0)
    w = net["wavelength"]
    net["wavelength"] = w + 0.3 * np.diff(w).mean() * rng.uniform(-1, 1, len(w))
    obs = synth.obs_grid(net["wavelength"], 1200, inset=2.0)
    th7 = synth.draw_candidates(6, seed=3)
    ref = np.array([O.genspec(net, list(_th8(t)[0]), outwave=obs)[1] for t in th7])
    out, _, _ = emul(net, obs, None, None, _th8(th7), 2)
    assert np.nanmax(np.abs(out - ref)) <= 1e-6


def test_emulator_under_address_and_ub_sanitizers(tmp_path):
    """Sanitizers run on the CPU build only (no GPU ASan on this pool)."""
    exe = tmp_path / "emul_san"
    main = tmp_path / "main.cpp"
    main.write_text(r'''
#include <cmath>
#include <cstdio>
#include <vector>
extern "C" int payne_emul_post(const double*, int, double, const double*, const double*, const double*, int, int,
    const double*, int, int, double, const float*, int, float*, int, double*, int*, int, int);
int main() {
  const int npix = 700, nobs = 500, B = 3;
  std::vector<double> w(npix), ow(nobs), fl(nobs, 1.0), ef(nobs, 0.01), th(B * 8);
  for (int i = 0; i < npix; ++i) w[i] = 5150.0 * std::pow(1.0 + 1.0 / (3 * 32000 * 2.3548), i);
  for (int i = 0; i < nobs; ++i) ow[i] = w[20] + (w[npix - 21] - w[20]) * i / (nobs - 1.0);
  std::vector<float> raw(B * npix);
  for (int i = 0; i < B * npix; ++i) raw[i] = 0.05f * std::sin(0.05f * i);
  for (int b = 0; b < B; ++b) { double* t = &th[b * 8]; t[0]=5770; t[1]=4.4; t[2]=0; t[3]=0; t[4]=10.0*b; t[5]=b; t[6]=NAN; t[7]=25000+2000*b; }
  std::vector<float> out(B * nobs); std::vector<double> chi(B); std::vector<int> info(3 * B);
  int rc = payne_emul_post(w.data(), npix, 32000 * 2.3548, ow.data(), fl.data(), ef.data(), nobs, 0, th.data(), 8, B, 2.355,
                           raw.data(), 2, out.data(), nobs, chi.data(), info.data(), 256, 0);
  std::printf("rc=%d chi=%g\n", rc, chi[1]);
  return rc;
}''')
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           str(main), os.path.join(ROOT, "tests", "emul", "cpu_emul.cpp"), "-o", str(exe)]
    subprocess.run(cmd, check=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "rc=0" in res.stdout


@pytest.mark.parametrize("general", [0, 1])
def test_output_on_the_model_grid_itself(emul, general):
    """getspec(outwave=None, inst_R=...): the output grid IS the Doppler-shifted model grid, so its end pixels
    coincide with the ends of the resampled grid.  The reference keeps or drops them by one rounding of
    exp(log(.)) (the host applies that verdict, predict/_spec.py native_grid_edges); the kernel must treat them
    as inside and interpolate at position 0 / n-1 -- not wrap to the other end of the convolved buffer."""
    net = synth.make_yst_net(npix=700, H=32, seed=21, line_depth=0.3)
    wave = net["wavelength"]
    for rv, vrot, R in ((-159.20149211002166, 0.0, 33704.72291801852), (0.0, 3.0, 28000.0), (12.0, 4.0, 30000.0),
                        (250.0, 0.0, 41000.0), (-33.3, 8.0, 22000.0)):
        grid = np.ascontiguousarray(wave * (1.0 + rv / 299792.458) if rv != 0.0 else wave)
        th8 = np.array([[5600.0, 4.3, -0.2, 0.1, rv, vrot, np.nan, R]])
        out, _, _ = emul(net, grid, None, None, th8, 2, factor=1.0, general=general)
        with np.errstate(all="ignore"):
            _, ref = O.getspec(net, Teff=5600.0, logg=4.3, feh=-0.2, afe=0.1, rad_vel=rv, rot_vel=vrot, inst_R=R, outwave=None)
        assert not np.isnan(out[0]).any()
        ok = ~np.isnan(ref)
        assert ok.sum() >= len(ref) - 2 and np.abs(out[0][ok] - ref[ok]).max() <= 1e-6
        # the end pixels are the convolved buffer's own ends, not the wrapped-around other end
        assert abs(out[0][0] - out[0][1]) < 5e-3 and abs(out[0][-1] - out[0][-2]) < 5e-3


@pytest.mark.parametrize("npix,nobs", [(20000, 3000), (40000, 5000), (70000, 4000)])
def test_four_step_transform_of_the_big_kernel(emul, npix, nobs):
    """Spectra larger than LDS (payne_post_big_kernel): the four-step transform (512-point sub-transforms in an
    LDS tile, M = 512 x {32, 64, 128}) against the runtime-geometry passes and against the oracle."""
    net = synth.make_yst_net(npix=npix, lam0=4000.0, R_fwhm=100000.0, H=8, seed=3, line_depth=0.3)
    obs = synth.obs_grid(net["wavelength"], nobs, inset=0.0005, relative=True)
    th8 = np.array([[5600.0, 4.3, -0.2, 0.1, 12.0, 6.0, np.nan, 60000.0],
                    [6200.0, 3.8, -0.8, 0.2, -35.0, 0.0, np.nan, 80000.0]])
    plain, _, info0 = emul(net, obs, None, None, th8, 2, factor=1.0, general=1, nthreads=512)
    tiled, _, info1 = emul(net, obs, None, None, th8, 2, factor=1.0, general=2, nthreads=512)
    assert np.array_equal(info0, info1) and np.array_equal(np.isnan(plain), np.isnan(tiled))
    assert 0.0 < np.nanmax(np.abs(plain - tiled)) < 5e-7          # same arithmetic up to the order of the twiddles
    for i, t in enumerate(th8):
        with np.errstate(all="ignore"):
            _, ref = O.getspec(net, Teff=t[0], logg=t[1], feh=t[2], afe=t[3], rad_vel=t[4], rot_vel=t[5], inst_R=t[7], outwave=obs)
        assert np.array_equal(np.isnan(tiled[i]), np.isnan(ref))
        assert np.nanmax(np.abs(tiled[i] - ref)) <= 1e-6


def test_four_step_transform_on_a_geometric_grid(emul):
    """The C5 shape (65 536 pixels on a geometric grid: the vsini resampling is the identity) through the four-step transform of
    the global-workspace kernel (what payne_post_big_kernel runs under PAYNE_V_BIG_WORKSPACE); against the oracle."""
    npix, nobs = 65536, 3000
    net = synth.make_yst_net(npix=npix, lam0=4000.0, R_fwhm=100000.0, H=8, seed=3, line_depth=0.3)
    obs = synth.obs_grid(net["wavelength"], nobs, inset=0.0005, relative=True)
    th8 = np.array([[5600.0, 4.3, -0.2, 0.1, 12.0, 6.0, np.nan, 60000.0]])
    got, _, _ = emul(net, obs, None, None, th8, 2, factor=1.0, general=3, nthreads=512)
    with np.errstate(all="ignore"):
        _, ref = O.getspec(net, Teff=5600.0, logg=4.3, feh=-0.2, afe=0.1, rad_vel=12.0, rot_vel=6.0, inst_R=60000.0, outwave=obs)
    assert np.array_equal(np.isnan(got[0]), np.isnan(ref)) and np.nanmax(np.abs(got[0] - ref)) <= 1e-6


@pytest.mark.parametrize("n,layout", [(1024, 0), (4096, 0), (65536, 1), (32768, 2)])
def test_output_layer_restated_for_rows_in_the_frequency_domain(emul, n, layout):
    """host_tables.hpp freq_rows (what payne_ctx_create uploads beside the pixel weights): every hidden unit's pixel vector and the
    bias - 1 as the half-length complex transform Z = FFT_M(v[2m] + i v[2m+1]) (numpy's sign convention), bins placed where the post
    kernels read them: pair layout (slot j = Z[j], Z[M-j]; slot 0 = Z[0], Z[M/2]) for the LDS kernel, the register order of the
    on-chip stages for 65 536 / 32 768 points.  Linearity is then the whole argument: W_z a + b_z is the transform of W a + b - 1."""
    rng = np.random.default_rng(n)
    K, M = 5, n // 2
    W = rng.normal(0, 0.05, (n, K)).astype(np.float32)
    b = (1.0 + rng.normal(0, 0.02, n)).astype(np.float32)
    Wz, bz = np.zeros((n, K), np.float32), np.zeros(n, np.float32)
    assert emul.lib.payne_emul_freq_rows(W.ctypes.data_as(fp), b.ctypes.data_as(fp), ctypes.c_double(-1.0), n, K, layout,
                                         Wz.ctypes.data_as(fp), bz.ctypes.data_as(fp)) == 0
    # the bin k of every slot, and its partner
    if layout == 0:
        k = np.arange(M // 2)
    elif layout == 1:                                            # virtual thread vt = 32 h + l combines k = h + 32 l + 1024 r
        r, vt = np.divmod(np.arange(16 * 1024), 1024)
        k = (vt >> 5) + 32 * (vt & 31) + 1024 * r
    else:                                                        # slot r * 512 + 16 h + k2a holds k = h + 32 k2a + 512 r
        r, q = np.divmod(np.arange(16 * 512), 512)
        h, k2a = np.divmod(q, 16)
        k = h + 32 * k2a + 512 * r
    kb = np.where(k == 0, M // 2, M - k)
    seen = np.zeros(M, int)
    np.add.at(seen, k, 1); np.add.at(seen, kb, 1)
    assert (seen == 1).all()                                     # every bin of the transform exactly once

    def place(v):
        Z = np.fft.fft(v[0::2].astype(np.float64) + 1j * v[1::2].astype(np.float64))
        out = np.empty(n)
        out[0::4], out[1::4], out[2::4], out[3::4] = Z[k].real, Z[k].imag, Z[kb].real, Z[kb].imag
        return out
    for col in range(K):
        ref = place(W[:, col])
        assert np.abs(Wz[:, col] - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())
    ref = place(b.astype(np.float64) - 1.0)
    assert np.abs(bz - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())


def test_mask_counts_by_arithmetic_are_the_tables_counts():
    """prep_candidate (the record the hidden-layer launch writes for every candidate) finds the instrumental stage's mask counts on
    geometric grids by arithmetic -- (ln obs_min + ln(1 - 20/R) - ln(1 + rv/c) - ln lam_0) / dln, two short series, no table value: the
    dependent memory round trip of the launch's last workgroups is gone -- wherever the position is more than 1e-5 pixel from a
    boundary, and from the table otherwise.  Both must be THE counts of lam_i (1 + rv/c) <= wl, < wh over every pixel: 60 000 random
    candidates on two grids, plus sweeps of R and rv fine enough to walk positions across pixel boundaries (the table path's cases)."""
    import ctypes as C
    import __graft_entry__ as G
    lib = C.CDLL(G.build_emul())
    fn = lib.payne_emul_prep_counts
    fn.restype = C.c_int
    dp = C.POINTER(C.c_double)
    rng = np.random.default_rng(11)
    for cfg_name, nobs in (("C2", 3600), ("small", None)):
        cfg = synth.CONFIGS[cfg_name]
        net = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=8, seed=0)
        wave = np.ascontiguousarray(net["wavelength"], dtype=np.float64)
        obs = np.ascontiguousarray(synth.obs_grid(wave, nobs or cfg["nobs"]), dtype=np.float64)
        n = 30000
        rv = np.concatenate([rng.uniform(-700.0, 700.0, n - 6000), np.linspace(9.0, 9.02, 3000), np.zeros(3000)])
        rin = np.concatenate([rng.uniform(0.4, 0.95, n - 6000) * cfg["R"], np.full(3000, 0.8 * cfg["R"]),
                              np.linspace(0.5, 0.500004, 3000) * cfg["R"]])
        # ... and candidates placed ON pixel boundaries: rv chosen so that the lower limit's position is an integer +- a few 1e-8 pixel
        # (the table decides these; the arithmetic must know that it cannot)
        lnw = np.log(wave)
        dln = (lnw[-1] - lnw[0]) / (len(wave) - 1)
        rs_b = 0.8 * cfg["R"] * 2.355
        ln_wl = np.log(obs.min()) + np.log1p(-20.0 / rs_b)
        k0 = int((ln_wl - lnw[0]) / dln)
        kk = k0 - np.arange(1, 201)
        dop_b = ln_wl - lnw[0] - kk * dln + rng.uniform(-3e-8, 3e-8, len(kk)) * dln
        rv_b = 299792.458 * np.expm1(dop_b)
        rv = np.concatenate([rv[:-200], rv_b]); rin = np.concatenate([rin[:-200], np.full(200, 0.8 * cfg["R"])])
        rv, rin = np.ascontiguousarray(rv), np.ascontiguousarray(rin)
        n_arith, first_bad = C.c_int(0), C.c_int(-1)
        bad = fn(wave.ctypes.data_as(dp), len(wave), C.c_double(float(net["resolution"])), obs.ctypes.data_as(dp), len(obs),
                 rv.ctypes.data_as(dp), rin.ctypes.data_as(dp), n, C.c_double(2.355), C.byref(n_arith), C.byref(first_bad))
        assert bad == 0, (cfg_name, bad, first_bad.value, rv[first_bad.value], rin[first_bad.value])
        assert 0.99 * n < n_arith.value <= n - 150, (cfg_name, n_arith.value)     # (the boundary candidates, and a few others, took the table)
