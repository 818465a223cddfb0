"""Host-side logic (no GPU): prior transforms / ln-priors and helpers against golden
vectors frozen from the reference; the C-ABI library loads and exports its symbols."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from thepayne_amd import synth
from thepayne_amd.fitting.prior import prior
from thepayne_amd.fitting import fitutils

ALL_PARS = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Vmic', 'Inst_R',
            'log(R)', 'Dist', 'log(A)', 'Av', 'Rv', 'CarbonScale']
SPEC_PARS = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Inst_R']
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fitpars_for(on, npoly=0):
    names = list(ALL_PARS) + ['pc_%d' % i for i in range(npoly)]
    return [names, {p: (p in on or p.startswith('pc_')) for p in names}]


KINDS = {
    "uniform": {'pv_uniform': [4000.0, 8000.0]}, "uniform_rev": {'pv_uniform': [8000.0, 4000.0]},
    "gaussian": {'pv_gaussian': [5770.0, 100.0]}, "tgaussian": {'pv_tgaussian': [25000.0, 37000.0, 28800.0, 1000.0]},
    "exp": {'pv_exp': [0.0, 2.0]}, "texp": {'pv_texp': [0.0, 10.0, 2.0]}, "default": None,
}


def test_priortrans_every_kind(golden):
    g = golden("g6_prior")
    u = g["u"]
    fitargs = {'fixedpars': {}}
    for key in g.files:
        m = re.match(r"(spec|phot)_(.+)_(uniform_rev|uniform|gaussian|tgaussian|exp|texp|default)$", key)
        if not m:
            continue
        which, par, kname = m.groups()
        pd = {} if KINDS[kname] is None else {par: KINDS[kname]}
        rb = [True, False, False, False, False] if which == "spec" else [False, True, False, True, False]
        P = prior(fitargs, pd, fitpars_for([par]), rb)
        with np.errstate(all="ignore"):
            got = np.array([P.priortrans([ui])[0] for ui in u], dtype=float)
            got_b = P.priortrans_batch(u[:, None])[:, 0]
        np.testing.assert_allclose(got, g[key], rtol=1e-13, atol=0, err_msg=key)
        np.testing.assert_allclose(got_b, g[key], rtol=1e-13, atol=0, err_msg=key + " (batch)")


def test_priortrans_photonly_and_blaze(golden):
    g = golden("g6_prior")
    fitargs = {'fixedpars': {}}
    P = prior(fitargs, {'Teff': {'pv_gaussian': [5770.0, 200.0]}, 'Av': {'pv_uniform': [0.0, 1.0]}},
              fitpars_for(['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'log(A)', 'Av']), [False, True, False, True, False])
    np.testing.assert_allclose(P.priortrans_batch(g["photonly_u"]), g["photonly_theta"], rtol=1e-13)
    pd = synth.demo_priordict()
    pd['blaze_coeff'] = [[0.0, 0.5], [0.1, 0.2], [-0.05, 0.1], [0.0, 0.01]]
    P = prior(fitargs, pd, fitpars_for(SPEC_PARS, npoly=4), [True, False, True, False, False])
    np.testing.assert_allclose(P.priortrans_batch(g["blaze_u"]), g["blaze_theta"], rtol=1e-13)
    np.testing.assert_allclose(np.array([P.priortrans(u) for u in g["blaze_u"]]), g["blaze_theta"], rtol=1e-13)


def test_lnprior_additional(golden):
    g = golden("g6_prior")
    fitargs = {'fixedpars': {}}
    pd = synth.demo_priordict()
    pd['Teff']['gaussian'] = [5770.0, 50.0]
    pd['[Fe/H]']['uniform'] = [-0.05, 0.08]
    P = prior(fitargs, pd, fitpars_for(SPEC_PARS), [True, False, False, False, False])
    np.testing.assert_allclose(P.priortrans_batch(g["lnprior_spec_u"]), g["lnprior_spec_theta"], rtol=1e-13)
    got = P.lnprior_batch(g["lnprior_spec_theta"])
    assert np.array_equal(np.isinf(got), np.isinf(g["lnprior_spec"])) and np.isinf(got).any()
    ok = np.isfinite(got)
    np.testing.assert_allclose(got[ok], g["lnprior_spec"][ok], rtol=1e-13)
    pd = synth.demo_priordict()
    pd['Av'] = {'pv_uniform': [0.0, 1.0], 'gaussian': [0.1, 0.05], 'uniform': [0.02, 0.9]}
    pd['log(A)'] = {'pv_uniform': [-3.0, 7.0]}
    P = prior(fitargs, pd, fitpars_for(SPEC_PARS + ['log(A)', 'Av']), [True, True, False, True, False])
    got = P.lnprior_batch(g["lnprior_joint_theta"])
    assert np.array_equal(np.isinf(got), np.isinf(g["lnprior_joint"]))
    ok = np.isfinite(got)
    np.testing.assert_allclose(got[ok], g["lnprior_joint"][ok], rtol=1e-13)
    # demo priors: no additional priors -> exactly 0.0 (prior.py:364-365)
    P = prior(fitargs, synth.demo_priordict(), fitpars_for(SPEC_PARS), [True, False, False, False, False])
    assert P.lnpriorfn([5770.0, 4.4, 0, 0, 10, 3, 28000.0]) == 0.0


def test_advanced_priors(golden):
    """IMF / VROT in lnpriorfn, GAL in the transform of Dist, VTOT / AngDia without effect --
    against values frozen from the reference (oracle/gen_golden.py g10)."""
    g = golden("g10_advpriors")
    fitargs = {'fixedpars': {}}
    names = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Inst_R', 'log(R)', 'Dist', 'Av']
    base = {'Dist': {'pv_uniform': [10.0, 20000.0]}}
    cases = {
        "imf": dict(base, IMF={'IMF_type': 'Kroupa'}),
        "vrot": dict(base, VROT={}),
        "imf_vrot_gauss": dict(base, IMF={'IMF_type': 'Kroupa'}, VROT={}, Vrad={'gaussian': [5.0, 30.0]}),
        "vtot_angdia": dict(base, VTOT={'pmra': 30.0, 'pmdec': -12.0}, AngDia={'gaussian': [0.5, 0.05]}),
    }
    for tag, pd in cases.items():
        P = prior(fitargs, pd, fitpars_for(names), [True, True, False, False, False])
        ref = g["lnp_" + tag]
        for got in (np.array([P.lnpriorfn(list(t)) for t in g["theta"]]), P.lnprior_batch(g["theta"])):
            assert np.array_equal(np.isinf(got), np.isinf(ref)), tag
            ok = np.isfinite(ref)
            np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-12, atol=1e-300, err_msg=tag)
    assert np.isinf(g["lnp_imf"]).any() and (g["lnp_vrot"] < -9).any()       # both tails are exercised
    names_a = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Inst_R', 'log(A)', 'Av']
    P = prior(fitargs, {'VROT': {}}, fitpars_for(names_a), [True, True, False, True, False])
    np.testing.assert_allclose(P.lnprior_batch(g["theta_a"]), g["lnp_vrot_logA"], rtol=1e-12, atol=1e-300)
    for tag, rng_d in (("disk", [10.0, 20000.0]), ("pole", [100.0, 100000.0]), ("nodist", None)):
        pd = {'GAL': {'lb_coords': list(g["gal_lb_" + tag])}}
        if rng_d is not None:
            pd['Dist'] = {'pv_uniform': rng_d}
        P = prior(fitargs, pd, fitpars_for(['Dist']), [False, True, False, False, False])
        one = np.array([P.priortrans([ui])[0] for ui in g["u"]])
        np.testing.assert_allclose(one, g["gal_" + tag], rtol=1e-12, err_msg=tag)
        np.testing.assert_allclose(P.priortrans_batch(g["u"][:, None])[:, 0], g["gal_" + tag], rtol=1e-12)
    # a spectrum-only fit has no log(R): the mass cannot be formed (KeyError in the reference too)
    P = prior(fitargs, {'IMF': {'IMF_type': 'Kroupa'}}, fitpars_for(SPEC_PARS), [True, False, False, False, False])
    with pytest.raises(KeyError):
        P.lnpriorfn([5770.0, 4.4, 0, 0, 10, 3, 28000.0])


def test_fitutils(golden):
    g = golden("g7_misc")
    for c, ref in zip(g["coefs"], g["poly"]):
        np.testing.assert_allclose(fitutils.polycalc(c, g["wave"]), ref, atol=1e-14)
    np.testing.assert_allclose(fitutils.airtovacuum(g["air"]), g["vac"], rtol=1e-14)
    np.testing.assert_allclose(fitutils.vacuumtoair(fitutils.airtovacuum(g["air"])), g["air"], rtol=2e-8)


def test_c_abi_library_loads_and_exports_the_header():
    """Every function include/payne_hip.h declares is exported by the built library
    (no compute calls here: no GPU)."""
    from thepayne_amd.build import build_lib
    from thepayne_amd import _lib
    path = build_lib()
    lib = ctypes.CDLL(path)
    hdr = open(os.path.join(ROOT, "include", "payne_hip.h")).read()
    declared = set(re.findall(r"\b(payne_[a-z_]+)\s*\(", hdr))
    declared -= {"payne_ctx"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    L = _lib.load(path)
    assert L.payne_version() == _lib.ABI_VERSION
    assert L.payne_kernel_name(1) == b"payne_post_kernel"
    # struct layouts agree with the header's field order (sizes on LP64)
    assert ctypes.sizeof(_lib.Layer) == 32 and ctypes.sizeof(_lib.Opts) == 16
    assert ctypes.sizeof(_lib.ModelDesc) == 8 + 8 * 32 + 8 + 8 + 8 + 8 + 8 + 8


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from thepayne_amd.engine import PayneEngine
    from thepayne_amd import nnio
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        PayneEngine(nnio.normalize_spec_net(synth.make_yst_net(npix=256, H=16)))


def _bf16_rne(x):
    """fp32 -> bf16 (round to nearest even) -> fp32, as the device's (__bf16) cast does."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def test_three_bf16_parts_hold_an_fp32_value_exactly():
    """The output layer multiplies operands split as x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)
    (thepayne_amd/csrc/dense_kernels.hpp, split3).  The claim its accuracy rests on: the three parts sum to x EXACTLY
    (24 significant bits = 8 + 8 + 8, with the signs of the remainders absorbing the rounding), and the six partial products
    the kernel keeps differ from the full product by less than 2^-22 |a b|."""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(0, 1, 200000), rng.normal(0, 1e-3, 50000), rng.uniform(-300, 300, 50000),
                        [0.0, 1.0, -1.0, 0.1, 3.0e-30, 1.0e30, 1.0 + 2.0 ** -23, 255.99998]]).astype(np.float32)
    x1 = _bf16_rne(x)
    r1 = (x - x1).astype(np.float32)
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - x1.astype(np.float64))      # the subtraction is exact
    x2 = _bf16_rne(r1)
    r2 = (r1 - x2).astype(np.float32)
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - x2.astype(np.float64))
    x3 = _bf16_rne(r2)
    assert np.array_equal(x3, r2)                                                                    # nothing is left over
    tot = x1.astype(np.float64) + x2.astype(np.float64) + x3.astype(np.float64)
    assert np.array_equal(tot, x.astype(np.float64))
    # the six products kept: a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1
    a, b = x[:100000], x[100000:200000]
    parts = lambda v: (lambda v1: (lambda v2: (v1, v2, _bf16_rne((v - v1 - v2).astype(np.float32))))(_bf16_rne((v - v1).astype(np.float32))))(_bf16_rne(v))
    a1, a2, a3 = [t.astype(np.float64) for t in parts(a)]
    b1, b2, b3 = [t.astype(np.float64) for t in parts(b)]
    kept = a1 * b1 + a1 * b2 + a2 * b1 + a1 * b3 + a2 * b2 + a3 * b1
    full = a.astype(np.float64) * b.astype(np.float64)
    assert np.all(np.abs(kept - full) <= 2.0 ** -22 * np.abs(full) + 1e-300)


def test_turn_kernel_releases_once_and_exchanges_inside_the_wave(tmp_path):
    """`payne_ns_turn_kernel` (the boundary between two proposal queues, one workgroup): ONE system-scope write-back in front of the
    completion word -- a `__threadfence_system()` per wave was 6 us of its 29 --, no scratch memory (1024 threads: 128 registers), and
    the sort network's exchanges at distances below 64 as DPP moves / row swaps in the unrolled 1024-element form (NOTES R5.12e).
    Read off the assembly of the unit that holds it."""
    import re
    import subprocess
    from thepayne_amd import build
    asm = tmp_path / "payne_hip.s"
    cmd = [build._hipcc()] + [f for f in build.HIPCC_FLAGS if f != "-fPIC"] + ["-I", os.path.join(build.ROOT, "include"), "-S", "--cuda-device-only",
                                                                           os.path.join(build.CSRC, "payne_hip.hip"), "-o", str(asm)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    text = asm.read_text()
    m = re.search(r"^_Z20payne_ns_turn_kernel8TurnArgs:.*?s_endpgm", text, re.S | re.M)
    assert m
    body = m.group(0)
    assert body.count("buffer_wbl2") == 1, body.count("buffer_wbl2")
    assert "scratch_" not in body
    assert body.count("v_permlane16_swap") >= 15 and body.count("v_permlane32_swap") >= 12          # 6 resp. 5 stages x 3 dwords
    assert body.count("ds_bpermute") <= 6                                                          # (the generic sizes' loop only)
    m = re.search(r"^_Z22payne_stage_out_kernel\w*:.*?s_endpgm", text, re.S | re.M)
    assert m and m.group(0).count("buffer_wbl2") <= 3, m and m.group(0).count("buffer_wbl2")


def test_likelihood_post_kernels_keep_two_workgroups_per_cu(tmp_path):
    """The likelihood-only post kernels of the LDS-resident sizes must stay at <= 128 vector registers: at 129 the hardware runs ONE
    512-thread workgroup per compute unit instead of two and the C2 step loses 6 us (NOTES.md 3.3d; it happened three times while
    unrelated code moved).  Read off the compiler's own resource summary of the unit that holds them."""
    import re
    import subprocess
    from thepayne_amd import build
    asm = tmp_path / "k_post_lean.s"
    cmd = [build._hipcc()] + [f for f in build.HIPCC_FLAGS if f != "-fPIC"] + ["-I", os.path.join(build.ROOT, "include"), "-S", "--cuda-device-only",
                                                                           os.path.join(build.CSRC, "k_post_lean.hip"), "-o", str(asm)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    text = asm.read_text()
    seen = {}
    for m in re.finditer(r"^(_Z17payne_post_kernelILi(1[012])ELb1ELb1EE\w*):.*?; NumVgprs: (\d+).*?; Occupancy: (\d+)", text, re.S | re.M):
        seen[int(m.group(2))] = (int(m.group(3)), int(m.group(4)))
    assert set(seen) == {10, 11, 12}, seen
    for log2n, (vgprs, occ) in seen.items():
        assert vgprs <= 128 and occ >= 4, (log2n, vgprs, occ)
    # ... and the C2 kernel asks for its row in the SAME memory round trip as for its twiddle table: no wait for vector loads between the
    # table's requests (the first global loads) and the slots' (non-temporal, off a preloaded base).  The compiler's wait-count pass once put
    # `s_waitcnt vmcnt(0)` there -- a phantom hazard with the pixel-row path of the same block -- and it cost every workgroup a round
    # trip for two rounds before anybody read the assembly (NOTES R5.12d).
    m = re.search(r"^_Z17payne_post_kernelILi12ELb1ELb1EE\w*:.*?s_endpgm", text, re.S | re.M)
    body = m.group(0).splitlines()
    first_load = next(i for i, l in enumerate(body) if "global_load_dword" in l)
    first_nt = next(i for i, l in enumerate(body) if re.search(r"global_load_dwordx4 .*s\[\d+:\d+\] nt\s*$", l))   # the slots: a preloaded base
    assert first_load < first_nt < first_load + 200, (first_load, first_nt)
    between = [l for l in body[first_load:first_nt] if "s_waitcnt" in l and "vmcnt" in l]
    assert not between, between
    # ... with the arguments its first requests hang off preloaded into registers (13 or 14 dwords)
    pre = re.search(r"amdhsa_kernel _Z17payne_post_kernelILi12ELb1ELb1EE.*?amdhsa_user_sgpr_kernarg_preload_length (\d+)", text, re.S)
    assert pre and int(pre.group(1)) >= 13, pre and pre.group(1)


def test_on_chip_stages_of_the_usual_path_never_touch_scratch_memory(tmp_path):
    """The 65 536-point likelihood kernel calls its convolution stages as functions that use the whole register file.  A function
    that itself CALLS something (the rotation stage's exact re-evaluation of taper bins beyond the table) keeps its live registers in
    callee-saved ones and saves / restores those through scratch memory on every call: 63 dwords a thread each way, 258 KB a candidate,
    a fifth of the bytes a C5 launch moved for two rounds.  The usual path's stages (rotation without the cold call: chip_conv<true,
    false>; instrumental stage onto the observed grid: chip_conv_obs) must not touch scratch at all; the kernel itself may spill a
    handful of registers around its calls, not more."""
    import re
    import subprocess
    from thepayne_amd import build
    asm = tmp_path / "k_post_big.s"
    cmd = [build._hipcc()] + [f for f in build.HIPCC_FLAGS if f != "-fPIC"] + ["-I", os.path.join(build.ROOT, "include"), "-S", "--cuda-device-only",
                                                                           os.path.join(build.CSRC, "k_post_big.hip"), "-o", str(asm)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = asm.read_text().splitlines()
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    ops = {}
    for k, (i, name) in enumerate(starts):
        end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
        ops[name] = sum(1 for l in lines[i:end] if l.startswith("\t") and "scratch_" in l.split(";")[0])
    rot = [n for n in ops if "chip_convILb1ELb0E" in n]              # chip_conv<true, false>
    obs = [n for n in ops if "chip_conv_obs" in n]
    assert len(rot) == 1 and len(obs) == 1, sorted(ops)
    assert ops[rot[0]] == 0 and ops[obs[0]] <= 2, (ops[rot[0]], ops[obs[0]])
    far = [n for n in ops if "chip_convILb1ELb1E" in n]              # (the general sequence's rotation stage keeps its call -- and its saves)
    assert len(far) == 1 and ops[far[0]] > 100
    kern = [n for n in ops if n.startswith("_Z22payne_post_chip_kernel")]
    assert len(kern) == 1
    m = re.search(r"\.name:\s+_Z22payne_post_chip_kernel\w*.*?\.vgpr_spill_count:\s+(\d+)", "\n".join(lines), re.S)
    assert m and int(m.group(1)) <= 16, m and m.group(1)


def test_instruction_census_of_the_post_kernels_phases(tmp_path):
    """tools/valu_census.py (the table behind profiles/rN_c2_valu_census.txt) compiles every phase of the C2 post kernel as a kernel of
    its own and counts the compiler's instructions: the tool still builds against the phase code, every phase is found, and the short
    forms of the per-pixel phases stay short (the general loops they fall back to are counted beside them)."""
    import subprocess
    import sys
    out = tmp_path / "census.txt"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "valu_census.py"), "--out", str(out)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    rows = {}
    for line in out.read_text().splitlines():
        f = line.split()
        if len(f) > 2 and f[0].startswith(("census_", "payne::fft_fixed")) and f[1].isdigit():
            rows[f[0]] = int(f[1])
    for k in ("census_first", "census_resample", "census_resample_general", "census_taper", "census_obs", "census_obs_general"):
        assert k in rows and rows[k] > 0, (k, rows)
    assert rows["census_resample"] < rows["census_resample_general"] and rows["census_obs"] < rows["census_obs_general"], rows
    assert rows["census_obs"] <= 200 and rows["census_first"] <= 200 and rows["census_resample"] <= 130, rows


def test_bench_line_is_small_and_complete():
    """bench.py's LAST stdout line is what the driver parses: the contract's fields, `roofline` and `cpu_baseline` in it, numbers only,
    under 4 KB whatever the run measured (round 5's 21 KB line was not parsed); the full record goes to bench_detail.json."""
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r5_bench_default.json")))       # a real run's full record (21 KB)
    full["end_to_end"].update({"iterations_per_s": 3.9e5, "calls_per_iteration": 37.1,
                               "fit": {"fit_wall_s": 0.0812, "iterations": 27411, "calls": 1017000, "logz": -1814.2, "logzerr": 0.21,
                                       "dlogz": 0.01, "walks": 25, "nlive": 512, "iterations_per_s": 3.4e5, "cpu_port_projected_s": 87.3,
                                       "what": "x" * 500}})
    line = bench.final_line(full)
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT and json.loads(text) == line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "end_to_end", "also_measured"):
        assert k in line, k
    assert line["value"] == pytest.approx(full["value"], rel=1e-5) and line["metric"] == full["metric"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in line["roofline"], k
    assert line["roofline"]["frac"] == pytest.approx(line["roofline"]["achieved"] / line["roofline"]["peak"], rel=1e-4)
    for k in ("value", "unit", "cores", "kind", "sample", "per_core"):
        assert k in line["cpu_baseline"], k
    for k in ("value", "frac_of_kernel_only", "iterations_per_s", "calls_per_iteration", "fit"):
        assert k in line["end_to_end"], k
    assert line["end_to_end"]["fit"]["fit_wall_s"] == pytest.approx(0.0812) and "what" not in line["end_to_end"]["fit"]
    assert set(line["also_measured"]) == set(full["also_measured"])
    for blk in line["also_measured"].values():
        assert {"value", "ms_per_step", "roofline_frac"} <= set(blk)
    assert "workload" in line["config"] and "model" not in line["config"]
    # no prose: no string in the line is longer than the workload sentence
    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert max(len(x) for x in strings(line)) <= 160
    # a run that measured much more (ten more configurations, a multi-rank table) still fits: optional blocks shrink first
    big = json.loads(json.dumps(full))
    for i in range(30):
        big["also_measured"]["X%d" % i] = big["also_measured"]["C5"]
    big["n_gpus"], big["per_rank_evals_per_s"], big["collective_backend"] = 8, [1.5e7] * 8, "nccl"
    line = bench.final_line(big)
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    assert "roofline" in line and "cpu_baseline" in line and line["n_gpus"] == 8
    # a side run that failed is reported, shortly
    big["also_measured"] = {"C5": {"error": "RuntimeError: " + "y" * 900}}
    assert len(bench.final_line(big)["also_measured"]["C5"]["error"]) <= 80


def test_bench_times_blocks_until_the_clock_ramp_is_behind():
    """bench.py's stopping rule for timed blocks: never fewer than --repeats, then until --min-timed-ms of timed work are behind (an idle
    MI355X needs 15-20 ms of work to reach its clocks: fifteen 20-step blocks -- the driver's command -- are 9 ms), bounded."""
    import bench
    blocks = []
    while bench.more_blocks(blocks, 15, 0.120):
        blocks.append(20 * 29e-6)                                  # the driver's command: 20 steps of 29 us
    assert len(blocks) == 207 and sum(blocks) >= 0.120 > sum(blocks[:-1])
    blocks = []
    while bench.more_blocks(blocks, 15, 0.120):
        blocks.append(200 * 29e-6)                                 # the default run: fifteen blocks are 87 ms
    assert len(blocks) == 21
    blocks = []
    while bench.more_blocks(blocks, 3, 0.120):
        blocks.append(3 * 1.3e-3)                                  # C5: three steps of 1.3 ms
    assert len(blocks) == 31
    assert not bench.more_blocks([1.0] * 15, 15, 0.120) and bench.more_blocks([1.0] * 14, 15, 0.0) and bench.more_blocks([], 0, 0.0)
    blocks = []
    while bench.more_blocks(blocks, 1, 1.0, cap=50):
        blocks.append(1e-9)
    assert len(blocks) == 50

