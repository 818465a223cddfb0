"""GPU parity: the HIP path through the C ABI against (a) golden vectors frozen from
the reference and (b) the numpy oracle on seeded inputs.

Tolerances (SURVEY.md 8(d), fp32 flux path vs the reference's fp64):
  per-pixel model flux |d| <= 1e-6 ; |dlnL| <= 2e-6 |lnL| + 5e-3 ; NaN pattern identical.
"""
import numpy as np
import pytest

import oracle as O
from thepayne_amd import synth, nnio, _lib
from helpers import SPEC_PARS, theta_full, yst_problem, lnl_tol

pytestmark = pytest.mark.gpu
FLUX_TOL = 1e-6


@pytest.fixture(scope="module")
def Engine():
    from thepayne_amd.engine import PayneEngine
    return PayneEngine


def _net(raw, kind="YST1"):
    return nnio.normalize_spec_net(raw, kind)


# every kernel variant that ships (payne_opts.variant, include/payne_hip.h): the defaults, and the code paths that
# differently shaped nets / spectra take, forced onto the C2 problem
VARIANTS = {"default": 0, "out_generic": 1, "post_generic": 2, "tw_global": 4, "post_full": 8, "no_prep": 16,
            "post_generic+tw_global+no_prep": 2 | 4 | 16, "out_generic+post_full": 1 | 8, "out_bk64": 1024, "out_rolled": 2048, "out_bk64+rolled": 1024 | 2048, "out_f32": 4096, "out_f32+rolled": 4096 | 2048,
            "rows_pixel": 262144, "rows_pixel+post_full": 262144 | 8, "rows_pixel+no_prep": 262144 | 16,
            "out_planes": 524288, "out_planes+rows_pixel": 524288 | 262144, "out_bf16x3": 1048576, "out_bf16x3+rows_pixel": 1048576 | 262144,
            "hid_f32": 2097152, "hid_f32+out_bf16x3": 2097152 | 1048576, "out_whole_tile": 16777216, "out_whole_tile+rows_pixel": 16777216 | 262144}
# ... of which these hand the post kernel rows in the frequency domain (the output layer's weights restated: payne_hip.h, payne_last_kernel kind 4)
FREQ_ROWS = {"default", "post_full", "no_prep", "out_rolled", "out_planes", "out_bf16x3", "hid_f32", "hid_f32+out_bf16x3", "out_whole_tile"}


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_lnlike_c2_against_reference_golden(Engine, golden, variant):
    g = golden("g4_lnlike_c2")
    cfg = synth.CONFIGS["C2"]
    raw = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], seed=0)
    eng = Engine(_net(raw), obs=(g["obs_wave"], g["obs_flux"], g["obs_eflux"]), b_max=512, variant=VARIANTS[variant])
    lnl = eng.lnlike_batch(theta_full(g["theta"])).cpu().numpy()
    err = np.abs(lnl - g["lnlike"])
    assert np.all(err <= lnl_tol(g["lnlike"])), (err.max(), np.argmax(err))
    assert eng.kernels_used()["rows"] == ("frequency" if variant in FREQ_ROWS else "pixels"), (variant, eng.kernels_used())
    # determinism + independence of batch position
    lnl2 = eng.lnlike_batch(theta_full(g["theta"][::-1].copy())).cpu().numpy()[::-1]
    assert np.array_equal(lnl, lnl2)
    # a batch larger than b_max is chunked by the host layer
    big = np.tile(theta_full(g["theta"]), (2, 1))[:700]
    lnl3 = eng.lnlike_batch(big).cpu().numpy()
    assert np.array_equal(lnl3[:512], lnl) and np.array_equal(lnl3[512:], lnl[:188])


def test_weights_split_on_their_way_into_lds_give_the_planes_split_at_set_up(Engine, golden):
    """The six-product output layer at C2 (PAYNE_V_OUT_BF16X3: payne_dense_dma3f_kernel) reads its weights as fp32 and splits them into the
    three bf16 planes inside the kernel; PAYNE_V_OUT_PLANES reads planes split once at payne_ctx_create (payne_dense_dma3_kernel).  Same parts, same
    products, same order: the network's rows -- pixels and frequency rows -- and the likelihoods are equal TO THE BIT."""
    from thepayne_amd import _lib
    g = golden("g4_lnlike_c2")
    cfg = synth.CONFIGS["C2"]
    raw = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], seed=0)
    th = theta_full(g["theta"])
    out = {}
    for name, v in (("split_in_kernel", _lib.V_OUT_BF16X3), ("planes", _lib.V_OUT_PLANES), ("split_in_kernel_px", _lib.V_OUT_BF16X3 | _lib.V_ROWS_PIXEL),
                    ("planes_px", _lib.V_OUT_PLANES | _lib.V_ROWS_PIXEL)):
        eng = Engine(_net(raw), obs=(g["obs_wave"], g["obs_flux"], g["obs_eflux"]), b_max=512, variant=v)
        lnl = eng.lnlike_batch(th).cpu().numpy()
        used = eng.kernels_used()
        rows = eng.predict_batch(th[:70], stage=0).cpu().numpy()             # (a ragged batch: 70 rows of a 64-row tile grid)
        out[name] = (lnl, rows, used)
        eng.close()
    assert "dma3f" in out["split_in_kernel"][2]["out"] and "dma3f" not in out["planes"][2]["out"], out["planes"][2]
    for a, b in (("split_in_kernel", "planes"), ("split_in_kernel_px", "planes_px")):
        assert np.array_equal(out[a][0], out[b][0], equal_nan=True), (a, np.nanmax(np.abs(out[a][0] - out[b][0])))
        assert np.array_equal(out[a][1], out[b][1], equal_nan=True), a
    err = np.abs(out["split_in_kernel"][0] - g["lnlike"])
    assert np.all(err <= lnl_tol(g["lnlike"]))


def test_output_layer_in_split_products_is_as_accurate_as_the_fp32_chain(Engine):
    """The default output layer at C2 multiplies operands split in two fp16 parts (three exact partial products, fp32 accumulation:
    payne_dense_dma2h_kernel; rows scaled by powers of two); PAYNE_V_OUT_BF16X3 is the three-bf16-part, six-product form,
    PAYNE_V_OUT_F32 the fp32 matrix instruction.  All against the SAME network evaluated in fp64: the split forms must be as
    accurate as the fp32 fma chain (whose own roundings dominate all three), pixel rows and frequency rows alike, and sit far inside
    the flux tolerance.  Labels 15 % outside the training box on every side and NaN labels ride along: the calibrated scale of the
    activations has room for them, and NaN rows stay NaN."""
    from thepayne_amd import _lib
    cfg = synth.CONFIGS["C2"]
    raw = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], seed=0)
    net = _net(raw)
    rng = np.random.default_rng(8)
    B = 512
    lab = net["xmin"][:4] + rng.uniform(0.02, 0.98, size=(B, 4)) * (net["xmax"][:4] - net["xmin"][:4])
    lab[:16] = net["xmin"][:4] + rng.uniform(-0.15, 1.15, size=(16, 4)) * (net["xmax"][:4] - net["xmin"][:4])
    th7 = np.column_stack([lab, np.zeros(B), np.zeros(B), np.full(B, 20000.0)])
    th = theta_full(th7)
    th[40, 0] = np.nan
    # fp64 forward with the fp32 weights (ystpred.py:41-58): encode, two leaky-ReLU layers, linear output
    x = (lab - net["xmin"][:4]) / (net["xmax"][:4] - net["xmin"][:4]) - 0.5
    h = x.astype(np.float32).astype(np.float64)             # (the kernel encodes in fp64 and rounds to fp32)
    for W, b, act in net["layers"]:
        h = h @ W.astype(np.float64).T + b.astype(np.float64)
        if act == _lib.ACT_LRELU:
            h = np.maximum(h, 0.01 * h)
    ok = np.ones(B, bool); ok[40] = False
    errs, used = {}, {}
    for name, variant in (("f16x2", 0), ("bf16x3", _lib.V_OUT_BF16X3), ("f32", _lib.V_OUT_F32)):
        eng = Engine(net, obs=None, b_max=B, variant=variant)
        got = eng.predict_batch(th, stage=0).cpu().numpy().astype(np.float64)
        used[name] = eng.kernels_used()["out"]
        assert np.all(np.isnan(got[40])) and np.all(np.isfinite(got[ok]))
        errs[name] = np.abs(got[ok] - h[ok])
        eng.close()
    assert "dma2h" in used["f16x2"] and "dma3f" in used["bf16x3"], used
    # the hidden layers are the same fp32 kernels in all runs: their rounding is common to the errors
    for k in errs:
        assert errs[k].max() <= FLUX_TOL, (k, errs[k].max())
    rms = {k: float(np.sqrt(np.mean(v ** 2))) for k, v in errs.items()}
    for k in ("f16x2", "bf16x3"):
        assert rms[k] <= 1.25 * rms["f32"] + 1e-9, rms
        assert errs[k].max() <= 1.5 * errs["f32"].max() + 1e-8, (k, errs[k].max(), errs["f32"].max())


def test_second_layer_on_fp16_pairs_against_fp64_and_the_fp32_instruction(Engine):
    """The first hidden-layer launch multiplies the second layer on fp16 pairs (hk_tile_h2: the first layer transposed on the matrix
    cores, split in registers; three v_mfma_f32_16x16x32_f16 a 32-deep step); PAYNE_V_HID_F32 keeps the fp32 matrix instruction.  With
    an output layer that copies the second layer's units the network's rows ARE its activations: both forms against fp64, the pair form
    as accurate as the fp32 one, for a batch that fills the machine several times over (2048 candidates = 640 tiles), thirty times --
    every call the same bits (a 16-deep legacy instruction at the end of the chain returned wrong accumulator halves now and then while
    this was written) --, labels outside the box and a NaN row included."""
    from thepayne_amd import _lib
    cfg = synth.CONFIGS["C2"]
    raw = synth.make_yst_net(npix=512, lam0=cfg["lam0"], R_fwhm=cfg["R"], H=300, seed=0)
    raw["w_array_2"][:] = 0
    raw["b_array_2"][:] = 0
    for k in range(300):
        raw["w_array_2"][k, k] = 1.0
    net = _net(raw)
    B = 2048
    rng = np.random.default_rng(4)
    lab = net["xmin"][:4] + rng.uniform(0.0, 1.0, size=(B, 4)) * (net["xmax"][:4] - net["xmin"][:4])
    lab[:32] = net["xmin"][:4] + rng.uniform(-0.15, 1.15, size=(32, 4)) * (net["xmax"][:4] - net["xmin"][:4])
    th = theta_full(np.column_stack([lab, np.zeros(B), np.zeros(B), np.full(B, 20000.0)]))
    th[77, 1] = np.nan
    ref = _fp64_forward(net, lab)[:, :300]
    ok = np.ones(B, bool); ok[77] = False
    err = {}
    for name, v in (("f16x2", _lib.V_OUT_F32), ("f32", _lib.V_OUT_F32 | _lib.V_HID_F32)):
        eng = Engine(net, obs=None, b_max=B, variant=v)
        first = eng.predict_batch(th, stage=0).cpu().numpy()
        for rep in range(30 if name == "f16x2" else 2):
            again = eng.predict_batch(th, stage=0).cpu().numpy()
            assert np.array_equal(again, first, equal_nan=True), (name, rep)
        assert np.all(np.isnan(first[77, :300])) and np.all(np.isfinite(first[ok]))
        err[name] = np.abs(first[ok, :300].astype(np.float64) - ref[ok])
        eng.close()
    rms = {k: float(np.sqrt(np.mean(v ** 2))) for k, v in err.items()}
    assert err["f16x2"].max() <= FLUX_TOL and err["f32"].max() <= FLUX_TOL, {k: v.max() for k, v in err.items()}
    assert rms["f16x2"] <= 1.5 * rms["f32"] + 1e-9, rms


@pytest.mark.parametrize("scale", [1e-4, 1.0, 3e3])
def test_calibrated_scales_follow_the_networks_own_magnitudes(Engine, scale):
    """fp16 has five exponent bits: the pair forms scale every weight row and every layer's activations by powers of two found at
    payne_ctx_create (the rows' maxima; the activations' maxima over the label box).  The same network with its hidden activations
    blown up or shrunk -- W1, b1 times `scale`, W2 divided by it: the same function -- must come out as accurately: against fp64, and no
    worse than the six-product / fp32-instruction forms on the same weights."""
    from thepayne_amd import _lib
    cfg = synth.CONFIGS["C2"]
    raw = synth.make_yst_net(npix=1024, lam0=cfg["lam0"], R_fwhm=cfg["R"], H=300, seed=3)
    raw["w_array_1"] = (raw["w_array_1"].astype(np.float64) * scale).astype(np.float32)
    raw["b_array_1"] = (raw["b_array_1"].astype(np.float64) * scale).astype(np.float32)
    raw["w_array_2"] = (raw["w_array_2"].astype(np.float64) / scale).astype(np.float32)
    net = _net(raw)
    B = 256
    rng = np.random.default_rng(12)
    lab = net["xmin"][:4] + rng.uniform(0.0, 1.0, size=(B, 4)) * (net["xmax"][:4] - net["xmin"][:4])
    th = theta_full(np.column_stack([lab, np.zeros(B), np.zeros(B), np.full(B, 20000.0)]))
    ref = _fp64_forward(net, lab)
    err = {}
    for name, v in (("pairs", 0), ("older", _lib.V_OUT_BF16X3 | _lib.V_HID_F32)):
        eng = Engine(net, obs=None, b_max=B, variant=v)
        got = eng.predict_batch(th, stage=0).cpu().numpy().astype(np.float64)
        used = eng.kernels_used()
        eng.close()
        assert ("dma2h" in used["out"]) == (name == "pairs"), used
        err[name] = np.abs(got - ref)
    assert err["pairs"].max() <= FLUX_TOL and err["older"].max() <= FLUX_TOL, {k: v.max() for k, v in err.items()}
    rms = {k: float(np.sqrt(np.mean(v ** 2))) for k, v in err.items()}
    assert rms["pairs"] <= 1.5 * rms["older"] + 1e-9, rms


def test_rows_in_the_frequency_domain_from_fp16_pairs_against_the_six_product_form(Engine, golden):
    """The same comparison where the rows are handed over as their transform (the default at C2: the restated weights have a wide
    range of magnitudes from row to row, which the per-row scales of the fp16 planes absorb): likelihoods and getspec outputs of
    the two split forms against each other and against the reference's frozen values."""
    from thepayne_amd import _lib
    g = golden("g4_lnlike_c2")
    cfg = synth.CONFIGS["C2"]
    raw = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], seed=0)
    th = theta_full(g["theta"])
    res = {}
    for name, v in (("f16x2", 0), ("bf16x3", _lib.V_OUT_BF16X3)):
        eng = Engine(_net(raw), obs=(g["obs_wave"], g["obs_flux"], g["obs_eflux"]), b_max=512, variant=v)
        lnl = eng.lnlike_batch(th).cpu().numpy()
        assert eng.kernels_used()["rows"] == "frequency"
        spec = eng.predict_batch(th[:64], stage=2, fwhm_R=True).cpu().numpy().astype(np.float64)
        res[name] = (lnl, spec)
        eng.close()
    for name, (lnl, spec) in res.items():
        err = np.abs(lnl - g["lnlike"])
        assert np.all(err <= lnl_tol(g["lnlike"])), (name, err.max())
    fin = np.isfinite(res["f16x2"][1]) & np.isfinite(res["bf16x3"][1])
    assert np.array_equal(np.isnan(res["f16x2"][1]), np.isnan(res["bf16x3"][1]))
    assert np.abs(res["f16x2"][1] - res["bf16x3"][1])[fin].max() <= 2e-7          # (two roundings of the same fp32-class rows)
    d = np.abs(res["f16x2"][0] - res["bf16x3"][0]); ok = np.isfinite(d)
    assert np.all(d[ok] <= 0.25 * lnl_tol(g["lnlike"][ok]))


@pytest.mark.parametrize("H", [300, 100])
def test_output_layer_on_many_big_tiles_against_fp64(Engine, H):
    """The output layer with many tiles per compute unit (C5's 65 536 pixels): payne_dense_big3_kernel -- persistent workgroups,
    128 x 256 tiles, a four-stage ring across tile boundaries, blocked tile order -- against the SAME network in fp64 and against the
    64 x 128-tile form (PAYNE_V_OUT_SMALL_TILES), which must give the same values to the last bit but the order of the k-steps'
    partial sums (both accumulate a 32 x 32 block's six products per 16-deep step in the same order: bit-equal) -- and the default
    there, the same kernel on two fp16 planes an operand (three products a block), against fp64.  Two batch sizes:
    512 (whole tiles: the big-tile kernel) and 500 (not a multiple of 128: the launch must fall back by itself).  H = 100: a width
    whose padded tail (128 columns) is cut at the seventh 16-deep step."""
    from thepayne_amd import _lib
    cfg = synth.CONFIGS["C5"]
    raw = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=H, seed=0)
    net = _net(raw)
    rng = np.random.default_rng(21)
    B = 512
    lab = net["xmin"][:4] + rng.uniform(0.02, 0.98, size=(B, 4)) * (net["xmax"][:4] - net["xmin"][:4])
    th = theta_full(np.column_stack([lab, np.zeros(B), np.zeros(B), np.full(B, 60000.0)]))
    ref = _fp64_forward(net, lab)
    got, used = {}, {}
    for name, variant in (("big_h2", 0), ("small_h2", _lib.V_OUT_SMALL_TILES), ("big", _lib.V_OUT_BF16X3),
                          ("small", _lib.V_OUT_BF16X3 | _lib.V_OUT_SMALL_TILES)):
        eng = Engine(net, obs=None, b_max=B, variant=variant)
        got[name] = eng.predict_batch(th, stage=0).cpu().numpy()
        used[name] = eng.kernels_used()["out"]
        if name in ("big", "big_h2"):
            got[name + "_part"] = eng.predict_batch(th[:500], stage=0).cpu().numpy()       # 500 rows: not whole 128-row tiles
        eng.close()
    assert used["big_h2"] == "payne_dense_big3_kernel<true>" and used["big"] == "payne_dense_big3_kernel<false>", used
    assert "dma2h" in used["small_h2"] and "dma3_kernel" in used["small"], used
    # the tile shapes of a form give the same values to the last bit, whole tiles or not
    for big, small in (("big", "small"), ("big_h2", "small_h2")):
        assert np.abs(got[big].astype(np.float64) - ref).max() <= FLUX_TOL
        assert np.array_equal(got[big], got[small])
        assert np.array_equal(got[big + "_part"], got[small][:500])
    # the default (two fp16 planes, three products a block) is as accurate in the mean as the six-product form
    e2, e3 = np.abs(got["big_h2"].astype(np.float64) - ref), np.abs(got["big"].astype(np.float64) - ref)
    assert np.sqrt(np.mean(e2 ** 2)) <= 1.25 * np.sqrt(np.mean(e3 ** 2)) + 1e-9, (e2.max(), e3.max())


def _fp64_forward(net, lab):
    from thepayne_amd import _lib
    nl = lab.shape[1]
    x = (lab - net["xmin"][:nl]) / (net["xmax"][:nl] - net["xmin"][:nl]) - 0.5
    h = x.astype(np.float32).astype(np.float64)
    for W, b, act in net["layers"]:
        h = h @ W.astype(np.float64).T + b.astype(np.float64)
        if act == _lib.ACT_LRELU:
            h = np.maximum(h, 0.01 * h)
    return h


@pytest.mark.parametrize("H,npix,B,D", [(16, 160, 1, 4), (40, 1000, 63, 4), (64, 1029, 65, 4), (100, 4101, 130, 4),
                                        (300, 777, 200, 5), (296, 640, 70, 4), (304, 515, 33, 4), (320, 2048, 64, 4), (352, 1500, 100, 4)])
def test_dense_layers_on_odd_shapes(Engine, H, npix, B, D):
    """Ragged sizes through every form of the dense layers: hidden widths that are / are not multiples of 32 (one, two,
    ten, eleven k-steps; 296 / 300 / 304: the widths whose operand tiles go straight into LDS, K a multiple of 16 or not; 320 and 352
    take the register-staged tiles, 352 in two chunks), pixel counts that are not multiples of the 128-column
    tile, batches around the 64-row tile, the fifth label.  Output of the network (stage 0) against fp64."""
    from thepayne_amd import _lib
    raw = synth.make_yst_net(npix=npix, H=H, seed=H + npix, D=D)
    net = _net(raw)
    rng = np.random.default_rng(B)
    lab = net["xmin"][:D] + rng.uniform(0.05, 0.95, size=(B, D)) * (net["xmax"][:D] - net["xmin"][:D])
    ref = _fp64_forward(net, lab)
    for variant in (0, _lib.V_OUT_F32, _lib.V_OUT_ROLLED, _lib.V_OUT_GENERIC):
        eng = Engine(net, obs=None, b_max=max(B, 8), variant=variant)
        th = np.full((B, eng.ncols), np.nan)
        th[:, 0:4] = lab[:, 0:4]
        if D == 5:
            th[:, 6] = lab[:, 4]
        th[:, 4], th[:, 5], th[:, 7] = 0.0, 0.0, 20000.0
        got = eng.predict_batch(th, stage=0).cpu().numpy().astype(np.float64)
        eng.close()
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= FLUX_TOL, (variant, np.abs(got - ref).max())


@pytest.mark.parametrize("npix,B", [(4096, 512), (4096, 100), (1000, 64), (2048, 129), (640, 7), (8192, 256)])
def test_output_tile_in_two_halves_is_the_whole_tile_to_the_bit(Engine, npix, B):
    """One output tile a compute unit: the tile is finished in two halves (payne_dense_dma2hh_kernel: activations resident in LDS, the weights'
    halves streamed one after the other, the left half's rows stored under the right half's products) -- the same products in the same
    order as the whole tile (payne_dense_dma2h_kernel<5, 64>, PAYNE_V_OUT_WHOLE_TILE): the same rows to the bit, whole and ragged tiles,
    pixel rows and rows in the frequency domain, and the same likelihoods."""
    from thepayne_amd import _lib
    cfg = synth.CONFIGS["C2"]
    raw = synth.make_yst_net(npix=npix, lam0=cfg["lam0"], R_fwhm=cfg["R"], H=300, seed=npix + B)
    net = _net(raw)
    obs = synth.obs_grid(raw["wavelength"], max(64, int(0.85 * npix)))
    th = theta_full(synth.draw_candidates(B, seed=B))
    th[0, 5] = 0.0                                                   # (a candidate that does not rotate)
    e0 = Engine(net, obs=(obs,), b_max=B)
    clean = e0.predict_batch(th[:1], stage=2, fwhm_R=True).cpu().numpy()[0].astype(np.float64)
    e0.close()
    flux = clean + np.random.default_rng(0).normal(0, 0.01, len(obs))
    out = {}
    for v in (0, _lib.V_OUT_WHOLE_TILE):
        eng = Engine(net, obs=(obs, flux, np.full(len(obs), 0.01)), b_max=B, variant=v)
        rows = eng.predict_batch(th, stage=0).cpu().numpy()
        name0 = eng.kernels_used()["out"]
        lnl = eng.lnlike_batch(th).cpu().numpy()
        out[v] = (rows, lnl, name0, eng.kernels_used()["out"], eng.kernels_used()["rows"])
        eng.close()
    a, b = out[0], out[_lib.V_OUT_WHOLE_TILE]
    tiles = ((B + 63) // 64) * ((npix + 127) // 128)
    if tiles <= 256:                                                 # (a tile a compute unit: the two kernels this test is about)
        assert a[2] == "payne_dense_dma2hh_kernel<5>" and b[2] == "payne_dense_dma2h_kernel<5, 64>", (a[2], b[2])
    else:
        assert a[2] == b[2], (a[2], b[2])
    assert np.array_equal(a[0], b[0]), np.abs(a[0] - b[0]).max()
    assert np.array_equal(np.nan_to_num(a[1]), np.nan_to_num(b[1])) and int(np.isfinite(a[1]).sum()) >= B - 4
    assert a[4] == b[4]


@pytest.mark.parametrize("domain", ["frequency", "pixels"])
def test_predict_stages_against_reference_golden(Engine, golden, domain):
    """predictspec, the spectrum after rotation (Vrot = 0 and values from 1e-3 to 50 km/s) and getspec on the
    observed grid against the reference's own outputs -- with the output layer handing over transformed rows (default here: 1024
    pixels on a geometric grid) and pixel rows (PAYNE_V_ROWS_PIXEL: what every other shape runs)."""
    from thepayne_amd import _lib
    g = golden("g2_getspec")
    raw = synth.make_yst_net(npix=1024, H=64, seed=5, line_depth=0.3)
    eng = Engine(_net(raw), obs=(g["obs_wave"],), b_max=64, variant=0 if domain == "frequency" else _lib.V_ROWS_PIXEL)
    lab = g["labels"]
    rows = g["theta_rows"]
    th = np.full((len(rows), eng.ncols), np.nan)
    th[:, 0:4] = lab
    th[:, 4], th[:, 5], th[:, 7] = rows[:, 0], rows[:, 1], rows[:, 2]
    s0 = eng.predict_batch(th[:1], stage=0).cpu().numpy()[0]
    assert np.abs(s0 - g["raw"]).max() <= FLUX_TOL
    assert eng.kernels_used()["rows"] == "pixels"           # (the network's own output is always asked for in pixels)
    for v, ref in zip(g["vrot_values"], g["after_rot"]):
        t = th[:1].copy(); t[0, 5] = v
        s1 = eng.predict_batch(t, stage=1).cpu().numpy()[0]
        assert np.abs(s1 - ref).max() <= FLUX_TOL, v
    assert eng.kernels_used()["rows"] == domain
    s2 = eng.predict_batch(th, stage=2, fwhm_R=True).cpu().numpy()
    assert np.array_equal(np.isnan(s2), np.isnan(g["final"]))
    assert np.nanmax(np.abs(s2 - g["final"])) <= FLUX_TOL
    assert np.isnan(g["final"]).any()


def test_ann_kinds_against_reference_golden(Engine, golden):
    g = golden("g1_ann")
    lab = g["labels"]
    th = np.full((len(lab), 12), np.nan); th[:, :4] = lab; th[:, 4:6] = 0.0
    eng = Engine(_net(synth.make_yst_net(npix=256, H=48, seed=3)), b_max=16)
    assert np.abs(eng.predict_batch(th, stage=0).cpu().numpy() - g["yst"]).max() <= FLUX_TOL
    rawk = synth.make_yst_net(npix=256, H=48, seed=3); rawk["x_min"][0] /= 1000; rawk["x_max"][0] /= 1000
    eng = Engine(_net(rawk), b_max=16)
    assert np.abs(eng.predict_batch(th, stage=0).cpu().numpy() - g["yst_kfix"]).max() <= FLUX_TOL
    eng = Engine(_net(synth.make_yst_net(npix=256, H=48, seed=4, D=5)), b_max=16)
    th5 = th.copy(); th5[:, 6] = g["labels5"][:, 4]
    assert np.abs(eng.predict_batch(th5, stage=0).cpu().numpy() - g["yst5"]).max() <= FLUX_TOL
    for kind in ("LinNet", "SMLP"):       # the reference itself runs these in torch fp32
        eng = Engine(_net(synth.make_torch_net(kind, npix=256, seed=7), kind), b_max=16)
        assert np.abs(eng.predict_batch(th, stage=0).cpu().numpy() - g[kind.lower()]).max() <= 3e-6, kind


# The reference evaluates LinNet / SMLP in torch fp32 (predictspec.py:61-74): ITS outputs carry the rounding of five (three)
# 300-term fp32 sums in torch's order, ours the same sums in the matrix instruction's order.  Allowance on top of SURVEY 8(d)'s
# 1e-6 for the fp64 pipeline behind the network: 1e-6 on the network's output (the numpy-fp32 restatement of the same net sits
# 1.2e-7 from torch; test_oracle_golden.py::test_g14_*), hence 2e-6 on a flux; the lnL tolerance is 8(d)'s, unchanged.
TORCH_FP32_FLUX_TOL = 2e-6


@pytest.mark.parametrize("kind,seed", [("LinNet", 21), ("SMLP", 22)])
def test_default_network_at_full_width_against_reference_golden(Engine, golden, kind, seed):
    """FitPayne's default NNtype='LinNet' (fitstar.py:81) as the reference defines it -- D -> 300 x5 -> Npix, five sigmoids
    (NNmodels.py:140-168) -- and SMLP 3 x 300, on the C2 shape (4096 model pixels, 3600 observed, H = 300): network output,
    getspec on the observed grid and lnlikefn against vectors frozen from the reference (g14)."""
    from thepayne_amd import _lib
    g = golden("g14_%s300" % kind.lower())
    cfg = synth.CONFIGS["C2"]
    raw = synth.make_torch_net(kind, npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=(300, 300, 300), seed=seed)
    eng = Engine(_net(raw, kind), obs=(g["obs_wave"], g["obs_flux"], g["obs_eflux"]), b_max=128)
    lab = g["labels"]
    th = np.full((len(lab), eng.ncols), np.nan); th[:, :4] = lab; th[:, 4:6] = 0.0
    s0 = eng.predict_batch(th, stage=0).cpu().numpy()
    assert np.abs(s0 - g["raw"]).max() <= TORCH_FP32_FLUX_TOL, np.abs(s0 - g["raw"]).max()
    thd = theta_full(g["theta"])
    s2 = eng.predict_batch(thd[:4], stage=2, fwhm_R=True).cpu().numpy()
    assert np.abs(s2 - g["getspec4"]).max() <= TORCH_FP32_FLUX_TOL, np.abs(s2 - g["getspec4"]).max()
    lnl = eng.lnlike_batch(thd).cpu().numpy()
    ref = g["lnlike"]
    assert np.all(np.isfinite(ref)) and np.all(np.abs(lnl - ref) <= lnl_tol(ref)), np.abs(lnl - ref).max()
    # which kernels a net of this depth takes: every layer on the matrix cores, the output layer as bf16 products
    names = eng.kernels_used()
    assert names["out"].startswith("payne_dense_dma"), names
    assert names["hidden"] == "payne_dense_hidden_kernel<false, 4>", names
    # LinNet's hidden layers past the second as ONE launch with hand-offs inside a row block (PAYNE_V_HID_CHAIN: measured slower than a
    # launch per layer, kept as a variant): the same tiles, the same bits, whatever the batches before it were (the hop counters only grow)
    if kind == "LinNet":
        e1 = Engine(_net(raw, kind), obs=(g["obs_wave"], g["obs_flux"], g["obs_eflux"]), b_max=128, variant=_lib.V_HID_CHAIN)
        assert np.array_equal(e1.lnlike_batch(thd).cpu().numpy(), lnl)
        assert e1.kernels_used()["hidden"] == "payne_dense_chain_kernel"
        for nrow in (5, 128, 33, 64, 1, 128):
            assert np.array_equal(e1.lnlike_batch(thd[:nrow]).cpu().numpy(), lnl[:nrow]), nrow
        e1.close()
    # a sigmoid net's first launch runs its tiles on eight waves (the first layer's forty activations a lane become twenty-four); four
    # waves -- what leaky-ReLU nets use -- split the same arithmetic differently: the same bits
    if kind == "LinNet":
        e3 = Engine(_net(raw, kind), obs=(g["obs_wave"], g["obs_flux"], g["obs_eflux"]), b_max=128, variant=_lib.V_HID_WAVES4)
        assert np.array_equal(e3.lnlike_batch(thd).cpu().numpy(), lnl)
        e3.predict_batch(th[:3], stage=0); assert e3.kernels_used()["hidden"] == "payne_dense_hidden_kernel<false, 4>"
        e3.close()
        # (a three-layer sigmoid net: the first launch is the only hidden launch, its kernel's name says how many waves)
        for D, nl in ((4, "4"), (5, "PAYNE_MAX_LABELS")):                        # (five labels: Vmic in column 6, the kernel's other label count)
            raw3 = synth.make_yst_net(npix=512, H=300, seed=5, D=D)
            n3 = _net(raw3); n3["layers"] = [(w, b, _lib.ACT_SIGMOID if a != _lib.ACT_NONE else a) for (w, b, a) in n3["layers"]]
            th3 = np.full((40, 12), np.nan); th3[:, :4] = thd[:40, :4]; th3[:, 4:6] = 0.0; th3[:, 6] = np.linspace(0.6, 2.2, 40)
            outs = {}
            for v in (0, _lib.V_HID_WAVES4):
                e4 = Engine(n3, b_max=40, variant=v)
                outs[v] = e4.predict_batch(th3, stage=0).cpu().numpy()
                want = "payne_dense_hidden_kernel<true, %s, 8>" % nl if v == 0 else "payne_dense_hidden_kernel<true, %s>" % nl
                assert e4.kernels_used()["hidden"] == want, e4.kernels_used()
                e4.close()
            assert np.array_equal(outs[0], outs[_lib.V_HID_WAVES4]) and np.all(np.isfinite(outs[0])), D
            # ... and against the fp64 restatement of the same net
            xs = (th3[:, [0, 1, 2, 3, 6][:D]] - n3["xmin"]) / (n3["xmax"] - n3["xmin"]) - 0.5
            h = xs
            for (w, b, a) in n3["layers"]:
                z = h @ w.astype(np.float64).T + b.astype(np.float64)
                h = 1.0 / (1.0 + np.exp(-z)) if a == _lib.ACT_SIGMOID else z
            assert np.abs(outs[0] - h).max() <= TORCH_FP32_FLUX_TOL, (D, np.abs(outs[0] - h).max())
    # other shipped forms of the output layer on the same net: same likelihoods
    for v in (1, 4096, 2048):
        e2 = Engine(_net(raw, kind), obs=(g["obs_wave"], g["obs_flux"], g["obs_eflux"]), b_max=128, variant=v)
        l2 = e2.lnlike_batch(thd).cpu().numpy()
        assert np.all(np.abs(l2 - ref) <= lnl_tol(ref)), (v, np.abs(l2 - ref).max())


def test_activations_over_the_whole_fp32_range():
    """The layers' activation functions as their epilogues compute them (payne_activation_batch).  torch.sigmoid of NNmodels.py:92-121 is
    2^n v_exp_f32(f) and v_rcp_f32 + a Newton step here, not the library's expf and quotient: within 2.5 ulp of the exact value for every
    finite argument (2.4 measured: the exponential's 1.35, the sum's 0.5, the quotient's 0.5; the library's pair: 2), on average not
    behind torch's own fp32 sigmoid, 1 and 0 at the infinities, NaN kept; leaky ReLU and none exact."""
    import torch
    from thepayne_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    z = np.concatenate([np.linspace(-110.0, 110.0, 1 << 21), rng.normal(0, 1, 1 << 18), rng.normal(0, 12, 1 << 18),
                        np.ldexp(rng.normal(0, 1, 1 << 16), rng.integers(-140, 120, 1 << 16)),
                        [0.0, -0.0, 87.9, -87.9, 88.0, -88.0, 88.1, -88.1, 103.9, -103.9, 1e30, -1e30, 3.4e38, -3.4e38, 1e-45, -1e-45,
                         np.inf, -np.inf, np.nan]]).astype(np.float32)
    zd = torch.as_tensor(z, device="cuda")
    out = torch.empty_like(zd)

    def run(act):
        rc = lib.payne_activation_batch(zd.data_ptr(), len(z), act, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        return out.cpu().numpy()

    y = run(_lib.ACT_SIGMOID)
    z64 = z.astype(np.float64)
    with np.errstate(over="ignore", invalid="ignore"):
        ref = 1.0 / (1.0 + np.exp(-z64))
    fin = np.isfinite(z64)
    assert np.isnan(y[~fin][-1]) and y[~fin][0] == 1.0 and y[~fin][1] == 0.0, y[~fin]           # +inf, -inf, NaN
    assert np.all(np.isfinite(y[fin])) and np.all(y[fin] >= 0.0) and np.all(y[fin] <= 1.0)
    normal = fin & (ref >= 1.2e-38)                        # (below: the exact value is a denormal, ours 0 or a denormal)
    ulp = np.spacing(ref[normal].astype(np.float32)).astype(np.float64)
    err = np.abs(y[normal].astype(np.float64) - ref[normal]) / ulp
    assert err.max() <= 2.5, (err.max(), z[normal][np.argmax(err)])
    assert np.all(y[fin & ~normal] <= 1.2e-38)
    # not worse than the pair it replaces (torch's own fp32 sigmoid on the same arguments: the reference's arithmetic)
    t = torch.sigmoid(zd).cpu().numpy()
    err_t = np.abs(t[normal].astype(np.float64) - ref[normal]) / ulp
    assert err.mean() <= err_t.mean() * 1.5 + 0.05, (err.mean(), err_t.mean())
    # monotone where a step of the sorted block moves the exact value by more than the allowance (z < 5)
    blk = y[: 1 << 21][z[: 1 << 21] < 5.0]
    assert np.all(np.diff(blk) >= 0.0)
    yl = run(_lib.ACT_LRELU)
    with np.errstate(invalid="ignore", over="ignore"):
        refl = np.where(z > 0, z, np.float32(0.01) * z).astype(np.float32)
    assert np.array_equal(yl[fin], refl[fin]) and np.isnan(yl[-1]) and yl[-3] == np.inf and yl[-2] == -np.inf
    assert np.array_equal(run(_lib.ACT_NONE)[:-1], z[:-1])
    assert lib.payne_activation_batch(zd.data_ptr(), 4, 7, out.data_ptr(), None) != 0


@pytest.mark.parametrize("kind,H,npix,nobs,B", [("LinNet", (50, 37, 96), 1000, 700, 19), ("LinNet", (128, 128, 128), 2048, 1800, 40),
                                                ("SMLP", (300, 300, 300), 3000, 2500, 33), ("SMLP", (33, 65, 17), 512, 400, 7)])
def test_torch_nets_of_any_shape_vs_oracle(Engine, kind, H, npix, nobs, B):
    """LinNet / SMLP with hidden widths that differ from layer to layer (the register-staged dense kernel, no bf16 planes), that are
    not multiples of the tiles, and equal (the matrix-core path), spectra that are not powers of two, odd batches: network output
    and likelihood against the numpy-fp32 restatement of the reference's torch forward pass."""
    raw = synth.make_torch_net(kind, npix=npix, H=H, seed=5)
    obs = synth.obs_grid(raw["wavelength"], nobs, inset=0.06 * (raw["wavelength"][-1] - raw["wavelength"][0]))
    th7 = synth.draw_candidates(B, seed=npix + H[0])
    flux0 = O.genspec(raw, list(theta_full(th7[:1])[0, :8]), outwave=obs)[1]
    flux = flux0 + np.random.default_rng(3).normal(0, 0.01, nobs)
    eflux = np.full(nobs, 0.01)
    eng = Engine(_net(raw, kind), obs=(obs, flux, eflux), b_max=16)
    th = theta_full(th7)
    s0 = eng.predict_batch(th, stage=0).cpu().numpy()
    ref0 = np.array([O.torchnet_forward(raw, t[:4]) for t in th7])
    assert np.abs(s0 - ref0).max() <= TORCH_FP32_FLUX_TOL, np.abs(s0 - ref0).max()
    L = O.OracleLikelihood(raw, obs, flux, eflux, SPEC_PARS)
    with np.errstate(all="ignore"):
        ref = np.array([L.lnlikefn(t) for t in th7])
    lnl = eng.lnlike_batch(th).cpu().numpy()
    assert np.array_equal(np.isnan(lnl), np.isnan(ref))
    ok = np.isfinite(ref)
    assert ok.sum() >= B - 2 and np.all(np.abs(lnl[ok] - ref[ok]) <= lnl_tol(ref[ok])), np.abs(lnl[ok] - ref[ok]).max()


def test_lnlike_modpoly_against_reference_golden(Engine, golden):
    g = golden("g4_lnlike_modpoly")
    cfg = synth.CONFIGS["small"]
    raw = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=64, seed=0)
    eng = Engine(_net(raw), obs=(g["obs_wave"], g["obs_flux"], g["obs_eflux"]), npoly=3, b_max=64)
    th = theta_full(g["theta"][:, :7], npoly=3, pc=g["theta"][:, 7:10])
    lnl = eng.lnlike_batch(th).cpu().numpy()
    assert np.all(np.abs(lnl - g["lnlike"]) <= lnl_tol(g["lnlike"]))


@pytest.mark.parametrize("tag,photscale", [("scaled", True), ("dist", False)])
def test_lnlike_joint_against_reference_golden(Engine, golden, tag, photscale):
    g = golden("g4_lnlike_joint_" + tag)
    cfg = synth.CONFIGS["small"]
    raw = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=64, seed=0)
    phot = synth.make_phot_nets()
    obs_phot = {f: (m, e) for f, m, e in zip(phot["filters"], g["obs_mag"], g["obs_magerr"])}
    eng = Engine(_net(raw), obs=(g["obs_wave"], g["obs_flux"], g["obs_eflux"]), phot=phot, obs_phot=obs_phot,
                 photscale=photscale, b_max=64)
    names = [str(s) for s in g["fitpars_i"]]
    th = np.full((len(g["theta"]), eng.ncols), np.nan)
    col = {n: i for i, n in enumerate(['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Vmic', 'Inst_R'])}
    col.update({'log(A)': 8, 'log(R)': 8, 'Dist': 9, 'Av': 10, 'Rv': 11})
    for j, n in enumerate(names):
        th[:, col[n]] = g["theta"][:, j]
    lnl = eng.lnlike_batch(th).cpu().numpy()
    ref = g["lnlike"]
    assert np.array_equal(np.isnan(lnl), np.isnan(ref))          # one draw has Inst_R above the ANN's R -> NaN
    ok = np.isfinite(ref)
    assert np.all(np.abs(lnl[ok] - ref[ok]) <= lnl_tol(ref[ok])), np.abs(lnl[ok] - ref[ok]).max()


def test_sed_against_reference_golden(Engine, golden):
    g = golden("g5_sed")
    phot = synth.make_phot_nets()
    eng = Engine(phot=phot, b_max=32)
    p = g["pars"]
    nanc = np.full(len(p), np.nan)
    dist = np.column_stack([p[:, 0], p[:, 1], p[:, 2], p[:, 3], p[:, 4], p[:, 5], p[:, 6], p[:, 7], nanc])
    scal = np.column_stack([p[:, 0], p[:, 1], p[:, 2], p[:, 3], p[:, 4], p[:, 5], nanc, nanc, p[:, 8]])
    assert np.abs(eng.sed_batch(dist).cpu().numpy() - g["mags_dist"]).max() <= 1e-9
    assert np.abs(eng.sed_batch(scal).cpu().numpy() - g["mags_scaled"]).max() <= 1e-9


@pytest.mark.parametrize("npix,nobs,H,B", [(256, 200, 32, 5), (1000, 700, 50, 37), (3000, 2500, 96, 64),
                                            (8192, 7000, 300, 24), (16384, 15000, 64, 8)])
def test_lnlike_vs_oracle_shapes(Engine, npix, nobs, H, B):
    """Ragged sizes: npix not a power of two, hidden width not a multiple of 4/32,
    batch not a multiple of the tile; big spectra up to the LDS limit."""
    raw = synth.make_yst_net(npix=npix, H=H, seed=11, line_depth=0.1)
    obs = synth.obs_grid(raw["wavelength"], nobs, inset=0.06 * (raw["wavelength"][-1] - raw["wavelength"][0]))
    th7 = synth.draw_candidates(B, seed=npix)
    ref_flux = np.array([O.genspec(raw, list(theta_full(t)[0, :8]), outwave=obs)[1] for t in th7])
    rng = np.random.default_rng(3)
    flux = ref_flux[0] + rng.normal(0, 0.01, nobs)
    eflux = np.full(nobs, 0.01)
    eng = Engine(_net(raw), obs=(obs, flux, eflux), b_max=16)
    got = eng.predict_batch(theta_full(th7), stage=2, fwhm_R=True).cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(ref_flux))
    assert np.nanmax(np.abs(got - ref_flux)) <= FLUX_TOL
    L = O.OracleLikelihood(raw, obs, flux, eflux, SPEC_PARS)
    with np.errstate(all="ignore"):
        ref = np.array([L.lnlikefn(t) for t in th7])
    lnl = eng.lnlike_batch(theta_full(th7)).cpu().numpy()
    assert np.array_equal(np.isnan(lnl), np.isnan(ref))
    ok = np.isfinite(ref)
    assert ok.sum() >= B - 2 and np.all(np.abs(lnl[ok] - ref[ok]) <= lnl_tol(ref[ok]))
    # rows reach the post kernel transformed where its geometry is compiled in (n1 = 1024 .. 8192; 8192: twiddles from L2) -- with
    # vsini maps that are not the identity (1000, 3000 pixels) as the transform of the RESAMPLED spectrum, the output layer's weights
    # carrying the resampling as well -- and a 3-layer net hands the records over that say whether every candidate rotates
    assert eng.kernels_used()["rows"] == ("frequency" if npix in (1000, 3000, 8192) else "pixels"), eng.kernels_used()


@pytest.mark.parametrize("npix,nobs,H", [(700, 600, 32), (1500, 1200, 64), (3600, 3200, 300), (6000, 5000, 48)])
def test_rows_of_a_resampled_grid_vs_oracle(Engine, npix, nobs, H):
    """Model grids that are not a power of two long (what a trained network has: Payne/utils/readc3k.py:441-447 builds the grid from
    a range and a resolution; smoothing.py:649-668 resamples it to 2^k): the output layer's restated weights carry the rotation
    stage's static resampling AND its forward transform, the post kernel starts at the taper (n1 = 1024 .. 8192).  A candidate
    that does not rotate needs the un-resampled pixels (ystpred.py:212-224 skips the stage): the thread that writes its record
    says so, and the output layer and the post kernel of that batch run on pixels -- decided on the device, the same launches.
    Against the oracle at the usual tolerances in all three cases (every candidate rotates; one does not; none does), and the
    fall-back gives the pixel variant's numbers to the bit."""
    raw = synth.make_yst_net(npix=npix, lam0=5150.0, R_fwhm=32000.0, H=H, seed=5, line_depth=0.2)
    obs = synth.obs_grid(raw["wavelength"], nobs)
    B = 24
    th7 = synth.draw_candidates(B, seed=npix)
    flux = O.genspec(raw, list(theta_full(th7[0])[0, :8]), outwave=obs)[1] + np.random.default_rng(1).normal(0, 0.01, nobs)
    eflux = np.full(nobs, 0.01)
    L = O.OracleLikelihood(raw, obs, flux, eflux, SPEC_PARS)
    eng = Engine(_net(raw), obs=(obs, flux, eflux), b_max=32)
    pix = Engine(_net(raw), obs=(obs, flux, eflux), b_max=32, variant=_lib.V_ROWS_PIXEL)
    for case in ("all rotate", "one does not", "none does"):
        th = th7.copy()
        if case == "one does not":
            th[3, 5] = 0.0
        if case == "none does":
            th[:, 5] = 0.0
        with np.errstate(all="ignore"):
            ref = np.array([L.lnlikefn(t) for t in th])
            ref_flux = np.array([O.genspec(raw, list(theta_full(t)[0, :8]), outwave=obs)[1] for t in th])
        lnl = eng.lnlike_batch(theta_full(th)).cpu().numpy()
        assert eng.kernels_used()["rows"] == "frequency"
        ok = np.isfinite(ref)
        assert np.array_equal(np.isnan(lnl), np.isnan(ref)) and np.all(np.abs(lnl[ok] - ref[ok]) <= lnl_tol(ref[ok])), case
        got = eng.predict_batch(theta_full(th), stage=2, fwhm_R=True).cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref_flux)) and np.nanmax(np.abs(got - ref_flux)) <= FLUX_TOL, case
        lp = pix.lnlike_batch(theta_full(th)).cpu().numpy()
        assert pix.kernels_used()["rows"] == "pixels"
        if case != "all rotate":                                     # the fall-back IS the pixel path
            assert np.array_equal(np.nan_to_num(lnl), np.nan_to_num(lp))
        else:
            assert np.all(np.abs(lnl[ok] - lp[ok]) <= lnl_tol(ref[ok]))
    # a batch after the fall-back is back on transformed rows (the word carries a sequence number, nothing is reset)
    lnl = eng.lnlike_batch(theta_full(th7)).cpu().numpy()
    ref = np.array([L.lnlikefn(t) for t in th7])
    assert np.all(np.abs(lnl - ref) <= lnl_tol(ref))


@pytest.mark.parametrize("rows", ["default", "pixels"])
def test_nan_and_branch_semantics_vs_oracle(Engine, rows):
    """Inst_R NaN/<=0 -> plain interpolation; Inst_R above the ANN's R -> NaN lnL;
    Vrot = 0 / Vrad = 0 skip their stages; obs outside the model range -> NaN; a NaN label (the network's whole output NaN):
    a flat spectrum out of whichever stage scrubs first (smoothing.py:285 nan_to_num), NaN when none runs."""
    from thepayne_amd import _lib
    raw, obs, flux, eflux = yst_problem("small", H=64)
    eng = Engine(_net(raw), obs=(obs, flux, eflux), b_max=16, variant=0 if rows == "default" else _lib.V_ROWS_PIXEL)
    base = synth.draw_candidates(1, seed=5)[0]
    cases = []
    for vrad, vrot, R, teff in [(0.0, 0.0, 28000.0, None), (12.0, 0.0, np.nan, None), (0.0, 4.0, -1.0, None), (10.0, 2.0, 40000.0, None),
                                (300.0, 1.0, 28000.0, None), (-300.0, 0.0, np.nan, None),
                                (5.0, 3.0, 28000.0, np.nan), (5.0, 0.0, 28000.0, np.nan), (5.0, 0.0, np.nan, np.nan), (0.0, 2.0, np.nan, np.nan)]:
        t = base.copy(); t[4], t[5], t[6] = vrad, vrot, R
        if teff is not None: t[0] = teff
        cases.append(t)
    cases = np.array(cases)
    L = O.OracleLikelihood(raw, obs, flux, eflux, SPEC_PARS)
    with np.errstate(all="ignore"):
        ref = np.array([L.lnlikefn(t) for t in cases])
    lnl = eng.lnlike_batch(theta_full(cases)).cpu().numpy()
    assert np.array_equal(np.isnan(lnl), np.isnan(ref)), (lnl, ref)
    ok = ~np.isnan(ref)
    assert np.all(np.abs(lnl[ok] - ref[ok]) <= lnl_tol(ref[ok]))
    assert np.isnan(ref).sum() >= 3 and (~np.isnan(ref)).sum() >= 3


def test_abi_error_paths(Engine):
    import ctypes as C
    raw, obs, flux, eflux = yst_problem("tiny", H=32)
    eng = Engine(_net(raw), obs=(obs,), b_max=4)
    th = eng.make_theta(8)
    out = eng.torch.empty(8, dtype=eng.torch.float64, device=eng.device)
    rc = eng.lib.payne_lnlike_batch(eng._ctx, th.data_ptr(), 8, out.data_ptr(), None)
    assert rc == -4 and b"b_max" in eng.lib.payne_last_error(eng._ctx)          # PAYNE_E_BATCH
    rc = eng.lib.payne_lnlike_batch(eng._ctx, th.data_ptr(), 4, out.data_ptr(), None)
    assert rc == -1                                                             # no flux bound
    # payne_smooth_direct: "the reference raises ValueError here" (PAYNE_E_SIGMA) is not "malformed call" (PAYNE_E_INVALID)
    from thepayne_amd import _lib
    w = np.linspace(5000.0, 5010.0, 64); sp = np.ones(64); ow = w[8:56].copy(); res = np.empty(len(ow))
    def direct(kind, sig, inres=0.0):
        sig = np.ascontiguousarray(sig, dtype=np.float64)
        return eng.lib.payne_smooth_direct(eng.device.index, kind, w.ctypes.data, sp.ctypes.data, len(w), ow.ctypes.data, len(ow),
                                           sig.ctypes.data, len(sig), float(inres), 0, 10.0, res.ctypes.data)
    assert direct(_lib.SMOOTH_WAVE_DIRECT, [0.05], inres=0.08) == _lib.E_SIGMA    # target sigma below the input's (smoothing.py:381-383)
    assert direct(_lib.SMOOTH_WAVE_DIRECT, np.full(7, 0.3)) == _lib.E_INVALID     # a sigma vector that is neither scalar nor [n]
    assert direct(_lib.SMOOTH_WAVE_DIRECT, [0.3]) == 0 and np.allclose(res, 1.0, atol=1e-12)


@pytest.mark.parametrize("variant", [0, 65536], ids=["default", "global_workspace"])
def test_spectra_larger_than_lds_vs_oracle(Engine, variant):
    """n1 > 16384 takes the kernels for spectra larger than LDS (persistent workgroups): a 40 000-pixel net and the 65 536-pixel
    C5 grid (R ~ 100k); the reference's own demo length, 25 600 pixels (demo/runPayne.py:43-50: n1 = 32 768, resampling maps that
    are not the identity) and a 32 768-pixel R ~ 60k grid (payne_post_chip2_kernel: two candidates at a time, odd batches, a batch
    walked in chunks); a few candidates each (the oracle needs ~0.1 s per evaluation)."""
    for npix, nobs, lam0, R, B in ((40000, 30000, 5150.0, 32000.0, 3), (65536, 60000, 4000.0, 100000.0, 5),
                                   (25600, 20000, 5150.0, 32000.0, 5), (32768, 30000, 4500.0, 60000.0, 5)):
        raw = synth.make_yst_net(npix=npix, lam0=lam0, R_fwhm=R, H=16, seed=31, line_depth=0.1)
        obs = synth.obs_grid(raw["wavelength"], nobs, inset=0.0005, relative=True)
        th7 = synth.draw_candidates(B, seed=npix)
        th7[:, 6] = np.linspace(0.6, 0.85, B) * R                  # instrument R below the ANN's own
        th7[1, 5] = 0.0                                            # one candidate that does not rotate
        rows = [list(theta_full(t)[0, :8]) for t in th7]
        ref_flux = np.array([O.genspec(raw, r, outwave=obs)[1] for r in rows])
        flux = ref_flux[0] + np.random.default_rng(1).normal(0, 0.01, nobs)
        eflux = np.full(nobs, 0.01)
        eng = Engine(_net(raw), obs=(obs, flux, eflux), b_max=2, variant=variant)   # b_max < B: chunked, grid < batch
        got = eng.predict_batch(theta_full(th7), stage=2, fwhm_R=True).cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref_flux))
        assert np.nanmax(np.abs(got - ref_flux)) <= FLUX_TOL, npix
        L = O.OracleLikelihood(raw, obs, flux, eflux, SPEC_PARS)
        ref = np.array([L.lnlikefn(t) for t in th7])
        lnl = eng.lnlike_batch(theta_full(th7)).cpu().numpy()
        assert np.all(np.abs(lnl - ref) <= lnl_tol(ref)), (npix, np.abs(lnl - ref).max())
        # the on-chip stages of the 65 536- and 32 768-point geometric grids take the row as its transform (the output layer's weights restated)
        assert eng.kernels_used()["rows"] == ("frequency" if (npix in (65536, 32768) and variant == 0) else "pixels"), (npix, eng.kernels_used())
        s1 = eng.predict_batch(theta_full(th7[:1]), stage=1).cpu().numpy()[0]
        r1 = O.getspec(raw, Teff=th7[0, 0], logg=th7[0, 1], feh=th7[0, 2], afe=th7[0, 3], rot_vel=th7[0, 5])[1]
        assert np.abs(s1 - r1).max() <= FLUX_TOL


def test_observed_grid_on_the_compute_unit_and_its_fallbacks(Engine):
    """65 536-point spectra, likelihood only: the instrumental stage interpolates onto the observed grid from LDS, half a spectrum at a
    time (chip_conv_obs) -- when the observed wavelengths ascend and all lie inside the candidate's window.  The same pixels in
    descending order, and candidates whose Doppler shift pushes the window's end inside the observed range, take the stage that
    writes the spectrum out and the observed-grid phase behind it: same likelihoods, the oracle's NaNs."""
    npix, nobs, lam0, R, B = 65536, 20000, 4000.0, 100000.0, 6
    raw = synth.make_yst_net(npix=npix, lam0=lam0, R_fwhm=R, H=16, seed=33, line_depth=0.1)
    obs = synth.obs_grid(raw["wavelength"], nobs, inset=0.0002, relative=True)
    th7 = synth.draw_candidates(B, seed=77)
    th7[:, 6] = np.linspace(0.6, 0.85, B) * R
    th7[4, 4], th7[5, 4] = 250.0, -250.0                           # the window's end moves inside the observed range
    flux = O.genspec(raw, list(theta_full(th7[:1])[0, :8]), outwave=obs)[1] + np.random.default_rng(2).normal(0, 0.01, nobs)
    eflux = np.full(nobs, 0.01)
    L = O.OracleLikelihood(raw, obs, flux, eflux, SPEC_PARS)
    with np.errstate(all="ignore"):
        ref = np.array([L.lnlikefn(t) for t in th7])
    assert np.isfinite(ref[:4]).all()
    eng = Engine(_net(raw), obs=(obs, flux, eflux), b_max=B)
    lnl = eng.lnlike_batch(theta_full(th7)).cpu().numpy()
    eng.close()
    assert np.array_equal(np.isnan(lnl), np.isnan(ref)), (lnl, ref)
    ok = np.isfinite(ref)
    assert np.all(np.abs(lnl[ok] - ref[ok]) <= lnl_tol(ref[ok])), np.abs(lnl[ok] - ref[ok]).max()
    eng = Engine(_net(raw), obs=(obs[::-1].copy(), flux[::-1].copy(), eflux[::-1].copy()), b_max=B)
    rev = eng.lnlike_batch(theta_full(th7)).cpu().numpy()
    eng.close()
    assert np.array_equal(np.isnan(rev), np.isnan(ref))
    assert np.all(np.abs(rev[ok] - ref[ok]) <= lnl_tol(ref[ok])), np.abs(rev[ok] - ref[ok]).max()
    # an obs grid this short leaves the record table mostly padding at the block the on-chip loop walks: 2 (24576 - 20000) <= 20000 holds


def test_full_size_properties(Engine):
    """Size-independent checks at the benchmark configuration (C2, B=512)."""
    raw, obs, flux, eflux = yst_problem("C2")
    net = _net(raw)
    eng = Engine(net, obs=(obs, flux, eflux), b_max=512)
    th = theta_full(synth.draw_candidates(512, seed=9))
    lnl = eng.lnlike_batch(th).cpu().numpy()
    # an Inst_R draw above the ANN's own resolution is NaN by contract: rare tail of the prior
    fin = np.isfinite(lnl)
    assert fin.sum() >= 508 and np.array_equal(~fin, th[:, 7] * 2.355 > raw["resolution"])
    th, lnl = th[fin], lnl[fin]
    # chi^2 recomputed on the host from the predicted spectra agrees with the fused reduction
    spec = eng.predict_batch(th, stage=3, fwhm_R=True).cpu().numpy().astype(np.float64)
    chi = -0.5 * (((spec - flux) / eflux) ** 2).sum(axis=1)
    assert np.all(np.abs(chi - lnl) <= 2e-5 * np.abs(lnl) + 5e-3)
    # broadening has unit DC gain: the mean flux is preserved by the vsini stage to ~1e-6
    s0 = eng.predict_batch(th[:64], stage=0).cpu().numpy().astype(np.float64)
    s1 = eng.predict_batch(th[:64], stage=1).cpu().numpy().astype(np.float64)
    assert np.abs(s0[:, 64:-64].mean(1) - s1[:, 64:-64].mean(1)).max() < 2e-4
    # the truth maximises lnL among the draws (the obs spectrum was generated there)
    T = synth.TRUTH
    truth = theta_full(np.array([[T["Teff"], T["logg"], T["feh"], T["afe"], T["vrad"], T["vrot"], T["inst_R"]]]))
    assert eng.lnlike_batch(truth).cpu().numpy()[0] > lnl.max()


def test_c5_at_size(Engine):
    """BASELINE config 5 at its real shape AND its real batch: 65 536 pixels, H = 300, 2048 candidates (eight per persistent
    workgroup of payne_post_chip_kernel).  Eight rows spread over the batch against the oracle (0.2 s each), every row through
    size-independent properties: chi^2 recomputed on the host from the predicted spectra, determinism, independence of
    the position in the batch, the NaN contract of Inst_R above the network's resolution."""
    cfg = synth.CONFIGS["C5"]
    raw = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=300, seed=0)
    obs = synth.obs_grid(raw["wavelength"], cfg["nobs"])
    B = cfg["batch"]
    assert B == 2048
    th7 = synth.draw_candidates(B, seed=55)
    th7[:, 6] = np.linspace(0.55, 0.9, B) * cfg["R"]
    th7[7, 6] = 1.2 * raw["resolution"] / 2.355                       # above the ANN's own resolution: NaN by contract
    th7[2, 5] = 80.0                                                  # rotation whose taper runs past the table (u > 256): the stage with
    #                                                                   its exact re-evaluation, i.e. the general sequence for this candidate
    rows = [list(theta_full(t)[0, :8]) for t in th7[:3]]
    clean = np.array([O.genspec(raw, r, outwave=obs)[1] for r in rows])
    flux = clean[0] + np.random.default_rng(3).normal(0, 0.01, len(obs))
    eflux = np.full(len(obs), 0.01)
    eng = Engine(_net(raw), obs=(obs, flux, eflux), b_max=B)
    th = theta_full(th7)
    lnl = eng.lnlike_batch(th).cpu().numpy()
    assert np.isnan(lnl[7]) and np.isfinite(np.delete(lnl, 7)).all()
    L = O.OracleLikelihood(raw, obs, flux, eflux, SPEC_PARS)
    for k in (0, 1, 2, 200, 255, 256, 1023, 2047):                   # (first and last of a workgroup's walk, both ends of the batch)
        ref = L.lnlikefn(th7[k])
        assert abs(lnl[k] - ref) <= lnl_tol(np.array([ref]))[0], (k, lnl[k], ref)
    spec = eng.predict_batch(th[:3], stage=2, fwhm_R=True).cpu().numpy()
    assert np.nanmax(np.abs(spec - clean)) <= FLUX_TOL
    # chi^2 from the predicted spectra (fp64 on the host) agrees with the fused reduction, every row
    ok = np.isfinite(lnl)
    for s in range(0, B, 64):
        sp = eng.predict_batch(th[s:s + 64], stage=3, fwhm_R=True).cpu().numpy().astype(np.float64)
        chi = -0.5 * (((sp - flux) / eflux) ** 2).sum(axis=1)
        sel = ok[s:s + 64]
        assert np.all(np.abs(chi[sel] - lnl[s:s + 64][sel]) <= 2e-5 * np.abs(lnl[s:s + 64][sel]) + 5e-3)
    # same answers in reverse order (grid-stride walk of the batch) and on a second call
    assert np.array_equal(np.nan_to_num(eng.lnlike_batch(th[::-1].copy()).cpu().numpy()[::-1]), np.nan_to_num(lnl))
    assert np.array_equal(np.nan_to_num(eng.lnlike_batch(th).cpu().numpy()), np.nan_to_num(lnl))


def test_c32k_at_size(Engine):
    """Spectra between the LDS-resident kernel and C5 at a real shape and batch: 32 768 pixels (R ~ 60k), H = 300, 30 000 observed
    pixels, 1024 candidates = 512 pairs over payne_post_chip2_kernel's persistent workgroups.  Rows against the oracle; every row
    against chi^2 recomputed on the host from the predicted spectra; pairs whose two candidates take DIFFERENT paths (no rotation,
    no instrumental smoothing, Inst_R above the network's resolution) beside pairs that take the usual one; order independence."""
    cfg = synth.CONFIGS["C32k"]
    raw = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=300, seed=0)
    obs = synth.obs_grid(raw["wavelength"], cfg["nobs"])
    B = cfg["batch"]
    th7 = synth.draw_candidates(B, seed=77)
    th7[:, 6] = np.linspace(0.55, 0.9, B) * cfg["R"]
    th7[6, 5] = 0.0                                                    # no rotation: its pair (6, 7) goes candidate by candidate
    th7[11, 6] = np.nan                                                # Inst_R absent: plain interpolation (ystpred.py:271-272)
    th7[21, 6] = 1.2 * raw["resolution"] / 2.355                       # above the ANN's own resolution: NaN by contract
    th = theta_full(th7)
    rows = [list(th[k, :8]) for k in (0, 1, 6)]
    clean = np.array([O.genspec(raw, r, outwave=obs)[1] for r in rows])
    flux = clean[0] + np.random.default_rng(3).normal(0, 0.01, len(obs))
    eflux = np.full(len(obs), 0.01)
    eng = Engine(_net(raw), obs=(obs, flux, eflux), b_max=B)
    lnl = eng.lnlike_batch(th).cpu().numpy()
    assert eng.kernels_used()["post"] == "payne_post_chip2_kernel"
    assert np.isnan(lnl[21]) and np.isfinite(np.delete(lnl, 21)).all()
    L = O.OracleLikelihood(raw, obs, flux, eflux, SPEC_PARS)
    for k in (0, 1, 6, 7, 10, 11, 20, 511, 512, 1022, 1023):
        ref = L.lnlikefn(th7[k])
        assert abs(lnl[k] - ref) <= lnl_tol(np.array([ref]))[0], (k, lnl[k], ref)
    spec = eng.predict_batch(th[[0, 1, 6]], stage=2, fwhm_R=True).cpu().numpy()
    assert np.nanmax(np.abs(spec - clean)) <= FLUX_TOL
    ok = np.isfinite(lnl)
    for s in range(0, B, 64):
        sp = eng.predict_batch(th[s:s + 64], stage=3, fwhm_R=True).cpu().numpy().astype(np.float64)
        chi = -0.5 * (((sp - flux) / eflux) ** 2).sum(axis=1)
        sel = ok[s:s + 64]
        assert np.all(np.abs(chi[sel] - lnl[s:s + 64][sel]) <= 2e-5 * np.abs(lnl[s:s + 64][sel]) + 5e-3)
    # the same answers in reverse order (other partners, other workgroups), for an odd batch, and from the workspace kernel
    assert np.array_equal(np.nan_to_num(eng.lnlike_batch(th[::-1].copy()).cpu().numpy()[::-1]), np.nan_to_num(lnl))
    assert np.array_equal(np.nan_to_num(eng.lnlike_batch(th[:333]).cpu().numpy()), np.nan_to_num(lnl[:333]))
    ws = Engine(_net(raw), obs=(obs, flux, eflux), b_max=64, variant=65536)
    l2 = ws.lnlike_batch(th[:64]).cpu().numpy()
    assert ws.kernels_used()["post"] == "payne_post_big_kernel"
    assert np.all(np.abs(l2 - lnl[:64])[ok[:64]] <= lnl_tol(lnl[:64][ok[:64]]))


@pytest.mark.parametrize("variant", [0, 8192])
def test_c3_at_size(Engine, variant):
    """BASELINE config 3 at its real shape: C2's 4096-pixel 2 x 300 network and 3600 observed pixels PLUS
    photometry in seven filters (6-64-64-1 sigmoid nets, observed magnitudes 5.0 +- 0.05, the log(A) parametrisation),
    512 candidates.  Six rows against the oracle's joint likelihood (likelihood.py:84-117); every row against chi^2_spec
    recomputed on the host from the predicted spectra + chi^2_sed recomputed from the oracle's magnitudes
    (predictsed.py:75-103 in fp64, a few ms per row).  variant 8192 = the photometric nets as a launch of their own
    (PAYNE_V_SED_OWN_LAUNCH) instead of extra workgroups of the hidden-layer launch."""
    cfg = synth.CONFIGS["C3"]
    raw = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], seed=0)
    obs = synth.obs_grid(raw["wavelength"], cfg["nobs"])
    from thepayne_amd.engine import highav_coefficients
    phot = synth.make_phot_nets()
    phot["hiav"] = highav_coefficients(phot["filters"])       # highred.py's table, as the oracle's sed reads it
    obs_phot = synth.c3_obs_phot(phot["filters"])
    B = cfg["batch"]
    th9 = synth.draw_candidates_c3(B, seed=31)
    th9[5, 8] = 5.5                                          # one candidate in the high-Av branch (predictsed.py:86-90)
    T = synth.TRUTH
    row = [T["Teff"], T["logg"], T["feh"], T["afe"], T["vrad"], T["vrot"], np.nan, T["inst_R"]]
    clean = O.genspec(raw, row, outwave=obs)[1]
    flux = clean + np.random.default_rng(0).normal(0, 0.01, len(obs))
    eflux = np.full(len(obs), 0.01)
    eng = Engine(_net(raw), obs=(obs, flux, eflux), phot=phot, obs_phot=obs_phot, photscale=True, b_max=B, variant=variant)
    th = np.full((B, eng.ncols), np.nan)
    th[:, 0:6] = th9[:, 0:6]
    th[:, 7] = th9[:, 6]
    th[:, eng.phot_off] = th9[:, 7]
    th[:, eng.phot_off + 2] = th9[:, 8]
    lnl = eng.lnlike_batch(th).cpu().numpy()
    names = SPEC_PARS + ['log(A)', 'Av']
    L = O.OracleLikelihood(raw, obs, flux, eflux, names, phot=phot, obs_phot=obs_phot, photscale=True)
    for k in (0, 1, 2, 5, 300, 511):
        ref = L.lnlikefn(th9[k])
        assert np.isnan(lnl[k]) if np.isnan(ref) else abs(lnl[k] - ref) <= lnl_tol(np.array([ref]))[0], (k, lnl[k], ref)
    # all 512: chi^2_spec from the predicted spectra (fp64 on the host) + chi^2_sed from the oracle's magnitudes
    mo = np.array([v[0] for v in obs_phot.values()]); me = np.array([v[1] for v in obs_phot.values()])
    chi_sed = np.empty(B)
    for k in range(B):
        mags = np.atleast_1d(O.genphot_scaled(phot, [th9[k, 0], th9[k, 1], th9[k, 2], th9[k, 3], th9[k, 7], th9[k, 8], None]))
        chi_sed[k] = np.sum(((mags - mo) ** 2) / me ** 2)
    mg = eng.sed_batch(np.column_stack([np.log10(th9[:, 0]), th9[:, 1], th9[:, 2], th9[:, 3], th9[:, 8], np.full(B, 3.1),
                                        np.full(B, np.nan), np.full(B, np.nan), th9[:, 7]])).cpu().numpy()
    assert np.allclose(np.sum(((mg - mo) ** 2) / me ** 2, axis=1), chi_sed, rtol=1e-9, atol=1e-9)
    ok = np.isfinite(lnl)
    assert ok.sum() >= B - 4
    for s in range(0, B, 128):
        sp = eng.predict_batch(th[s:s + 128], stage=3, fwhm_R=True).cpu().numpy().astype(np.float64)
        chi = -0.5 * ((((sp - flux) / eflux) ** 2).sum(axis=1) + chi_sed[s:s + 128])
        sel = ok[s:s + 128]
        assert np.all(np.abs(chi[sel] - lnl[s:s + 128][sel]) <= 2e-5 * np.abs(lnl[s:s + 128][sel]) + 5e-3)
    assert np.array_equal(np.nan_to_num(eng.lnlike_batch(th[::-1].copy()).cpu().numpy()[::-1]), np.nan_to_num(lnl))


def test_row_domains_switch_within_one_context(Engine):
    """One context, calls in turn that hand the post kernel rows in the frequency domain (likelihood, getspec), in pixels
    (predictspec; spectra supplied by the caller; everything while a continuum network is bound) and back: every call gives what a
    fresh context gives for it, and payne_last_kernel(kind 4) names the domain each one ran in."""
    from thepayne_amd import nnio
    raw, obs, flux, eflux = yst_problem("small", H=64)
    th = theta_full(synth.draw_candidates(12, seed=77))
    th[3, 5] = 0.0                                                  # one candidate that does not rotate

    def fresh():
        return Engine(_net(raw), obs=(obs, flux, eflux), b_max=16)
    ref_l = fresh().lnlike_batch(th).cpu().numpy()
    ref_2 = fresh().predict_batch(th, stage=2, fwhm_R=True).cpu().numpy()
    ref_0 = fresh().predict_batch(th, stage=0).cpu().numpy()
    sp = ref_0.astype(np.float32)
    ref_s = fresh().smooth_batch(sp, th, stage=2, fwhm_R=True).cpu().numpy()
    eng = fresh()
    for _ in range(2):
        assert np.array_equal(eng.lnlike_batch(th).cpu().numpy(), ref_l, equal_nan=True) and eng.kernels_used()["rows"] == "frequency"
        assert np.array_equal(eng.predict_batch(th, stage=0).cpu().numpy(), ref_0) and eng.kernels_used()["rows"] == "pixels"
        assert np.array_equal(eng.predict_batch(th, stage=2, fwhm_R=True).cpu().numpy(), ref_2, equal_nan=True) and eng.kernels_used()["rows"] == "frequency"
        assert np.array_equal(eng.smooth_batch(sp, th, stage=2, fwhm_R=True).cpu().numpy(), ref_s, equal_nan=True) and eng.kernels_used()["rows"] == "pixels"
    # the same spectra through the network (frequency rows) and handed over as pixels (its own predictspec output): the same getspec
    assert np.nanmax(np.abs(ref_s - ref_2)) <= FLUX_TOL
    # a continuum network multiplies pixel by pixel: pixel rows while it is bound, frequency rows again once it is removed
    cnet = nnio.normalize_spec_net(synth.make_cont_net(npix=300, lam_lo=raw["wavelength"][0] - 2.0, lam_hi=raw["wavelength"][-1] + 2.0, H=16), "YST1")
    eng.set_continuum(cnet)
    with_c = eng.lnlike_batch(th).cpu().numpy()
    assert eng.kernels_used()["rows"] == "pixels" and not np.array_equal(with_c, ref_l, equal_nan=True)
    eng.set_continuum(None)
    assert np.array_equal(eng.lnlike_batch(th).cpu().numpy(), ref_l, equal_nan=True) and eng.kernels_used()["rows"] == "frequency"
