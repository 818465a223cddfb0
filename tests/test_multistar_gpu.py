"""C4 of BASELINE.json on the driver's one-GPU box: the multi-star job (tools/fit_stars.py -- full FitPayne fits
sharded over ranks, one gather of posterior summaries) with 1 rank and with 2 ranks sharing the GPU.  RCCL refuses
two ranks on one device, so the 2-rank launch gathers over gloo (thepayne_amd/dist.py:33-37) while both ranks
compute on cuda:0; on an 8-GPU node the same script gathers over RCCL."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, nproc, tag):
    out = str(tmp_path / ("table_%s.npy" % tag))
    args = [os.path.join(ROOT, "tools", "fit_stars.py"), "--stars", "4", "--npix", "1024", "--npoints", "128",
            "--dlogz", "0.5", "--out", out]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if nproc == 1:
        cmd = [sys.executable] + args
    else:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    return np.load(out), res.stdout


def test_fit_stars_one_rank_and_two_ranks_agree(tmp_path):
    one, log1 = _run(tmp_path, 1, "one")
    two, log2 = _run(tmp_path, 2, "two")
    assert one.shape == two.shape == (4, 5 + 5 * 7)
    assert "4 stars on 1 rank(s)" in log1 and "4 stars on 2 rank(s)" in log2
    assert np.all(np.isfinite(one)) and np.all(np.isfinite(two))
    # a star's fit depends on its own seed only: which rank ran it does not matter
    np.testing.assert_allclose(one, two, rtol=1e-9, atol=1e-9)
    # ... and the one line an 8-GPU run is compared by says so
    import re
    ck = [re.search(r"table checksum (\w+)", log).group(1) for log in (log1, log2)]
    assert ck[0] == ck[1], ck
    # each star recovers its own truth (Teff = 5770 + 25 i): posterior mean within 6 sigma
    for i, row in enumerate(one):
        assert abs(row[5] - (5770.0 + 25.0 * i)) < 6.0 * row[6] + 10.0, (i, row[5], row[6])


def test_a_finished_fits_context_serves_the_next_fit_of_the_same_network(tmp_path, monkeypatch):
    """Fits of a multi-star job share their network: the context of a finished fit (weights on the device, the output layer restated
    in the frequency domain, tables) is kept and the next fit only binds its own spectrum (fitting/genmod.py).  The second star's
    fit on a re-used context must be the fit on a fresh one to the last bit, the context must really be re-used, a context still
    held by a live sampler must not be handed out, and another network / batch size must not take it."""
    import numpy as np
    from thepayne_amd import synth, nnio
    from thepayne_amd.fitting import genmod
    from thepayne_amd.fitting.fitstar import FitPayne
    net = synth.make_yst_net(npix=1024, H=300, seed=0, line_depth=0.3)
    ann = str(tmp_path / "ann.npz")
    nnio.save_npz(ann, {k: (np.array([v]) if k == "resolution" else v) for k, v in net.items() if k != "kind"})
    obs = synth.obs_grid(net["wavelength"], 900)
    built = []
    real_new = genmod.GenMod.new_engine
    monkeypatch.setattr(genmod.GenMod, "new_engine", lambda self: built.append(1) or real_new(self))
    genmod.drop_idle_engines()
    GM = genmod.GenMod(device=0)
    GM._initspecnn(nnpath=ann, NNtype='YST1')
    T = synth.TRUTH

    def star(i):
        truth = [T["Teff"] + 25.0 * i, T["logg"], T["feh"], T["afe"], T["vrad"] + 0.1 * i, T["vrot"], np.nan, T["inst_R"]]
        _, clean = GM.genspec(truth, outwave=obs)
        return np.asarray(clean) + np.random.default_rng(1000 + i).normal(0, 0.01, len(obs))
    fluxes = [star(i) for i in range(5)]             # (made first: the spectra's own context is not what is counted below)

    def fit(i, npoints=128, keep=None):
        flux = fluxes[i]
        inputdict = {'spec': {'obs_wave': obs, 'obs_flux': flux, 'obs_eflux': np.full(len(obs), 0.01), 'convertair': False},
                     'specANNpath': ann, 'NNtype': 'YST1',
                     'sampler': {'samplertype': 'Static', 'samplerbounds': 'multi', 'samplemethod': 'rwalk', 'npoints': npoints,
                                 'walks': 25, 'delta_logz_final': 0.5, 'flushnum': 10 ** 9, 'seed': i},
                     'priordict': synth.demo_priordict(), 'output': str(tmp_path / ("star_%d.dat" % i))}
        F = FitPayne(device=0)
        S = F.run(inputdict=inputdict, verbose=False)
        if keep is not None:
            keep.append((F, S))
        return S.summary(), open(inputdict['output']).read()

    n0 = len(built)
    a1, _ = fit(0)                                   # builds a context ...
    assert len(built) == n0 + 1 and sum(len(v) for v in genmod._ENGINE_POOL.values()) == 1      # ... which waits for the next fit
    b1, rows1 = fit(1)                               # ... and serves it
    assert len(built) == n0 + 1
    genmod.drop_idle_engines()
    b2, rows2 = fit(1)                               # the same star on a context of its own
    assert len(built) == n0 + 2
    assert np.array_equal(b1, b2) and rows1 == rows2
    assert abs(a1[5] - 5770.0) < 6.0 * a1[6] + 10.0 and abs(b1[5] - 5795.0) < 6.0 * b1[6] + 10.0
    # a fit whose sampler is still alive keeps its context to itself
    alive = []
    fit(2, keep=alive)
    assert sum(len(v) for v in genmod._ENGINE_POOL.values()) == 0
    fit(3)
    assert len(built) == n0 + 3                      # (star 2 took the idle one of the `b2` fit, star 3 had to build)
    del alive[:]
    # another batch size is another context
    fit(4, npoints=64)
    assert len(built) == n0 + 4
    genmod.drop_idle_engines()
    assert not genmod._ENGINE_POOL
