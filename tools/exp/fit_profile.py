#!/usr/bin/env python
"""Where does the wall time of ONE complete FitPayne fit go on the GPU box?  (tools/fit_stars.py: 8 stars in 0.99 s = 124 ms a star,
of which the sampling loop is ~35 ms.)  Fits three stars like tools/fit_stars.py and prints cProfile's view of the third.

    python tools/exp/fit_profile.py [--npix 4096] [--npoints 512]
"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from thepayne_amd import synth, nnio  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--npix", type=int, default=4096)
ap.add_argument("--npoints", type=int, default=512)
ap.add_argument("--dlogz", type=float, default=0.01)
a = ap.parse_args()
from thepayne_amd.fitting.fitstar import FitPayne  # noqa: E402
from thepayne_amd.fitting.genmod import GenMod  # noqa: E402
tmp = tempfile.mkdtemp()
net = synth.make_yst_net(npix=a.npix, H=300, seed=0, line_depth=0.3)
annpath = os.path.join(tmp, "ann.npz")
nnio.save_npz(annpath, {k: (np.array([v]) if k == "resolution" else v) for k, v in net.items() if k != "kind"})
obs = synth.obs_grid(net["wavelength"], int(0.88 * a.npix))
GM = GenMod(device=0)
GM._initspecnn(nnpath=annpath, NNtype='YST1')
T = synth.TRUTH


def fit(i):
    truth = [T["Teff"] + 25.0 * i, T["logg"], T["feh"], T["afe"], T["vrad"] + 0.1 * i, T["vrot"], np.nan, T["inst_R"]]
    _, clean = GM.genspec(truth, outwave=obs)
    flux = np.asarray(clean) + np.random.default_rng(1000 + i).normal(0, 0.01, len(obs))
    inputdict = {'spec': {'obs_wave': obs, 'obs_flux': flux, 'obs_eflux': np.full(len(obs), 0.01), 'convertair': False},
                 'specANNpath': annpath, 'NNtype': 'YST1',
                 'sampler': {'samplertype': 'Static', 'samplerbounds': 'multi', 'samplemethod': 'rwalk', 'npoints': a.npoints,
                             'walks': 25, 'delta_logz_final': a.dlogz, 'flushnum': 10 ** 9, 'seed': i},
                 'priordict': synth.demo_priordict(), 'output': os.path.join(tmp, "star_%d.dat" % i)}
    F = FitPayne(device=0)
    return F.run(inputdict=inputdict, verbose=False).summary()


for i in range(2):
    t0 = time.perf_counter(); fit(i); print("star %d: %.1f ms" % (i, 1e3 * (time.perf_counter() - t0)))
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable(); fit(2); pr.disable()
print("star 2 (profiled): %.1f ms" % (1e3 * (time.perf_counter() - t0)))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(38)
print("\n".join(l[:170] for l in s.getvalue().splitlines()[:70]))
