"""C4 of BASELINE.json: a batch of independent synthetic stars, one fit per rank at a time, posterior summaries
gathered with ONE collective (RCCL over xGMI when launched with torchrun on GPUs; gloo on CPU ranks).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/fit_stars.py --stars 8
    python tools/fit_stars.py --stars 2                       # single process: the same code path without a group

Every star is the SURVEY 8(d) solar-like mock with its own noise seed and a small truth jitter; every rank prints
nothing, rank 0 prints one line per star (truth, posterior mean +/- std of Teff, ln Z, calls) and the wall time."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from thepayne_amd import dist as pdist, synth, nnio       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stars", type=int, default=8)
    ap.add_argument("--npix", type=int, default=4096)
    ap.add_argument("--npoints", type=int, default=512)
    ap.add_argument("--dlogz", type=float, default=0.1)
    ap.add_argument("--out", default=None, help="rank 0 saves the gathered [stars, summary] table here (.npy)")
    a = ap.parse_args()
    rank, world, local_rank = pdist.init_from_env()
    from thepayne_amd.fitting.fitstar import FitPayne
    from thepayne_amd.fitting.genmod import GenMod
    tmp = tempfile.mkdtemp()
    net = synth.make_yst_net(npix=a.npix, H=300, seed=0, line_depth=0.3)
    annpath = os.path.join(tmp, "ann_%d.npz" % rank)
    nnio.save_npz(annpath, {k: (np.array([v]) if k == "resolution" else v) for k, v in net.items() if k != "kind"})
    obs = synth.obs_grid(net["wavelength"], int(0.88 * a.npix))
    GM = GenMod(device=local_rank)
    GM._initspecnn(nnpath=annpath, NNtype='YST1')
    T = synth.TRUTH

    def truth_of(i):
        return [T["Teff"] + 25.0 * i, T["logg"], T["feh"], T["afe"], T["vrad"] + 0.1 * i, T["vrot"], np.nan, T["inst_R"]]

    def fit(star, i):
        _, clean = GM.genspec(truth_of(i), outwave=obs)
        flux = np.asarray(clean) + np.random.default_rng(1000 + i).normal(0, 0.01, len(obs))
        inputdict = {
            'spec': {'obs_wave': obs, 'obs_flux': flux, 'obs_eflux': np.full(len(obs), 0.01), 'convertair': False},
            'specANNpath': annpath, 'NNtype': 'YST1',
            'sampler': {'samplertype': 'Static', 'samplerbounds': 'multi', 'samplemethod': 'rwalk', 'npoints': a.npoints,
                        'walks': 25, 'delta_logz_final': a.dlogz, 'flushnum': 10 ** 9, 'seed': i},
            'priordict': synth.demo_priordict(), 'output': os.path.join(tmp, "star_%d.dat" % i)}
        F = FitPayne(device=local_rank)
        return F.run(inputdict=inputdict, verbose=False).summary()

    t0 = time.perf_counter()
    table = pdist.fit_stars(list(range(a.stars)), fit, summary_length=5 + 5 * 7)
    dt = time.perf_counter() - t0
    if rank == 0:
        if a.out:
            np.save(a.out, table)
        for i, row in enumerate(table):
            print("star %2d  Teff truth %7.1f  fit %7.1f +/- %5.1f   lnZ %10.2f +/- %.2f   %7d calls"
                  % (i, truth_of(i)[0], row[5], row[6], row[0], row[1], int(row[3])))
        print("%d stars on %d rank(s): %.2f s, %.2f M likelihood calls/s in all"
              % (a.stars, world, dt, table[:, 3].sum() / dt / 1e6))
        # a star's fit depends on its own seed only: the table of an 8-GPU run is the table of a 1-GPU run -- one comparison
        import hashlib
        print("table checksum %s (sha256 of the values at 6 decimals; to the bit: %s)  sum lnZ %.6f  calls %d"
              % (hashlib.sha256(np.round(table, 6).tobytes()).hexdigest()[:16], hashlib.sha256(table.tobytes()).hexdigest()[:16],
                 table[:, 0].sum(), int(table[:, 3].sum())))
    pdist.finalize()


if __name__ == "__main__":
    main()
