"""Mock solar demo: the counterpart of the reference's demo/runPayne.py on synthetic networks.

The reference script fits a UVES spectrum (+ photometry) of the Sun / Procyon with ANN files
that live outside its repository.  Neither the spectra nor the networks are available, so this
demo builds a synthetic 2 x 300 network (the construction of SURVEY section 8(d)), observes a
solar-like star with it (Teff 5770, log g 4.44, [Fe/H] 0 -- the point the reference script
announces for its mock run, runPayne.py:17-19; obs_eflux = flux / 25, runPayne.py:50) and hands
FitPayne the same `inputdict` the reference script assembles (runPayne.py:38-150): every
likelihood call of the nested sampler is then a batch on the GPU.

    python demo/run_mock_sun.py [--phot] [--dynamic] [--npix 4096] [--npoints 125] [--out demo_sun.dat]
"""
import argparse
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from thepayne_amd import synth, nnio                      # noqa: E402
from thepayne_amd.fitting import fitstar                  # noqa: E402
from thepayne_amd.fitting.genmod import GenMod            # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--phot", action="store_true", help="joint fit with seven synthetic broad-band magnitudes")
    ap.add_argument("--dynamic", action="store_true", help="samplertype 'Dynamic' instead of 'Static'")
    ap.add_argument("--host-turn", action="store_true",
                    help="sampler['pipeline'] = 'host': the turn between two proposal queues on the host (the default makes it on the GPU)")
    ap.add_argument("--device-turn", action="store_true",
                    help="sampler['pipeline'] = 'device': the live set on the GPU, the turn between two proposal queues made there (static sampler)")
    ap.add_argument("--npix", type=int, default=4096)
    ap.add_argument("--npoints", type=int, default=125)
    ap.add_argument("--out", default="demo_sun.dat")
    a = ap.parse_args()

    print('mock solar fit on synthetic networks: Teff 5770, log g 4.44, [Fe/H] 0, Vrad 10, Vrot 3, R 28800')
    print('spectrum: yes   photometry: {0}   sampler: {1}'.format('yes' if a.phot else 'no', 'Dynamic' if a.dynamic else 'Static'))

    # the networks: files in the reference's key layout (npz instead of HDF5: no h5py here)
    tmp = tempfile.mkdtemp()
    net = synth.make_yst_net(npix=a.npix, H=300, seed=0, line_depth=0.3)
    annpath = os.path.join(tmp, "synthANN.npz")
    nnio.save_npz(annpath, {k: (np.array([v]) if k == "resolution" else v) for k, v in net.items() if k != "kind"})
    T = synth.TRUTH
    truth = [T["Teff"], T["logg"], T["feh"], T["afe"], T["vrad"], T["vrot"], np.nan, T["inst_R"]]

    # the mock observation, made by the model itself
    GM = GenMod()
    GM._initspecnn(nnpath=annpath, NNtype='YST1')
    obs_wave = synth.obs_grid(net["wavelength"], int(0.88 * a.npix))
    _, flux = GM.genspec(truth, outwave=obs_wave)
    flux = np.asarray(flux, dtype=np.float64)
    eflux = flux / 25.0
    flux = flux + np.random.default_rng(0).normal(0.0, 1.0, len(flux)) * eflux

    inputdict = {'spec': {}, 'specANNpath': annpath, 'NNtype': 'YST1'}
    inputdict['spec']['obs_wave'] = obs_wave
    inputdict['spec']['obs_flux'] = flux
    inputdict['spec']['obs_eflux'] = eflux
    inputdict['spec']['normspec'] = False
    inputdict['spec']['convertair'] = False

    if a.phot:
        phot = synth.make_phot_nets()
        GM._initphotnn(phot["filters"], nnpath=phot)
        mags = GM.genphot_scaled([T["Teff"], T["logg"], T["feh"], T["afe"], 0.0, 0.5, None])
        inputdict['phot'] = {fn: [float(m), 0.05] for fn, m in mags.items()}
        inputdict['photANNpath'] = phot
        inputdict['photscale'] = True

    # sampler and priors: the reference demo's settings (runPayne.py:110-141)
    inputdict['sampler'] = {'samplertype': 'Dynamic' if a.dynamic else 'Static', 'samplerbounds': 'multi',
                            'samplemethod': 'rwalk', 'npoints': a.npoints, 'flushnum': 100,
                            'delta_logz_final': 0.1, 'bootstrap': 0, 'walks': 25, 'maxbatch': 4}
    if a.device_turn and not a.dynamic:
        inputdict['sampler']['pipeline'] = 'device'
    if a.host_turn and not a.dynamic:
        inputdict['sampler']['pipeline'] = 'host'
    inputdict['priordict'] = synth.demo_priordict()
    inputdict['priordict']['Teff'] = {'pv_uniform': [5000.0, 6500.0]}
    if a.phot:
        inputdict['priordict']['log(A)'] = {'pv_uniform': [-1.0, 1.0]}
        inputdict['priordict']['Av'] = {'pv_uniform': [0.0, 1.0]}
    inputdict['output'] = a.out

    FS = fitstar.FitPayne()
    print('priors:')
    for kk, vv in inputdict['priordict'].items():
        print('   {0:>8s}  {1}'.format(kk, vv))
    sys.stdout.flush()
    result = FS.run(inputdict=inputdict)
    summ = result.summary()
    print('log(Z) = {0:.3f} +/- {1:.3f}   ({2:d} samples, {3:d} likelihood calls)'.format(
        summ[0], summ[1], int(summ[2]), int(summ[3])))
    for i, name in enumerate(FS.likeobj.fitpars_i):
        mean, std = summ[5 + 5 * i], summ[6 + 5 * i]
        print('  {0:>8s} = {1:12.4f} +/- {2:.4f}'.format(name, mean, std))
    return result


if __name__ == '__main__':
    main()
