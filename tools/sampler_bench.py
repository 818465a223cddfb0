"""End-to-end throughput of the batched nested sampler on the C2 problem (GPU box):
likelihood calls per second as the sampler sees them, host proposals vs device proposals.

  python tools/sampler_bench.py [--maxcall 400000] [--nlive 512] [--walks 25]

The synthetic spectrum is produced with the engine itself (no oracle: this is a timing tool).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from thepayne_amd import synth, nnio                                  # noqa: E402
from thepayne_amd.fitting.likelihood import likelihood               # noqa: E402
from thepayne_amd.fitting.prior import prior                         # noqa: E402
from thepayne_amd.fitting.fitstar import lnprob_batch                # noqa: E402
from thepayne_amd.sampler import NestedSampler                       # noqa: E402

SPEC = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Inst_R']
ALL = ['Teff', 'log(g)', '[Fe/H]', '[a/Fe]', 'Vrad', 'Vrot', 'Vmic', 'Inst_R', 'log(R)', 'Dist', 'log(A)', 'Av', 'Rv',
       'CarbonScale']


def make_problem(config="C2", nlive=512, variant=0):
    """(likelihood, prior) of the synthetic fit the end-to-end numbers are quoted on: the config's network and observed grid,
    a noisy spectrum of synth.TRUTH made with the engine itself, the demo's prior box."""
    cfg = synth.CONFIGS["C2" if config == "LinNet300" else config]
    joint = bool(cfg.get("phot"))
    tmp = tempfile.mkdtemp()
    nntype = 'YST1'
    if config == "LinNet300":                 # FitPayne's default network (fitstar.py:81) at full width, in the reference's file layout
        nntype = 'LinNet'
        raw = synth.make_torch_net("LinNet", npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=(300, 300, 300), seed=21)
        path = os.path.join(tmp, "linnet.npz")
        d = {("model/" + k if k.startswith("lin") else k): v for k, v in raw.items() if k != "kind"}
        d["wavelengths"] = d.pop("wavelength")
        d["resolution"] = np.array([d["resolution"]])
        nnio.save_npz(path, d)
    else:
        raw = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=300, seed=0, line_depth=0.3)
        path = os.path.join(tmp, "yst.npz")
        nnio.save_npz(path, {k: (np.array([v]) if k == "resolution" else v) for k, v in raw.items() if k != "kind"})
    obs = synth.obs_grid(raw["wavelength"], cfg["nobs"])
    free = SPEC + (['log(A)', 'Av'] if joint else [])
    fitpars = [list(ALL), {p: p in free for p in ALL}]
    rb = [True, joint, False, joint, False]                 # spec, phot, modpoly, photscale, carbon
    fitargs = {'obs_wave_fit': obs, 'obs_flux_fit': np.ones(len(obs)), 'obs_eflux_fit': np.full(len(obs), 0.01),
               'specANNpath': path, 'NNtype': nntype, 'fixedpars': {}}
    if joint:
        phot = synth.make_phot_nets()
        fitargs.update({'photANNpath': phot, 'obs_phot': synth.c3_obs_phot(phot["filters"])})
    L = likelihood(fitargs, fitpars, rb, b_max=nlive, verbose=False)
    T = synth.TRUTH
    truth = np.array([[T["Teff"], T["logg"], T["feh"], T["afe"], T["vrad"], T["vrot"], T["inst_R"]] + ([0.0, 0.1] if joint else [])])
    clean = L.GM.engine.predict_batch(L.theta_rows(truth), stage=3, fwhm_R=True).cpu().numpy()[0].astype(np.float64)
    L.GM.engine.close()
    fitargs['obs_flux_fit'] = clean + np.random.default_rng(0).normal(0, 0.01, len(obs))
    L = likelihood(fitargs, fitpars, rb, b_max=nlive, verbose=False, variant=variant)
    P = prior(fitargs, synth.c3_priordict() if joint else synth.demo_priordict(), fitpars, rb)
    return L, P


def run(config="C2", maxcall=300000, nlive=512, walks=25, modes=("host", "device", "device_chunks", "device2_chunks"),
        verbose=False, bound='multi', variant=0, seed=1, dlogz=0.01, complete=False):
    """Likelihood calls per second as the nested sampler sees them (prior transform, proposals, transfers,
    bookkeeping included).  Returns {mode: {...}}.  config 'C3' = C2 + photometry in seven filters (joint fit, photscale).
    `seed`: the sampler's random stream; `dlogz`: stopping threshold (tiny: the run ends at `maxcall`); `complete`: the run is a whole
    fit (maxcall=None, dlogz as the reference's delta_logz_final) and the remaining live points are added inside the timed region.
    Modes: "device_chunks" = proposals on the device, the sampler's default loop (the turn between two queues on the device where the
    proposer offers it); "..._hostturn" = the turn made on the host, queues launched ahead; "..._serial" = every queue launched after
    the one before is consumed; "..._devturn" = pipeline='device' asked for by name."""
    L, P = make_problem(config, nlive, variant)
    out = {}
    for mode in modes:
        proposer = None
        queue = nlive
        kw = {}
        if mode.endswith("_serial"):                        # "device_chunks_serial": every queue launched after the one before is consumed
            kw["pipeline"] = False
            mode_ = mode[:-len("_serial")]
        elif mode.endswith("_devturn"):                     # "device_chunks_devturn": the turn between two queues made on the device
            kw["pipeline"] = 'device'
            mode_ = mode[:-len("_devturn")]
        elif mode.endswith("_hostturn"):                    # "device_chunks_hostturn": the turn on the host, the next queue launched ahead
            kw["pipeline"] = True
            mode_ = mode[:-len("_hostturn")]
        else:
            mode_ = mode
        if mode.startswith("device2"):                      # two chain populations in flight
            from thepayne_amd.sampler.device import MultiPopProposer
            proposer = MultiPopProposer(L, P, k_max=nlive, n_pop=2)
            queue = 2 * nlive
        elif mode.startswith("device"):
            from thepayne_amd.sampler.device import DeviceProposer
            proposer = DeviceProposer(L, P, k_max=nlive)
        S = NestedSampler(lnprob_batch, P.priortrans_batch, L.ndim, logl_args=[L, P], nlive=nlive, bound=bound,
                          sample='rwalk', walks=walks, batched=True, queue_size=queue,
                          rstate=np.random.default_rng(seed), proposer=proposer, **kw)
        t0 = time.perf_counter()
        c0 = S.ncall
        nell = 1
        if mode_.endswith("chunks"):
            for _ in S.sample_chunks(maxcall=maxcall, dlogz=dlogz):
                nell = max(nell, len(S._ells))
        else:
            for _ in S.sample(maxcall=maxcall, dlogz=dlogz):
                pass
        if complete:                                        # a whole fit: + the remaining live points (fitstar.py:410)
            for _ in S.add_live_points():
                pass
        dt = time.perf_counter() - t0
        out[mode] = {"calls": int(S.ncall - c0), "iterations": int(S.it - 1), "seconds": round(dt, 4),
                     "logzerr": float(np.sqrt(max(S.logzvar, 0.0))),
                     "evals_per_s": round((S.ncall - c0) / dt), "logz": float(S.logz), "scale": float(S.scale), "max_ellipsoids": nell,
                     **({"resyncs": int(S._dev_desync)} if getattr(S, "_dev_turn", False) else {})}
        if verbose:
            print(mode, json.dumps(out[mode]), flush=True)
        if proposer is not None:
            proposer.close()
    L.GM.engine.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--maxcall", type=int, default=400000)
    ap.add_argument("--nlive", type=int, default=512)
    ap.add_argument("--walks", type=int, default=25)
    ap.add_argument("--modes", default="host,device,device_chunks,device2_chunks")
    ap.add_argument("--bound", default="multi")
    ap.add_argument("--variant", type=int, default=0, help="payne_opts.variant (PAYNE_V_* bits)")
    ap.add_argument("--dlogz", type=float, default=0.01, help="stopping threshold (tiny: the run ends at --maxcall)")
    a = ap.parse_args()
    out = run(a.config, a.maxcall, a.nlive, a.walks, tuple(a.modes.split(",")), verbose=True, bound=a.bound, variant=a.variant, dlogz=a.dlogz)
    print(json.dumps({"sampler_bench": out, "config": a.config, "nlive": a.nlive, "walks": a.walks}))


if __name__ == "__main__":
    main()
