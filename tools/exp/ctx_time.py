"""Time payne_ctx_create (PayneEngine construction) for the C2 YST1 net and LinNet 5 x 300 (GPU box)."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from thepayne_amd import synth, nnio, _lib
from thepayne_amd.engine import PayneEngine
cfg = synth.CONFIGS["C2"]
raw = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=300, seed=0)
obs = synth.obs_grid(raw["wavelength"], cfg["nobs"])
nets = {"YST1": nnio.normalize_spec_net(raw),
        "LinNet300": nnio.normalize_spec_net(synth.make_torch_net("LinNet", npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], H=(300, 300, 300), seed=21), "LinNet")}
for name, net in nets.items():
    for v in (0, _lib.V_OUT_BF16X3 | _lib.V_HID_F32):
        ts = []
        for rep in range(4):
            t0 = time.perf_counter()
            e = PayneEngine(net, obs=(obs, np.ones(len(obs)), np.full(len(obs), 0.01)), b_max=512, variant=v)
            ts.append(time.perf_counter() - t0)
            e.close()
        print(name, "variant", v, "create ms", [round(1e3 * t, 1) for t in ts])
