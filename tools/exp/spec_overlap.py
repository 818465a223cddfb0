"""Would the hidden layers of the NEXT walk step (both outcomes: 2 x 512 rows) hide beside the output layer and the post kernel of the
current one?  (Next 2.)  A proxy that needs no new kernels: engine A runs the C2 likelihood step on one stream; engine B
-- the same hidden layers, 1024 rows, but a 32-pixel output layer and no observed grid, so that its launch IS its hidden-layer kernel
(+ a 1024 x 32 output GEMM of ~2 us) -- runs once per A-step on a second stream.  Printed: A's step alone, B's launch alone, both
together per A-step.  Overlap would show as (together - A alone) << B alone."""
import ctypes as C
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from thepayne_amd import synth, nnio
from thepayne_amd.engine import PayneEngine
from helpers import theta_full

cfg = synth.CONFIGS["C2"]
raw = synth.make_yst_net(npix=cfg["npix"], lam0=cfg["lam0"], R_fwhm=cfg["R"], seed=0)
obs = synth.obs_grid(raw["wavelength"], cfg["nobs"])
A = PayneEngine(nnio.normalize_spec_net(raw), obs=(obs, np.ones(len(obs)), np.full(len(obs), 0.01)), b_max=512)
small = synth.make_yst_net(npix=32, lam0=cfg["lam0"], R_fwhm=cfg["R"], seed=0)
Bn = PayneEngine(nnio.normalize_spec_net(small), obs=None, b_max=1024)
thA = A._theta(theta_full(synth.draw_candidates(512, seed=1)), A.ncols)
thB = Bn._theta(theta_full(synth.draw_candidates(1024, seed=2)), Bn.ncols)
lnl = torch.empty(512, dtype=torch.float64, device=thA.device)
outB = torch.empty((1024, 32), dtype=torch.float32, device=thA.device)
sA, sB = torch.cuda.current_stream(), torch.cuda.Stream()

def run(n, a=True, b=True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if a:
            A.lnlike_batch(thA, out=lnl)
        if b:
            rc = Bn.lib.payne_predict_batch(Bn._ctx, thB.data_ptr(), 1024, 0, 0, outB.data_ptr(), 32, C.c_void_p(sB.cuda_stream))
            assert rc == 0
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n

for _ in range(3):
    run(50)
res = {"A alone": np.median([run(300, True, False) for _ in range(5)]), "B alone": np.median([run(300, False, True) for _ in range(5)]),
       "A + B, two streams": np.median([run(300, True, True) for _ in range(5)])}
for k, v in res.items():
    print("%-22s %.2f us per step" % (k, v))
print("B's launches beside A cost A's step %.2f us of B's %.2f" % (res["A + B, two streams"] - res["A alone"], res["B alone"]))
print("kernels of B:", Bn.kernels_used())
