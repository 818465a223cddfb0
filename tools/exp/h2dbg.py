import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from thepayne_amd import synth, nnio, _lib
from thepayne_amd.engine import PayneEngine
from helpers import theta_full
cfg=synth.CONFIGS["C2"]
raw=synth.make_yst_net(npix=512, lam0=cfg["lam0"], R_fwhm=cfg["R"], H=300, seed=0)
# output layer = identity on the first 300 hidden units: rows ARE the second layer's activations
raw["w_array_2"][:]=0; raw["b_array_2"][:]=0
for k in range(300): raw["w_array_2"][k,k]=1.0
net=nnio.normalize_spec_net(raw)
B=2048
th=theta_full(synth.draw_candidates(B, seed=55))
ref=None
e0=PayneEngine(net, obs=None, b_max=B, variant=_lib.V_HID_F32|_lib.V_OUT_F32)
ref=e0.predict_batch(th, stage=0).cpu().numpy().astype(np.float64)[:, :300]; e0.close()
e=PayneEngine(net, obs=None, b_max=B, variant=_lib.V_OUT_F32)
for rep in range(40):
    got=e.predict_batch(th, stage=0).cpu().numpy().astype(np.float64)[:, :300]
    d=np.abs(got-ref)
    bad=np.argwhere(d>1e-5)
    if len(bad) or rep % 10 == 0: print("rep",rep,"max",d.max(),"nbad",len(bad))
    if len(bad):
        rows=np.unique(bad[:,0]); cols=np.unique(bad[:,1])
        print("  rows",rows[:40]); print("  cols",cols[:60], "ncols", len(cols))
        r0,c0=bad[0]; print("  sample got",got[r0,c0],"ref",ref[r0,c0])
