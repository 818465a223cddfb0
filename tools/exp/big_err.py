import os, sys, numpy as np, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O
from thepayne_amd import synth, nnio
from thepayne_amd.predict.ystpred import PayneSpecPredict
tmp = tempfile.mkdtemp()
for npix, depth in ((20000, 0.3), (20000, 0.02), (40000, 0.3), (65536, 0.3)):
    net = synth.make_yst_net(npix=npix, H=32, seed=26, D=4, line_depth=depth)
    path = os.path.join(tmp, "n%d_%g.npz" % (npix, depth))
    nnio.save_npz(path, {k: (np.array([v]) if k == "resolution" else v) for k, v in net.items() if k != "kind"})
    PP = PayneSpecPredict(nnpath=path, NNtype='YST1')
    w = net["wavelength"]
    obs = np.linspace(w[0] + 1, w[-1] - 1, 5000)
    worst = 0
    for R, vrot, rv in ((18002.4, 0.0, 0.0), (30000.0, 5.0, 30.0), (50000.0, 20.0, -100.0), (12000.0, 1.0, 5.0)):
        kw = dict(Teff=4345.0, logg=1.61, feh=-0.5, afe=0.33, rad_vel=rv, rot_vel=vrot, inst_R=R, outwave=obs)
        _, f = PP.getspec(**kw)
        _, fo = O.getspec(net, **kw)
        ok = ~np.isnan(fo)
        worst = max(worst, np.abs(f[ok] - fo[ok]).max())
    print(npix, depth, "tiled=" + os.environ.get("PAYNE_BIG_TILED", "1"), "max err %.3g" % worst, "flux range", float(np.nanmin(fo)), float(np.nanmax(fo)))
