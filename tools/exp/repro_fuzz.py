"""Re-run one iteration of tests/test_fuzz_gpu.py::test_random_getspec_calls and show where the difference sits."""
import os, sys, numpy as np, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O
from thepayne_amd import synth, nnio
from thepayne_amd.predict.ystpred import PayneSpecPredict
SEED0, seed, D, target = int(sys.argv[1]), int(sys.argv[2]), 4, int(sys.argv[3])
rng = np.random.default_rng(100 + seed + 1000 * SEED0)
net = synth.make_yst_net(npix=[512, 700, 1024, 600, 4096, 3000, 20000, 40000][seed], H=32, seed=20 + seed, D=D, line_depth=0.3)
tmp = tempfile.mkdtemp(); path = os.path.join(tmp, "n.npz")
nnio.save_npz(path, {k: (np.array([v]) if k == "resolution" else v) for k, v in net.items() if k != "kind"})
PP = PayneSpecPredict(nnpath=path, NNtype='YST1')
wave = net["wavelength"]
alias = {"Teff": ["Teff", "logt"], "logg": ["logg", "log(g)"], "feh": ["feh", "[Fe/H]"], "afe": ["afe", "aFe", "[a/Fe]", "[alpha/Fe]"]}
for it in range(target + 1):
    kw, canon = {}, {}
    lab = dict(Teff=rng.uniform(4000, 7500), logg=rng.uniform(0.5, 5.2), feh=rng.uniform(-2, 0.4), afe=rng.uniform(-0.1, 0.5))
    for k, v in lab.items():
        if rng.uniform() < 0.15: continue
        name = alias[k][rng.integers(len(alias[k]))]
        kw[name] = np.log10(v) if name == "logt" else v
        canon[k] = 10.0 ** kw[name] if name == "logt" else v
    if D == 5 or rng.uniform() < 0.2: kw['vmic'] = rng.uniform(0.5, 2.5) if D == 5 else np.nan
    if rng.uniform() < 0.75: kw['rot_vel'] = [0.0, 1e-3, rng.uniform(0.2, 60.0)][rng.integers(3)]
    if rng.uniform() < 0.75: kw['rad_vel'] = [0.0, rng.uniform(-300, 300)][rng.integers(2)]
    nobs = int(rng.integers(50, 400))
    lo, hi = np.sort(rng.uniform(wave[0] - 2.0, wave[-1] + 2.0, 2))
    outwave = np.linspace(lo, max(hi, lo + 1.0), nobs) if rng.uniform() < 0.7 else None
    if outwave is not None: kw['outwave'] = outwave
    mode = rng.integers(7)
    if mode in (1, 3): kw['inst_R'] = float(rng.uniform(8000, 60000))
    elif mode == 2: kw['inst_R'] = [np.nan, 0.0, -5.0, float(net["resolution"]) * 1.2][rng.integers(4)]
    elif mode == 4 and outwave is not None:
        x = np.linspace(-0.5, 0.5, nobs)
        kw['inst_R'] = 0.08 * (1.0 + rng.uniform(-0.5, 0.5) * x + rng.uniform(0, 0.5) * x ** 2)
    canon.update({k: v for k, v in kw.items() if k in ('vmic', 'rot_vel', 'rad_vel', 'inst_R', 'outwave')})
print({k: v for k, v in kw.items() if np.ndim(v) == 0}, None if outwave is None else (len(outwave), outwave[0], outwave[-1]), wave[0], wave[-1])
with np.errstate(all="ignore"):
    w_o, f_o = O.getspec(net, **canon)
w, f = PP.getspec(**kw)
d = np.abs(f - f_o); i = int(np.nanargmax(d))
print("max err %.3g at pixel %d of %d; neighbours" % (d[i], i, len(d)), d[max(0, i - 2):i + 3], "flux", f_o[max(0, i - 2):i + 3])
print("sorted top errors", np.sort(d[~np.isnan(d)])[-5:])
