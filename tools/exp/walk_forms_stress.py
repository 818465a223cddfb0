"""The three forms of the chain step (proposal made ahead / drawn at the tail / own launch) over random walks: chain counts, step
scales from tiny to cube-sized, one or several ellipsoids, thresholds -- the chains must agree to the bit every time (GPU box)."""
import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_sampler_gpu as T
from thepayne_amd import _lib

tmp = pathlib.Path(tempfile.mkdtemp())
n_bad = n_run = 0
for photscale, modpoly in ((True, False), (False, True)):
    props = []
    for variant in (0, _lib.V_NO_WALK_SPEC, _lib.V_NO_WALK_TAIL):
        L, P, _ = T._fit_objects(tmp, photscale=photscale, modpoly=modpoly, variant=variant)
        props.append((T._proposer(L, P), L))
    nd = props[0][1].ndim
    rng = np.random.default_rng(123 + nd)
    for rep in range(25):
        K = int(rng.integers(1, 65))
        U0 = rng.uniform(0.01, 0.99, size=(K, nd))
        scale = float(10 ** rng.uniform(-2.5, 0.0))
        n_ell = int(rng.integers(1, 4))
        axes = np.stack([np.tril(rng.normal(size=(nd, nd))) * 0.3 * scale + scale * np.eye(nd) for _ in range(n_ell)])
        ell = rng.integers(0, n_ell, size=K).astype(np.int32) if n_ell > 1 else None
        walks = int(rng.integers(1, 12))
        seed = int(rng.integers(0, 2 ** 40))
        outs = []
        for prop, _ in props:
            V0, lp0 = prop.lnprob_u(U0)
            lp0 = np.where(np.isnan(lp0), -np.inf, lp0)
            fin = lp0[np.isfinite(lp0)]
            lstar = -np.inf if (rep % 3 == 0 or len(fin) == 0) else float(np.percentile(fin, 40))
            outs.append(prop.rwalk(U0, V0, lp0, axes if n_ell > 1 else axes[0], 1.0, lstar, walks, seed=seed, ell=ell))
        n_run += 1
        for o in outs[1:]:
            if not all(np.array_equal(x, y) for x, y in zip(outs[0], o)):
                n_bad += 1
                print("MISMATCH", photscale, modpoly, rep, K, scale, n_ell, walks, seed)
    for prop, _ in props:
        prop.close()
print("walks compared: %d, mismatches: %d" % (n_run, n_bad))
sys.exit(1 if n_bad else 0)
