"""Where a queue cycle of the batched nested sampler goes (GPU box): wall time of the native walk call (GPU + transfers),
of the dead-point bookkeeping (payne_ns_consume), of the bound update, and of everything else, per cycle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import sampler_bench
from thepayne_amd.sampler import nested
T = {"walk": 0.0, "consume": 0.0, "bound": 0.0, "n_walk": 0, "n_consume": 0, "n_bound": 0}
def wrap(cls, name, key):
    f = getattr(cls, name)
    def g(self, *a, **k):
        t0 = time.perf_counter()
        try:
            return f(self, *a, **k)
        finally:
            T[key] += time.perf_counter() - t0; T["n_" + key] += 1
    setattr(cls, name, g)
wrap(nested.NestedSampler, "_fill_queue", "walk")
wrap(nested.NestedSampler, "_consume", "consume")
wrap(nested.NestedSampler, "_update_bound", "bound")
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
sampler_bench.run(cfg, maxcall=100000, modes=("device_chunks",), dlogz=1e-9)
for k in T: T[k] = 0 if k.startswith("n_") else 0.0
r = sampler_bench.run(cfg, maxcall=700000, modes=("device_chunks",), dlogz=1e-9)["device_chunks"]
n = T["n_walk"]
tot = r["seconds"]
print("%s: %d calls in %.4f s = %.3f M/s; %d cycles of %.3f ms" % (cfg, r["calls"], tot, r["calls"] / tot / 1e6, n, 1e3 * tot / n))
for k in ("walk", "consume", "bound"):
    print("  %-8s %.3f ms per cycle (%d calls, %.1f us each)" % (k, 1e3 * T[k] / n, T["n_" + k], 1e6 * T[k] / max(1, T["n_" + k])))
print("  other    %.3f ms per cycle" % (1e3 * (tot - T["walk"] - T["consume"] - T["bound"]) / n))
print("  likelihood steps per cycle: walks + 1 = 26; calls per cycle %.0f" % (r["calls"] / n))
