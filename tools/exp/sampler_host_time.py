"""Where the host time of a sampler cycle goes (GPU box): wall-clock of the pieces of NestedSampler.sample_chunks around the native
queue call, per cycle.  (cProfile inflates the Python parts; this wraps the methods with perf_counter.)"""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
import sampler_bench
from thepayne_amd.sampler import nested, device

acc = collections.defaultdict(lambda: [0, 0.0])
def wrap(cls, name, key=None):
    f = getattr(cls, name)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            e = acc[key or name]; e[0] += 1; e[1] += time.perf_counter() - t
    setattr(cls, name, g)
for n in ("_consume", "_update_bound", "_fit_bound", "_fill_queue", "_prefetch_bound", "_launch_ahead"):
    wrap(nested.NestedSampler, n)
for n in ("rwalk_queue", "rwalk_queue_begin", "rwalk_queue_end"):
    wrap(device.DeviceProposer, n)
lib_calls = {}
orig_load = None
CFG = sys.argv[1] if len(sys.argv) > 1 else "C2"
sampler_bench.run(CFG, maxcall=60000, modes=("device_chunks",))            # warm-up
acc.clear()
t0 = time.perf_counter()
r = sampler_bench.run(CFG, maxcall=700000, modes=("device_chunks",), dlogz=1e-9)["device_chunks"]
print(r)
ncyc = acc["_fill_queue"][0]
print("cycles", ncyc, " seconds in the sampler loop", r["seconds"], " per cycle %.1f us" % (1e6 * r["seconds"] / ncyc))
for k, (n, t) in sorted(acc.items(), key=lambda x: -x[1][1]):
    print("%-18s %5d calls  %8.1f us per cycle  (%.1f us per call)" % (k, n, 1e6 * t / ncyc, 1e6 * t / n))
