"""Where the host time of a queue cycle goes in the sampler's default loop (the turn on the device): wall clock of the pieces of
NestedSampler.sample_chunks around the native calls, per cycle; and how long the host WAITS in queue_dev_collect (GPU-bound) against how
long it works (host-bound).  GPU box."""
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
for n in ("_consume", "_update_bound", "_fit_bound", "_fill_queue", "_fill_queue_dev", "_prefetch_bound"):
    wrap(nested.NestedSampler, n)
for n in ("queue_dev_launch", "queue_dev_collect", "queue_dev_init"):
    wrap(device.DeviceProposer, n)
CFG = sys.argv[1] if len(sys.argv) > 1 else "C2"
sampler_bench.run(CFG, maxcall=60000, modes=("device_chunks",))            # warm-up
acc.clear()
r = sampler_bench.run(CFG, maxcall=2000000, modes=("device_chunks",), dlogz=1e-9)["device_chunks"]
print(r)
ncyc = acc["queue_dev_collect"][0]
print("cycles", ncyc, " seconds in the sampler loop", r["seconds"], " per cycle %.1f us" % (1e6 * r["seconds"] / ncyc))
for k, (n, t) in sorted(acc.items(), key=lambda x: -x[1][1]):
    print("%-18s %5d calls  %8.1f us per cycle  (%.1f us per call)" % (k, n, 1e6 * t / ncyc, 1e6 * t / n))
