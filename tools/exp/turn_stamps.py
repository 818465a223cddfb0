"""Cycle stamps inside payne_ns_turn_kernel (needs a build with the TST() stamps compiled in: a diagnostic edit, see NOTES)."""
import ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import sampler_bench                                                  # noqa: E402
from thepayne_amd import _lib                                         # noqa: E402
out = sampler_bench.run("C2", 200000, 512, 25, ("device_chunks",), dlogz=1e-9)
print(json.dumps(out))
lib = _lib.load()
st = (ctypes.c_ulonglong * 16)()
lib.payne_debug_turn_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
print("rc", lib.payne_debug_turn_stamps(st))
s = np.array(list(st), dtype=np.int64)
names = ["start", "export issued", "keys in LDS", "counters", "sorted", "threshold+scale", "live rows gathered", "barrier", "start slots", "chain rows", "fence", "barrier", "end"]
for k in range(1, 13):
    print("%-22s %7d cycles (100 MHz ticks x 24?)" % (names[k], s[k] - s[k - 1]))
print("total", s[12] - s[0])
