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
names = ["start", "loads requested", "counters, export's stores, scale", "sorted", "new live rows", "barrier", "start slots", "-",
         "chain rows", "bound's values stored", "barrier", "end (release, completion word)"]
for k in (1, 2, 3, 4, 5, 6, 8, 9, 10, 11):
    prev = k - 1 if k != 8 else 6
    print("%-36s %7d cycles" % (names[k], s[k] - s[prev]))
print("total", s[11] - s[0], "cycles of the shader clock (s_memtime)")
