"""Kernel times INSIDE the sampler's loop against the bare likelihood step (GPU box): where the 12-14 % between `end_to_end` and the
kernel-only figure sit.  payne_profile's HIP events around every launch of the engine while tools/sampler_bench.py runs its default mode."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import sampler_bench as sb
from thepayne_amd.sampler.nested import NestedSampler
from thepayne_amd.sampler.device import DeviceProposer

VAR = int(sys.argv[1]) if len(sys.argv) > 1 else 0        # e.g. 131072 = PAYNE_V_NO_WALK_SPEC
L, P = sb.make_problem("C2", 512, VAR)
eng = L.GM.engine
for prof in (False, True):
    prop = DeviceProposer(L, P, k_max=512)
    S = NestedSampler(sb.lnprob_batch, P.priortrans_batch, L.ndim, logl_args=[L, P], nlive=512, bound='multi', sample='rwalk', walks=25,
                      batched=True, queue_size=512, rstate=np.random.default_rng(1), proposer=prop)
    for _ in S.sample_chunks(maxcall=60000, dlogz=1e-9):
        pass
    if prof:
        eng.profile(True)
    c0, t0 = S.ncall, time.perf_counter()
    for _ in S.sample_chunks(maxcall=S.ncall + 600000, dlogz=1e-9):
        pass
    dt = time.perf_counter() - t0
    calls = S.ncall - c0
    print("profile", prof, "calls/s %.3f M" % (calls / dt / 1e6), "us per 512 calls %.2f" % (dt / calls * 512 * 1e6))
    if prof:
        k = eng.profile_read()
        eng.profile(False)
        print(json.dumps(k))
    prop.close()
