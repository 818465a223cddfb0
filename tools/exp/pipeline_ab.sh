#!/bin/bash
# The sampler's end-to-end rate with the next queue launched ahead (default) against every queue launched after the one before
# is consumed (pipeline=False), interleaved, same seeds.   bash tools/exp/pipeline_ab.sh [C2|C3] [repeats]
CFG=${1:-C2}; REP=${2:-3}
python - <<PY
import sys, os
sys.path.insert(0, "tools")
import sampler_bench
sampler_bench.run("$CFG", maxcall=60000, modes=("device_chunks",))           # warm-up
for i in range($REP):
    for mode in ("device_chunks", "device_chunks_serial"):
        r = sampler_bench.run("$CFG", maxcall=700000, nlive=512, walks=25, modes=(mode,), seed=1 + i, dlogz=1e-9)[mode]
        print("%-22s seed %d  %.3f M calls/s  calls %d  iterations %d  logz %.3f  scale %.3f" % (mode, 1 + i, r["evals_per_s"] / 1e6, r["calls"], r["iterations"], r["logz"], r["scale"]))
PY
