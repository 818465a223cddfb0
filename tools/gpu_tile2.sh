#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
for t in "$@"; do
PAYNE_OUT_TILE=$t timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -1
PAYNE_OUT_TILE=$t python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-e2e > $OUT/bench_tile_$t.log 2>&1
python - <<PY
import json
d=json.loads(open("$OUT/bench_tile_$t.log").read().strip().splitlines()[-1])
print("tile $t", round(d["value"]), "evals/s", round(d["ms_per_step"]*1e3,1), "us/step", {k: round(v,1) for k,v in d["kernels_us"].items()})
PY
done
