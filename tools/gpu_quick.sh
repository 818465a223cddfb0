#!/bin/bash
# parity + stamps + bench (C2), short
TAG=${1:-q2}
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
if [ -z "$NOSTAMPS" ]; then python tools/post_stamps.py > $OUT/stamps_$TAG.log 2>&1; tail -25 $OUT/stamps_$TAG.log; fi
python bench.py --steps 300 --warmup 30 --no-cpu-baseline > $OUT/bench_${TAG}.log 2>&1
python - <<PY
import json
d=json.loads(open("$OUT/bench_${TAG}.log").read().strip().splitlines()[-1])
print("bench", round(d["value"]), "evals/s", round(d["ms_per_step"]*1e3,1), "us/step", {k: round(v,1) for k,v in d["kernels_us"].items()})
PY
