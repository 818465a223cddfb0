#!/bin/bash
# full GPU parity + stamps + bench (C2) + bench without prep records
TAG=${1:-q3}
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/post_stamps.py > $OUT/stamps_$TAG.log 2>&1; tail -26 $OUT/stamps_$TAG.log
for v in base noprep; do
  if [ $v = noprep ]; then export PAYNE_NO_PREP=1; fi
  python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-e2e > $OUT/bench_${TAG}_$v.log 2>&1
  python - <<PY
import json
d=json.loads(open("$OUT/bench_${TAG}_$v.log").read().strip().splitlines()[-1])
print("$v", round(d["value"]), "evals/s", round(d["ms_per_step"]*1e3,1), "us/step", {k: round(v,1) for k,v in d["kernels_us"].items()})
PY
done
