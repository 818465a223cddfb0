#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
for cfg in "1 512" "2 256" "2 512" "4 128" "3 512"; do
set -- $cfg
python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-e2e --no-kernel-timing --streams $1 --batch $2 > $OUT/bench_s$1_b$2.log 2>&1
python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_s$1_b$2.log").read().strip().splitlines()[-1])
    print("streams $1 batch $2:", round(d["value"]), "evals/s", round(d["ms_per_step"]*1e3,1), "us/step")
except Exception as e: print("failed", open("$OUT/bench_s$1_b$2.log").read()[-300:])
PY
done
