#!/bin/bash
# quick GPU check: parity tests + bench variants (env knobs)
TAG=${1:-q}
OUT=$PWD/gpurun_out; mkdir -p $OUT
b() { name=$1; shift; env "$@" python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_${TAG}_$name.log 2>&1
python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_${TAG}_$name.log").read().strip().splitlines()[-1])
    print("$name", round(d["value"]), "evals/s", round(d["ms_per_step"]*1e3,1), "us/step", {k: round(v,1) for k,v in d["kernels_us"].items()})
except Exception as e: print("$name failed", e, open("$OUT/bench_${TAG}_$name.log").read()[-800:])
PY
}
python bench.py --config C5 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_${TAG}_C5.log 2>&1; tail -1 $OUT/bench_${TAG}_C5.log | cut -c1-1500
python bench.py --config C5 --batch 512 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_${TAG}_C5b512.log 2>&1; tail -1 $OUT/bench_${TAG}_C5b512.log | cut -c1-400
