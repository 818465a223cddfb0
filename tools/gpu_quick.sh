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
python -m pytest tests -m gpu -q > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_$TAG.log
b base A=1
