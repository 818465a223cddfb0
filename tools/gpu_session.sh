#!/bin/bash
# One GPU-box session: parity tests, post-kernel phase stamps, bench variants, rocprof stats.
# Usage (from the repo root on the GPU box): bash tools/gpu_session.sh [tag]
TAG=${1:-s}
OUT=$PWD/gpurun_out
mkdir -p $OUT
python -m pytest tests -m gpu -q > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"
python tools/post_stamps.py > $OUT/stamps_$TAG.log 2>&1; echo "stamps rc=$?"
for t in 0 1 2; do
  PAYNE_OUT_TILE=$t python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_${TAG}_tile$t.log 2>&1; echo "bench tile $t rc=$?"
done
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --batch 1024 > $OUT/bench_${TAG}_b1024.log 2>&1
PAYNE_TW_GLOBAL=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --batch 1024 > $OUT/bench_${TAG}_b1024_twg.log 2>&1
PAYNE_TW_GLOBAL=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_${TAG}_b512_twg.log 2>&1
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o kt -- python3 $REPO/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing > $OUT/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
cd $REPO
tail -4 $OUT/pytest_$TAG.log
cat $OUT/stamps_$TAG.log | tail -45
for t in 0 1 2; do python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_${TAG}_tile$t.log").read().strip().splitlines()[-1])
    print("tile $t", round(d["value"]), "evals/s", d["ms_per_step"], d["kernels_us"])
except Exception as e: print("tile $t failed", e)
PY
done
for f in b1024 b1024_twg b512_twg; do python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_${TAG}_$f.log").read().strip().splitlines()[-1])
    print("$f", round(d["value"]), "evals/s", d["ms_per_step"], d["kernels_us"])
except Exception as e: print("$f failed", e)
PY
done
