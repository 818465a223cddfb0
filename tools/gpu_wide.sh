#!/bin/bash
# A/B the DMA GEMM tile width
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for w in 1 0; do
PAYNE_DMA_WIDE=$w python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-e2e > $OUT/bench_wide_$w.log 2>&1
python - <<PY
import json
d=json.loads(open("$OUT/bench_wide_$w.log").read().strip().splitlines()[-1])
print("wide $w", round(d["value"]), "evals/s", round(d["ms_per_step"]*1e3,1), "us/step", {k: round(v,1) for k,v in d["kernels_us"].items()})
PY
done
