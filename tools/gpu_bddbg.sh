#!/bin/bash
# timing experiments on payne_dense_bx3dma_kernel (results are invalid with PAYNE_BD_DBG != 0)
OUT=$PWD/gpurun_out; mkdir -p $OUT
for d in 0 1 2 4 8 3 7 15; do
PAYNE_OUT_TILE=9 PAYNE_BD_DBG=$d python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-e2e > $OUT/bench_bd_$d.log 2>&1
python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_bd_$d.log").read().strip().splitlines()[-1])
    print("dbg $d dense_out %.1f us" % d["kernels_us"]["dense_out"])
except Exception as e: print("dbg $d failed", open("$OUT/bench_bd_$d.log").read()[-300:])
PY
done
