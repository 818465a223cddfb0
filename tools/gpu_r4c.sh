#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "larger_than_lds or c32k or c5_at_size" 2>&1 | tail -25
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
python bench.py --config C32k --steps 5 --warmup 2 --repeats 5 --no-cpu-baseline --no-e2e > $OUT/bench_c32k_r4c.json 2> $OUT/bench_c32k_r4c.err; python - <<PY
import json
d=json.loads(open("$OUT/bench_c32k_r4c.json").read().strip().splitlines()[-1])
print("C32k", round(d["value"]), "evals/s", round(d["ms_per_step"]*1e3,1), "us/step", {k: round(v,1) for k,v in d["kernels_us"].items()}, d["roofline"]["kernel"], round(d["roofline"]["frac"],3))
PY
tail -3 $OUT/bench_c32k_r4c.err
