#!/bin/bash
TAG=${1:-r3g}
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python tools/exp/cycle_times.py C2 2>&1 | tail -7
python bench.py > $OUT/bench_${TAG}.json 2> $OUT/bench_${TAG}.err; python - <<PY
import json
d=json.loads(open("$OUT/bench_${TAG}.json").read().strip().splitlines()[-1])
print("C2", round(d["value"]), d["ms_per_step"], d["kernels_us"], "e2e", d["end_to_end"]["value"], d["end_to_end"]["frac_of_kernel_only"], d["end_to_end"]["rates"])
for k,v in d["also_measured"].items(): print(k, round(v.get("value",0)), v.get("ms_per_step"), v.get("kernels_us"), v.get("end_to_end",{}).get("frac_of_kernel_only"), v.get("error"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
tail -3 $OUT/bench_${TAG}.err
