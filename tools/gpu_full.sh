#!/bin/bash
# all GPU tests + C2 bench (with e2e) + C5 bench
TAG=${1:-f2}
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 300 --warmup 30 --no-cpu-baseline > $OUT/bench_${TAG}_c2.log 2>&1
python bench.py --config C5 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $OUT/bench_${TAG}_c5.log 2>&1
python - <<PY
import json
for f in ("c2","c5"):
    try:
        d=json.loads(open("$OUT/bench_${TAG}_%s.log" % f).read().strip().splitlines()[-1])
        print(f, round(d["value"]), "evals/s", round(d["ms_per_step"]*1e3,1), "us/step", {k: round(v,1) for k,v in d["kernels_us"].items()}, d.get("end_to_end",{}).get("value"))
    except Exception as e:
        print(f, "failed", e, open("$OUT/bench_${TAG}_%s.log" % f).read()[-600:])
PY
