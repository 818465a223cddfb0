#!/bin/bash
# round-6 session: every -m gpu test, smoke, then bench.py exactly as the driver runs it (the LAST stdout line must parse, < 4 KB)
TAG=${1:-r6}
OUT=$PWD/gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_${TAG}_driver.out 2> $OUT/bench_${TAG}_driver.err
cp bench_detail.json $OUT/bench_${TAG}_driver_detail.json
python - <<PY
import json
t=open("$OUT/bench_${TAG}_driver.out").read().strip().splitlines()
print("stdout lines", len(t), "last line bytes", len(t[-1]))
d=json.loads(t[-1]); print(json.dumps(d)[:3000])
PY
tail -3 $OUT/bench_${TAG}_driver.err
