#!/bin/bash
# round-4 session: every -m gpu test, smoke, the full default bench line
TAG=${1:-r4}
OUT=$PWD/gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > $OUT/bench_${TAG}.json 2> $OUT/bench_${TAG}.err; tail -c 3000 $OUT/bench_${TAG}.json; tail -5 $OUT/bench_${TAG}.err
