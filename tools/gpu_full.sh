#!/bin/bash
# every -m gpu test, smoke, the C2 bench line (with end_to_end) and the C5 bench
TAG=${1:-full}
OUT=$PWD/gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > $OUT/bench_${TAG}.json 2> $OUT/bench_${TAG}.err; tail -c 1500 $OUT/bench_${TAG}.json; tail -5 $OUT/bench_${TAG}.err
python bench.py --config C5 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_c5_${TAG}.json 2> $OUT/bench_c5_${TAG}.err; tail -c 1200 $OUT/bench_c5_${TAG}.json
