#!/bin/bash
# round-3 session A: new tests first (C3 at size, LSF fuzz past 8192 px), then the whole GPU suite, then the bench line
TAG=${1:-r3a}
OUT=$PWD/gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "c3_at_size or joint or sed" 2>&1 | tail -15
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python bench.py > $OUT/bench_${TAG}.json 2> $OUT/bench_${TAG}.err; tail -c 3000 $OUT/bench_${TAG}.json; tail -5 $OUT/bench_${TAG}.err
python bench.py --config C3 --no-cpu-baseline --no-e2e > $OUT/bench_c3_${TAG}.json 2> $OUT/bench_c3_${TAG}.err; tail -c 1500 $OUT/bench_c3_${TAG}.json; tail -3 $OUT/bench_c3_${TAG}.err
python bench.py --config C3 --variant 8192 --no-cpu-baseline --no-e2e > $OUT/bench_c3v_${TAG}.json 2> $OUT/bench_c3v_${TAG}.err; tail -c 600 $OUT/bench_c3v_${TAG}.json; tail -3 $OUT/bench_c3v_${TAG}.err
