#!/bin/bash
# photometry in the hidden launch: parity + C3 bench (both forms)
TAG=${1:-r3b}
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_api_gpu.py -x -q -k "c3_at_size or joint or sed or likelihood_object" 2>&1 | tail -8
python bench.py --config C3 --no-cpu-baseline --no-e2e > $OUT/bench_c3_${TAG}.json 2> $OUT/bench_c3_${TAG}.err; python -c "
import json; d=json.loads(open('$OUT/bench_c3_${TAG}.json').read().strip().splitlines()[-1]); print('C3 fused', round(d['value']), d['ms_per_step'], d['kernels_us'])"; tail -2 $OUT/bench_c3_${TAG}.err
python bench.py --config C3 --variant 8192 --no-cpu-baseline --no-e2e > $OUT/bench_c3v_${TAG}.json 2> $OUT/bench_c3v_${TAG}.err; python -c "
import json; d=json.loads(open('$OUT/bench_c3v_${TAG}.json').read().strip().splitlines()[-1]); print('C3 own launch', round(d['value']), d['ms_per_step'], d['kernels_us'])"
python bench.py --no-cpu-baseline --no-e2e --no-also > $OUT/bench_c2_${TAG}.json 2> $OUT/bench_c2_${TAG}.err; python -c "
import json; d=json.loads(open('$OUT/bench_c2_${TAG}.json').read().strip().splitlines()[-1]); print('C2', round(d['value']), d['ms_per_step'], d['kernels_us'])"
