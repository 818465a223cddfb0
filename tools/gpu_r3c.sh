#!/bin/bash
TAG=${1:-r3c}
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_api_gpu.py -x -q 2>&1 | tail -5
for cfg in C2 C3; do
python bench.py --config $cfg --no-cpu-baseline --no-e2e --no-also > $OUT/bench_${cfg}_${TAG}.json 2> $OUT/bench_${cfg}_${TAG}.err; python -c "
import json; d=json.loads(open('$OUT/bench_${cfg}_${TAG}.json').read().strip().splitlines()[-1]); print('$cfg', round(d['value']), d['ms_per_step'], d['kernels_us'])"; tail -3 $OUT/bench_${cfg}_${TAG}.err | grep -v amdgpu.ids
done
timeout 600 python tools/exp/sed_stamps.py 2>&1 | tail -9
