#!/bin/bash
b() { PAYNE_HIP_LIB=$1 python bench.py --config ${3:-C2} --steps 400 --warmup 40 --no-cpu-baseline --no-e2e --no-also --unchecked 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', round(d['value']), round(d['ms_per_step']*1e3,2), {k: round(v,2) for k,v in d['kernels_us'].items()})"; }
timeout 900 python -m pytest tests/test_sampler_gpu.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for rep in 1 2; do
b $PWD/thepayne_amd/build/old/libpayne_hip_old.so old
b $PWD/thepayne_amd/libpayne_hip.so new
done
python tools/exp/cycle_times.py C2 2>&1 | tail -7
python tools/exp/cycle_times.py C3 2>&1 | tail -7
