#!/bin/bash
# A/B in one box session: the previous round's library against the current one (C2 bench, interleaved)
b() { PAYNE_HIP_LIB=$1 python bench.py --config ${3:-C2} --steps 400 --warmup 40 --no-cpu-baseline --no-e2e --no-also --unchecked 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', round(d['value']), round(d['ms_per_step']*1e3,2), {k: round(v,2) for k,v in d['kernels_us'].items()})"; }
V=$PWD/thepayne_amd/build/var
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for rep in 1 2 3; do
b $PWD/thepayne_amd/build/old/libpayne_hip_old.so old
b $PWD/thepayne_amd/libpayne_hip.so new
for f in $V/libpayne_hip_x*.so; do [ -f $f ] && b $f $(basename $f .so | sed 's/libpayne_hip_//'); done
done
b $PWD/thepayne_amd/libpayne_hip.so newC3 C3
