#!/bin/bash
# A/B in one box session: old library vs new (bench C2 x2 each, interleaved), then per-phase stamps of both diag builds
TAG=${1:-r3e}
OUT=$PWD/gpurun_out; mkdir -p $OUT
b() { PAYNE_HIP_LIB=$1 python bench.py --config ${3:-C2} --no-cpu-baseline --no-e2e --no-also --unchecked 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', round(d['value']), round(d['ms_per_step']*1e3,2), {k: round(v,2) for k,v in d['kernels_us'].items()})"; }
OLD=$PWD/thepayne_amd/build/old/libpayne_hip_old.so; NEW=$PWD/thepayne_amd/libpayne_hip.so
b $OLD old; b $NEW new; b $OLD old; b $NEW new
b $NEW newC3 C3
echo "---- stamps OLD"; STAMP_LIB=$PWD/thepayne_amd/build/old/libpayne_hip_diag_old.so timeout 300 python tools/post_stamps.py 2>&1 | head -34
echo "---- stamps NEW"; STAMP_LIB=$PWD/thepayne_amd/build/var/libpayne_hip_diag.so timeout 300 python tools/post_stamps.py 2>&1 | head -34
STAMP_LIB=$PWD/thepayne_amd/build/var/libpayne_hip_diag.so PAYNE_HIP_LIB=$PWD/thepayne_amd/build/var/libpayne_hip_diag.so timeout 300 python tools/exp/sed_stamps.py 2>&1 | tail -8
