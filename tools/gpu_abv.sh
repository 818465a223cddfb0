#!/bin/bash
# A/B of a variant library (thepayne_amd/build/var/libpayne_hip_<tag>.so) against the running build, interleaved, one box session
#   bash tools/gpu_abv.sh <tag> [C2|C3|C5] [repeats]
TAG=$1; CFG=${2:-C2}; REP=${3:-3}
VAR=$PWD/thepayne_amd/build/var/libpayne_hip_$TAG.so
STEPS=300; WARM=30
if [ $CFG = C5 ]; then STEPS=5; WARM=2; fi
one() {
  python bench.py --config $CFG --steps $STEPS --warmup $WARM --no-cpu-baseline --no-e2e --no-also 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$CFG', round(d['value']), 'evals/s', round(d['ms_per_step']*1e3,1), 'us/step', {k: round(v,2) for k,v in d['kernels_us'].items()})"
}
PAYNE_HIP_LIB=$VAR timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or c2 or full_size" 2>&1 | tail -2
for i in $(seq $REP); do
  PAYNE_HIP_LIB=$VAR one $TAG
  one base
done
