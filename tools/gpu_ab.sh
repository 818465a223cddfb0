#!/bin/bash
# A/B of the running build against a kept library (thepayne_amd/build/old/libpayne_hip_head.so) in ONE box session, interleaved runs.
#   bash tools/gpu_ab.sh [C2|C3|C5] [repeats]
CFG=${1:-C2}; REP=${2:-3}
OLD=${OLD:-$PWD/thepayne_amd/build/old/libpayne_hip_head.so}
STEPS=300; WARM=30
if [ $CFG = C5 ] || [ $CFG = C32k ]; then STEPS=5; WARM=2; fi
one() {
  python bench.py --config $CFG --steps $STEPS --warmup $WARM --no-cpu-baseline --no-e2e --no-also 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$CFG', round(d['value']), 'evals/s', round(d['ms_per_step']*1e3,1), 'us/step', {k: round(v,2) for k,v in d['kernels_us'].items()})"
}
for i in $(seq $REP); do
  one new
  if [ -f $OLD ]; then PAYNE_HIP_LIB=$OLD one old; fi
done
