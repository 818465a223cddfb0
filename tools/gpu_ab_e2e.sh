#!/bin/bash
# A/B of the sampler's end-to-end rate (bench.py's `end_to_end`) against a kept library, interleaved, in ONE box session;
# then the kernels' durations inside the sampler loop of both (tools/exp/sampler_gaps.sh).
#   bash tools/gpu_ab_e2e.sh [C2|C3] [repeats]
CFG=${1:-C2}; REP=${2:-2}
OLD=$PWD/thepayne_amd/build/old/libpayne_hip_head.so
mkdir -p gpurun_out
one() {
  python bench.py --config $CFG --steps 200 --warmup 30 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d.get('end_to_end',{}); print('$1', '$CFG', round(d['value']), 'evals/s kernel-only;', 'e2e', e.get('value'), e.get('frac_of_kernel_only'), e.get('runs'))"
}
for i in $(seq $REP); do
  one new
  if [ -f $OLD ]; then PAYNE_HIP_LIB=$OLD one old; fi
done
echo "--- kernels inside the sampler loop: new"; bash tools/exp/sampler_gaps.sh $CFG
if [ -f $OLD ]; then echo "--- old"; rm -rf gpurun_out/sampler_gaps; PAYNE_HIP_LIB=$OLD bash tools/exp/sampler_gaps.sh $CFG; fi
