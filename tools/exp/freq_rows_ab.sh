#!/bin/bash
# The output layer carrying the first stage's forward transform (default) against pixel rows (--variant 262144), interleaved.
#   bash tools/exp/freq_rows_ab.sh [C2|C3|LinNet300] [repeats]
CFG=${1:-C2}; REP=${2:-3}
one() {
  python bench.py --config $CFG --steps 300 --warmup 30 --variant $2 --no-cpu-baseline --no-e2e --no-also 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$CFG', round(d['value']), 'evals/s', round(d['ms_per_step']*1e3,1), 'us/step', {k: round(v,2) for k,v in d['kernels_us'].items()})"
}
for i in $(seq $REP); do one freq 0; one pixel 262144; done
