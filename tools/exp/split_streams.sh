# one batch of 512 as k sub-batches on k streams (and more than one full batch in flight), kernel-only rate
for cfg in "512 1" "256 2" "128 4" "512 2" "512 3" "256 4" "1024 2"; do
set -- $cfg
python bench.py --batch $1 --streams $2 --steps 200 --warmup 20 --repeats 5 --no-cpu-baseline --no-e2e --no-also --no-kernel-timing 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch', $1, 'streams', $2, round(d['value']), 'evals/s', round(d['ms_per_step']*1e3,1), 'us per sub-batch ->', round(d['ms_per_step']*1e3*512/$1,1), 'us per 512 candidates')"
done
