one() { python bench.py --config $2 --steps 300 --warmup 30 --no-cpu-baseline --no-e2e --no-also 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', round(d['value']), 'evals/s', round(d['ms_per_step']*1e3,2), 'us/step', {k: round(v,2) for k,v in d['kernels_us'].items()})"; }
for c in C2 LinNet300; do for i in 1 2 3; do PAYNE_HK_WAVES=8 one wide $c; PAYNE_HK_WAVES=4 one narrow $c; done; done
