for B in 256 512 1024 2048 4096; do
python bench.py --batch $B --steps 100 --warmup 10 --repeats 5 --no-cpu-baseline --no-e2e --no-also 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B', $B, round(d['value']), 'evals/s', round(d['ms_per_step']*1e3,1), 'us/step', {k: round(v,2) for k,v in d['kernels_us'].items()}, round(d['whole_step']['frac_of_fp32_peak_on_wall_time'],3))"
done
python bench.py --streams 2 --steps 200 --warmup 20 --repeats 5 --no-cpu-baseline --no-e2e --no-also 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams 2', round(d['value']), 'evals/s', round(d['ms_per_step']*1e3,1), 'us/step')"
