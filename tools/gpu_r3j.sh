#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "c5 or big or long or 65" 2>&1 | tail -15
python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 chip', round(d['value']), round(d['ms_per_step'],3), {k: round(v,1) for k,v in d['kernels_us'].items()})"
python bench.py --config C5 --steps 3 --warmup 1 --variant 65536 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 workspace', round(d['value']), round(d['ms_per_step'],3), {k: round(v,1) for k,v in d['kernels_us'].items()})"
