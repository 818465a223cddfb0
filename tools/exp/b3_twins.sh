for v in b3x1 b3x2 b3x4; do
PAYNE_HIP_LIB=$PWD/thepayne_amd/build/var/libpayne_hip_$v.so python bench.py --config C5 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-also --unchecked 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', {k: round(v,2) for k,v in d['kernels_us'].items()})"
done
python bench.py --config C5 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-also 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full', {k: round(v,2) for k,v in d['kernels_us'].items()})"
