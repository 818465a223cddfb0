#!/bin/bash
# C2 step and kernel times of library twins against the build in the tree, interleaved:  bash tools/exp/ab_bench.sh <twin.so> [<twin.so> ...]
for r in 1 2 3; do
  for lib in tree "$@"; do
    if [ $lib = tree ]; then unset PAYNE_HIP_LIB; else export PAYNE_HIP_LIB=$PWD/$lib; fi
    python bench.py --config C2 --steps 200 --warmup 5 --no-cpu-baseline --no-e2e --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']/1e6,3), 'M', d['ms_per_step'], {k: round(v,2) for k,v in d['kernels_us'].items()})"
  done
done
