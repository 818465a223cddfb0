#!/bin/bash
# end-to-end calls/s of the sampler's loops, the build in the tree against thepayne_amd/build/old/libpayne_hip_prev.so, interleaved
#   bash tools/exp/ab_e2e.sh [mode]        mode: device_chunks (default loop) | device_chunks_hostturn | ...
MODE=${1:-device_chunks}
for r in 1 2 3; do
  for which in prev new; do
    if [ $which = prev ]; then export PAYNE_HIP_LIB=$PWD/thepayne_amd/build/old/libpayne_hip_prev.so; else unset PAYNE_HIP_LIB; fi
    python tools/sampler_bench.py --config C2 --maxcall 700000 --modes $MODE --dlogz 1e-9 2>&1 | grep sampler_bench | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$which', '$MODE', d['sampler_bench']['$MODE']['evals_per_s'], d['sampler_bench']['$MODE']['logz'])"
  done
done
