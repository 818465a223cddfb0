for r in 1 2 3; do
  for which in prev new; do
    if [ $which = prev ]; then export PAYNE_HIP_LIB=$PWD/thepayne_amd/build/old/libpayne_hip_prev.so; else unset PAYNE_HIP_LIB; fi
    python tools/sampler_bench.py --config C2 --maxcall 700000 --modes device_chunks --dlogz 1e-9 2>&1 | grep sampler_bench | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$which', d['sampler_bench']['device_chunks']['evals_per_s'], d['sampler_bench']['device_chunks']['logz'])"
  done
done
