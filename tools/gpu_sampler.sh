#!/bin/bash
# GPU session: device-side sampler tests + sampler end-to-end throughput + the bench line
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_sampler_gpu.py tests/test_api_gpu.py -x -q 2>&1 | tail -5
timeout 600 python tools/sampler_bench.py --maxcall 300000 > $OUT/sampler_bench.log 2>&1; tail -4 $OUT/sampler_bench.log | cut -c1-400
timeout 900 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_e2e.log 2>&1; tail -1 $OUT/bench_e2e.log | cut -c1-2500
