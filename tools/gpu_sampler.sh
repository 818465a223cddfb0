#!/bin/bash
# GPU session: device-side sampler tests + sampler end-to-end throughput + the bench line
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_sampler_gpu.py -x -q 2>&1 | tail -3
timeout 600 python tools/sampler_bench.py --maxcall 400000 > $OUT/sampler_bench.log 2>&1; tail -5 $OUT/sampler_bench.log | cut -c1-300
