#!/bin/bash
# parity of the chip kernels + A/B against the kept library (C2, C5, C32k)
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_fuzz_gpu.py tests/test_api_gpu.py -x -q 2>&1 | tail -3
sed -i 's/if \[ \$CFG = C5 \]; then STEPS=5; WARM=2; fi/if [ $CFG = C5 ] || [ $CFG = C32k ]; then STEPS=5; WARM=2; fi/' tools/gpu_ab.sh
bash tools/gpu_ab.sh C5 2
bash tools/gpu_ab.sh C32k 2
bash tools/gpu_ab.sh C2 3
