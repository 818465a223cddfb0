#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "dense_fused or fused_dense" 2>&1 | tail -3
STAMP_LIB=$PWD/thepayne_amd/build/var/libpayne_hip_diag.so timeout 300 python tools/exp/fused_stamps.py 2>&1 | tail -20
