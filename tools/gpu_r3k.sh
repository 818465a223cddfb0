#!/bin/bash
for cfg in C2 C3 C5; do bash tools/gpu_profiles.sh r3 $cfg > gpurun_out/profiles_r3_$cfg.log 2>&1; tail -3 gpurun_out/profiles_r3_$cfg.log; done
ls gpurun_out | grep "^r3_" 
