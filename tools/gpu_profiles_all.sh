#!/bin/bash
# rocprofv3 evidence for all measured configurations (gpu_profiles.sh per config); copy gpurun_out/<tag>_* to profiles/ afterwards
TAG=${1:-r4}
for cfg in C2 C3 C2r C5 C32k LinNet300; do bash tools/gpu_profiles.sh $TAG $cfg > gpurun_out/profiles_${TAG}_$cfg.log 2>&1; tail -3 gpurun_out/profiles_${TAG}_$cfg.log; done
ls gpurun_out | grep "^${TAG}_" 
