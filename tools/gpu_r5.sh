#!/bin/bash
# round-5 session: every -m gpu test, smoke, then the running build against the kept library (thepayne_amd/build/old/libpayne_hip_head.so)
# interleaved on this box, then the cycle stamps of the running build.   bash tools/gpu_r5.sh [tag] [configs...]
TAG=${1:-r5}; shift
CFGS=${@:-C2}
OUT=$PWD/gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for c in $CFGS; do bash tools/gpu_ab.sh $c 3 2>&1 | tee -a $OUT/ab_${TAG}.log; done
if [ -z "$NOSTAMPS" ]; then timeout 600 python tools/post_stamps.py > $OUT/stamps_$TAG.log 2>&1; head -24 $OUT/stamps_$TAG.log; fi
