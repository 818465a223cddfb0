#!/bin/bash
# a launch of the step repeated right away (twins built with -DPAYNE_EXP_OUT2X: the output layer; -DPAYNE_EXP_HID2X: the hidden
# layers): durations of the first and the second dispatch = what warm caches / code would be worth to that kernel
#   bash tools/exp/out2x.sh [out2x|hid2x]
TAG=${1:-out2x}; PAT=dma3; [ $TAG = hid2x ] && PAT=hidden
OUT=$PWD/gpurun_out; REPO=$PWD
export PAYNE_HIP_LIB=$REPO/thepayne_amd/build/var/libpayne_hip_$TAG.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/$TAG -o kt -- python3 $REPO/bench.py --config C2 --steps 50 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-e2e --no-also --unchecked > $OUT/$TAG.log 2>&1
python3 - <<PY
import csv, statistics
rows = [r for r in csv.DictReader(open("$OUT/$TAG/kt_kernel_trace.csv")) if "$PAT" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
print(len(d), "dispatches; first of a pair median", statistics.median(d[0::2]), "ns; second", statistics.median(d[1::2]), "ns")
PY
