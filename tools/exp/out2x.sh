#!/bin/bash
# the output layer launched twice in a row (twin built with -DPAYNE_EXP_OUT2X): durations of the first and the second dispatch
OUT=$PWD/gpurun_out; REPO=$PWD
export PAYNE_HIP_LIB=$REPO/thepayne_amd/build/var/libpayne_hip_out2x.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/out2x -o kt -- python3 $REPO/bench.py --config C2 --steps 50 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-e2e --no-also --unchecked > $OUT/out2x.log 2>&1
python3 - <<PY
import csv, statistics
rows = [r for r in csv.DictReader(open("$OUT/out2x/kt_kernel_trace.csv")) if "dma3" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
print(len(d), "dispatches; first of a pair median", statistics.median(d[0::2]), "ns; second", statistics.median(d[1::2]), "ns")
PY
