#!/bin/bash
# the GPU's timeline of the end-to-end run with the walk's hidden layers computed ahead: cycle of a walk step, what runs beside what
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
V=${1:-0}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/ahead_tl_$V -o s -- python3 $REPO/tools/sampler_bench.py --config C2 --maxcall 300000 --modes device_chunks --variant $V --dlogz 1e-9 > $OUT/ahead_tl_$V.log 2>&1
cd $REPO
python3 - <<PY
import csv, glob, numpy as np
rows = []
for f in glob.glob("$OUT/ahead_tl_$V/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
def kind(n):
    return "hidden" if "hidden" in n else "out" if "dense_dma" in n else "post" if "payne_post" in n else "spec" if "rwalk_spec" in n else "other"
post = [r for r in rows if kind(r[2]) == "post" and "true, true" in r[2]]
starts = np.array([r[0] for r in post]); ends = np.array([r[1] for r in post])
cyc = np.diff(starts) / 1e3
print("variant $V: post-to-post cycle: median %.2f us, p10 %.2f, p90 %.2f (n=%d)" % (np.median(cyc), np.percentile(cyc, 10), np.percentile(cyc, 90), len(cyc)))
for k in ("hidden", "out", "post", "spec"):
    d = [(r[1] - r[0]) / 1e3 for r in rows if kind(r[2]) == k]
    if d: print("  %-7s n=%6d median %.2f us" % (k, len(d), np.median(d)))
# a typical stretch of the timeline
i0 = len(rows) // 2
t0 = rows[i0][0]
for r in rows[i0:i0 + 16]:
    print("  %8.2f -> %8.2f us  q%-3s %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, r[3], r[2][:60]))
PY
