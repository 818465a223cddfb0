#!/bin/bash
# the GPU's timeline of the end-to-end run (default loop): cycle of a walk step and the kernels' durations, by quarter of the run.
#   bash tools/exp/walk_timeline.sh [tag] [variant]      (PAYNE_HIP_LIB exported beforehand selects another build)
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
TAG=${1:-new}; V=${2:-0}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/walk_tl_$TAG -o s -- python3 $REPO/tools/sampler_bench.py --config C2 --maxcall 300000 --modes device_chunks --variant $V --dlogz 1e-9 > $OUT/walk_tl_$TAG.log 2>&1
cd $REPO
python3 - <<PY
import csv, glob, numpy as np
rows = []
for f in glob.glob("$OUT/walk_tl_$TAG/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
def kind(n):
    return "hidden" if "hidden" in n else "out" if "dense_dma" in n else "post" if "payne_post" in n else "other"
post = [r for r in rows if kind(r[2]) == "post" and "true, true" in r[2]]
starts = np.array([r[0] for r in post])
cyc = np.diff(starts) / 1e3
print("$TAG: post-to-post cycle: median %.2f us, p10 %.2f, p90 %.2f (n=%d)" % (np.median(cyc), np.percentile(cyc, 10), np.percentile(cyc, 90), len(cyc)))
t0, t1 = rows[0][0], rows[-1][1]
for q in range(4):
    lo, hi = t0 + (t1 - t0) * q / 4, t0 + (t1 - t0) * (q + 1) / 4
    line = "  quarter %d:" % (q + 1)
    for k in ("hidden", "out", "post"):
        d = [(r[1] - r[0]) / 1e3 for r in rows if kind(r[2]) == k and lo <= r[0] < hi and (k != "post" or "true, true" in r[2])]
        if d: line += "  %s %.2f us (n=%d)" % (k, np.median(d), len(d))
    print(line)
names = {}
for r in rows: names.setdefault(r[2][:70], []).append((r[1] - r[0]) / 1e3)
for n, d in sorted(names.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print("  %8.1f us total  n=%5d  median %.2f  %s" % (sum(d), len(d), np.median(d), n))
PY
