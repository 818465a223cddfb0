#!/bin/bash
# The GPU's timeline around the boundary between two proposal queues of the end-to-end sampler run (rocprofv3 kernel trace): every
# kernel from the last post kernel of a queue to the first hidden-layer launch of the next, with start / end relative to that post kernel's end.
CFG=${1:-C2}
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/turn_tl -o s -- python3 $REPO/tools/sampler_bench.py --config $CFG --maxcall 400000 --modes ${2:-device_chunks} --dlogz 1e-9 > $OUT/turn_tl.log 2>&1
cd $REPO
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/turn_tl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]) for r in csv.DictReader(open(f))))
rows = rows[int(0.5 * len(rows)):]
shown = 0
for i in range(1, len(rows)):
    if "rwalk_kernel" in rows[i][2] or "stage_out" in rows[i][2]:
        # find the post kernel before and the hidden kernel after
        j = i
        while j > 0 and "post_kernel" not in rows[j][2]: j -= 1
        k = i
        while k < len(rows) - 1 and "hidden" not in rows[k][2]: k += 1
        if k - j > 12 or j == 0: continue
        t0 = rows[j][1]
        print("--- boundary: idle %.1f us from post end to hidden start" % ((rows[k][0] - t0) / 1e3))
        for s, e, n in rows[j:k + 1]:
            print("   %8.1f .. %8.1f us  (%5.1f)  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
        shown += 1
        if shown >= 3: break
PY
