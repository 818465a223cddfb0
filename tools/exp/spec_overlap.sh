#!/bin/bash
# kernel timeline of tools/exp/spec_overlap.py: do B's hidden-layer launches overlap A's output-layer / post kernels, and what do they cost them?
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/spec_ov -o s -- python3 $REPO/tools/exp/spec_overlap.py > $OUT/spec_ov.log 2>&1
cd $REPO
tail -5 $OUT/spec_ov.log
python3 - <<PY
import csv, glob, numpy as np
rows = []
for f in glob.glob("$OUT/spec_ov/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "0"))))
rows.sort()
def kind(n):
    return "hidden" if "hidden" in n else "out" if "dense_dma" in n else "post" if "payne_post" in n else "other"
# B's kernels: the hidden launches with 1024 rows are the LONG hidden kernels; tell the two engines apart by queue
qs = sorted(set(r[3] for r in rows))
print("queues:", qs)
byq = {q: [r for r in rows if r[3] == q] for q in qs}
for q in qs:
    d = {}
    for s, e, n, _ in byq[q]:
        d.setdefault(kind(n), []).append((e - s) / 1e3)
    print("queue", q, {k: (len(v), round(float(np.median(v)), 2)) for k, v in d.items()})
# durations of A's out / post kernels split by whether a kernel of another queue overlapped them in time
import bisect
allk = rows
def overlapped(s, e, q):
    return any(r[3] != q and r[0] < e and r[1] > s and kind(r[2]) == "hidden" for r in allk[max(0, bisect.bisect_left(allk, (s - 100000,))):bisect.bisect_right(allk, (e,))])
for q in qs:
    for k in ("out", "post", "hidden"):
        a = [((e - s) / 1e3, overlapped(s, e, q)) for s, e, n, _ in byq[q] if kind(n) == k]
        if len(a) > 50:
            yes = [x for x, o in a if o]; no = [x for x, o in a if not o]
            print("queue", q, k, "alone: n=%d median %.2f us" % (len(no), np.median(no) if no else float("nan")), "| beside another queue's hidden kernel: n=%d median %.2f us" % (len(yes), np.median(yes) if yes else float("nan")))
PY
