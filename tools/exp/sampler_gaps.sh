#!/bin/bash
# GPU timeline of the end-to-end sampler run: where the GPU idles (rocprofv3 kernel trace; gaps between consecutive kernels)
CFG=${1:-C2}
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/sampler_gaps -o s -- python3 $REPO/tools/sampler_bench.py --config $CFG --maxcall 400000 --modes device_chunks --variant ${2:-0} > $OUT/sampler_gaps.log 2>&1
cd $REPO
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/sampler_gaps/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in csv.DictReader(open(f))))
# keep the steady state: the last 60 % of the trace
rows = rows[int(0.4 * len(rows)):]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = [(rows[i + 1][0] - rows[i][1], rows[i][2], rows[i + 1][2]) for i in range(len(rows) - 1)]
big = [g for g in gaps if g[0] > 20000]
small = [g for g in gaps if g[0] <= 20000]
print("kernels %d, span %.1f ms, busy %.1f ms (%.1f %%)" % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span))
print("gaps <= 20 us: %d, total %.2f ms, mean %.2f us" % (len(small), sum(g[0] for g in small) / 1e6, sum(g[0] for g in small) / max(1, len(small)) / 1e3))
print("gaps  > 20 us: %d, total %.2f ms, mean %.1f us" % (len(big), sum(g[0] for g in big) / 1e6, sum(g[0] for g in big) / max(1, len(big)) / 1e3))
c = collections.Counter((a, b) for _, a, b in big)
print(c.most_common(5))
per = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    per[n][0] += 1; per[n][1] += e - s
for n, (k, t) in sorted(per.items(), key=lambda x: -x[1][1])[:8]:
    print("%-42s %6d calls  %8.2f ms  %7.2f us each" % (n, k, t / 1e6, t / k / 1e3))
PY
