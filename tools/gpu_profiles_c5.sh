#!/bin/bash
# rocprofv3 evidence for the C5 (65 536-pixel, HBM/L2-bound) configuration: kernel stats + HBM counters
TAG=${1:-r1c5}
OUT=$PWD/gpurun_out; mkdir -p $OUT
REPO=$PWD
python bench.py --config C5 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $OUT/bench_${TAG}.json 2> $OUT/bench_${TAG}.err; tail -c 900 $OUT/bench_${TAG}.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o kt -- python3 $REPO/bench.py --config C5 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-e2e > $OUT/rocprof_$TAG.log 2>&1; echo "rocprof stats rc=$?"
for cn in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $cn --output-format csv -d $OUT/pmc_${TAG}_$cn -o c -- python3 $REPO/bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-e2e > $OUT/pmc_${TAG}_$cn.log 2>&1; echo "pmc $cn rc=$?"
done
cd $REPO
python3 - <<PY
import csv, glob, collections
rows = []
for cn in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/pmc_${TAG}_%s/**/*counter_collection.csv" % cn, recursive=True):
        agg = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cn:
                agg[r["Kernel_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]] += 1
        for k in agg:
            if "payne" in k: rows.append((cn, k, agg[k] / n[k], n[k]))
with open("$OUT/pmc_${TAG}_hbm.csv", "w") as fo:
    fo.write("counter,kernel,mean_value_per_launch_KB_raw,launches\n")
    for r in rows: fo.write('%s,"%s",%.1f,%d\n' % r)
print(open("$OUT/pmc_${TAG}_hbm.csv").read())
for f in glob.glob("$OUT/prof_$TAG/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:1200])
PY
