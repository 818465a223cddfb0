#!/bin/bash
# Full GPU-box session: parity tests, smoke, default bench (with CPU baseline), rocprof kernel stats, PMC traffic.
TAG=${1:-full}
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
python -m pytest tests -m gpu -q > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_$TAG.log
python __graft_entry__.py smoke > $OUT/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke_$TAG.log
python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; cat $OUT/bench_$TAG.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o kt -- python3 $REPO/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/rocprof_$TAG.log 2>&1; echo "rocprof rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${TAG}_fetch -o c -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1; echo "pmc fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_${TAG}_write -o c -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1; echo "pmc write rc=$?"
cd $REPO
head -5 $OUT/prof_$TAG/kt_kernel_stats.csv | cut -c1-160
python3 - <<PY
import csv, glob, collections
for kind in ("fetch", "write"):
    for f in glob.glob("$OUT/pmc_${TAG}_%s/*counter_collection.csv" % kind):
        agg = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:48]
            agg[k] += float(r["Counter_Value"]); n[k] += 1
        for k in agg:
            if "payne" in k: print(kind, k, round(agg[k] / n[k], 1), "KB per launch (raw counter)")
PY
