#!/bin/bash
# instruction-cache counters of the C2 step's kernels
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d $OUT/pmc_ic -o c -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-e2e > $OUT/pmc_ic.log 2>&1
echo rc=$?
cd $REPO
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/pmc_ic/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:44]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in agg:
        if "payne" in k: print(k, {c: round(v / n[(k, c)]) for c, v in agg[k].items()})
PY
tail -3 $OUT/pmc_ic.log
