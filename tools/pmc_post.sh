#!/bin/bash
# SQ counters of the C2 step's kernels (two passes, --no-also: the headline's three kernels only).   bash tools/pmc_post.sh [tag]
TAG=${1:-p}
OUT=$PWD/gpurun_out
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_${TAG}_$name -o c -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-e2e --no-also > $OUT/pmc_${TAG}_$name.log 2>&1
  echo "pmc $name rc=$?"
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM
run sq2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run sq3 SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL
cd $REPO
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/pmc_${TAG}_*/")):
    for f in glob.glob(d + "/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:44]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        print(d.split("/")[-2])
        for k in agg:
            if "payne" in k:
                print("  ", k, {c: round(v / n[(k, c)]) for c, v in agg[k].items()})
PY
