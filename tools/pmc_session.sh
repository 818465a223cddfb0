#!/bin/bash
# PMC counters for the kernels of one bench run (separate passes; no tracing domains beside kernel-trace).
TAG=${1:-p}
OUT=$PWD/gpurun_out
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_${TAG}_$name -o c -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-e2e > $OUT/pmc_${TAG}_$name.log 2>&1
  echo "pmc $name rc=$?"
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM
run sq2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run mem1 FETCH_SIZE
run mem2 WRITE_SIZE
run mfma SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
cd $REPO
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/pmc_${TAG}_*/")):
    for f in glob.glob(d + "/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        print(d.split("/")[-2])
        for k in agg:
            if "payne" in k:
                print("  ", k, {c: round(v / n[(k, c)]) for c, v in agg[k].items()})
PY
