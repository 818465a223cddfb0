#!/bin/bash
# GPU session focused on payne_post_kernel: parity, stamps, bench, SQ counters
TAG=${1:-post}
OUT=$PWD/gpurun_out; mkdir -p $OUT
REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
python tools/post_stamps.py > $OUT/stamps_$TAG.log 2>&1; tail -32 $OUT/stamps_$TAG.log
python bench.py --steps 300 --warmup 30 --no-cpu-baseline > $OUT/bench_${TAG}.log 2>&1
python - <<PY
import json
d=json.loads(open("$OUT/bench_${TAG}.log").read().strip().splitlines()[-1])
print("bench", round(d["value"]), "evals/s", round(d["ms_per_step"]*1e3,1), "us/step", {k: round(v,1) for k,v in d["kernels_us"].items()})
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -i -o "SQC_[A-Z_]*ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*" $OUT/counters_list.txt | sort -u | head -20
run() {
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_${TAG}_$name -o c -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing > $OUT/pmc_${TAG}_$name.log 2>&1
  echo "pmc $name rc=$?"
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM
run sq2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU
cd $REPO
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/pmc_${TAG}_*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        print(d.split("/")[-2])
        for k in agg:
            if "payne" in k:
                print("  ", k, {c: round(v / n[(k, c)]) for c, v in agg[k].items()})
PY
