#!/bin/bash
# Which group of post-kernel phases makes the LDS bank conflicts?  Twins with one group compiled out (-DPAYNE_EXP_SKIP=<mask>, built
# beforehand: PAYNE_VARIANT_DIR=thepayne_amd/build/var python tools/exp/ablate.py --build) under rocprofv3 PMC:
# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of payne_post_kernel per twin.
OUT=$PWD/gpurun_out; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for m in 0 1 2 4 8 16 32; do
  export PAYNE_HIP_LIB=$REPO/thepayne_amd/build/var/libpayne_hip_exp$m.so
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/ldsc_$m -o c -- python3 $REPO/bench.py --config C2 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-e2e --no-also --unchecked > $OUT/ldsc_$m.log 2>&1
  python3 - <<PY
import csv, collections
rows = [r for r in csv.DictReader(open("$OUT/ldsc_$m/c_counter_collection.csv")) if "payne_post_kernel<12, true, true>" in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in rows: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
v = {k: sum(x) / len(x) for k, x in agg.items()}
print("skip mask %2d: conflict cycles %9.0f  LDS cycles %9.0f  ratio %.3f  LDS instructions %8.0f" % ($m, v.get("SQ_LDS_BANK_CONFLICT", 0), v.get("SQ_LDS_IDX_ACTIVE", 0), v.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, v.get("SQ_LDS_IDX_ACTIVE", 1)), v.get("SQ_INSTS_LDS", 0)))
PY
done
