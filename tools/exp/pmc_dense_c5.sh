#!/bin/bash
# L2 hit / miss and fabric read requests of the C5 output-layer kernel (rocprofv3 PMC; own passes)
OUT=$PWD/gpurun_out; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
  tag=$(echo $set | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_dense_$tag -o c -- python3 $REPO/bench.py --config C5 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-e2e --no-also > $OUT/pmc_dense_$tag.log 2>&1
  python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$OUT/pmc_dense_$tag/c_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:40]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "dense" in k or "post" in k:
        print(k, {c: (len(v), sum(v) / len(v)) for c, v in d.items()})
PY
done
