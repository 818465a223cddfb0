#!/bin/bash
# rocprofv3 evidence for the bench command: kernel stats, HBM counters (FETCH_SIZE, WRITE_SIZE: separate passes) and one SQ
# pass; the summaries land in gpurun_out/ under the names profiles/ uses (copy them there: gpurun_out/ is scratch).
#   bash tools/gpu_profiles.sh [round-tag r4] [config C2|C3|C5|C32k|LinNet300]
TAG=${1:-r4}; CFG=${2:-C2}
cfg=$(echo $CFG | tr A-Z a-z)
OUT=$PWD/gpurun_out; mkdir -p $OUT
REPO=$PWD
STEPS=200; PSTEPS=20
if [ $CFG = C5 ] || [ $CFG = C32k ]; then STEPS=5; PSTEPS=3; fi
EXTRA=""; if [ $CFG != C2 ]; then EXTRA="--no-cpu-baseline"; fi
python bench.py --config $CFG --steps $STEPS --warmup 5 $EXTRA > $OUT/${TAG}_${cfg}_bench.json 2> $OUT/${TAG}_${cfg}_bench.err; tail -c 400 $OUT/${TAG}_${cfg}_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_$cfg -o kt -- python3 $REPO/bench.py --config $CFG --steps $STEPS --warmup 5 --no-cpu-baseline --no-kernel-timing --no-e2e --no-also > $OUT/rocprof_${TAG}_$cfg.log 2>&1; echo "rocprof stats rc=$?"
for cn in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $cn --output-format csv -d $OUT/pmc_${TAG}_${cfg}_$cn -o c -- python3 $REPO/bench.py --config $CFG --steps $PSTEPS --warmup 3 --no-cpu-baseline --no-kernel-timing --no-e2e --no-also > $OUT/pmc_${TAG}_${cfg}_$cn.log 2>&1; echo "pmc $cn rc=$?"
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_${TAG}_${cfg}_sq1 -o c -- python3 $REPO/bench.py --config $CFG --steps $PSTEPS --warmup 3 --no-cpu-baseline --no-kernel-timing --no-e2e --no-also > $OUT/pmc_${TAG}_${cfg}_sq1.log 2>&1; echo "pmc sq1 rc=$?"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_${TAG}_${cfg}_sq2 -o c -- python3 $REPO/bench.py --config $CFG --steps $PSTEPS --warmup 3 --no-cpu-baseline --no-kernel-timing --no-e2e --no-also > $OUT/pmc_${TAG}_${cfg}_sq2.log 2>&1; echo "pmc sq2 rc=$?"
cd $REPO
python3 - <<PY
import csv, glob, collections, json, subprocess
from thepayne_amd.build import source_hash
rows = []
for cn in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/pmc_${TAG}_${cfg}_%s/**/*counter_collection.csv" % cn, recursive=True):
        agg = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cn:
                agg[r["Kernel_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]] += 1
        for k in agg:
            if "payne" in k: rows.append((cn, k, agg[k] / n[k], n[k]))
hbm = "${TAG}_${cfg}_rocprofv3_pmc_hbm.csv"
with open("$OUT/" + hbm, "w") as fo:
    fo.write("counter,kernel,mean_value_per_launch_KB_raw,launches\n")
    for r in rows: fo.write('%s,"%s",%.1f,%d\n' % r)
print(open("$OUT/" + hbm).read())
sq = []
for part in ("sq1", "sq2"):
    for f in glob.glob("$OUT/pmc_${TAG}_${cfg}_%s/**/*counter_collection.csv" % part, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(r["Kernel_Name"], r["Counter_Name"])] += 1
        for k in agg:
            if "payne" in k:
                for c_, v in agg[k].items(): sq.append((k, c_, v / n[(k, c_)], n[(k, c_)]))
with open("$OUT/${TAG}_${cfg}_rocprofv3_pmc_sq.csv", "w") as fo:
    fo.write("kernel,counter,mean_value_per_launch,launches\n")
    for r in sq: fo.write('"%s",%s,%.1f,%d\n' % r)
for f in glob.glob("$OUT/prof_${TAG}_$cfg/**/*kernel_stats.csv", recursive=True):
    open("$OUT/${TAG}_${cfg}_rocprofv3_kernel_stats.csv", "w").write(open(f).read())
    print(open(f).read()[:1800])
meta = {"source_hash": source_hash(), "hbm_csv": hbm, "config": "$CFG", "command": "python bench.py --config $CFG (tools/gpu_profiles.sh $TAG $CFG)",
        "what": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KB per launch, raw (FETCH_SIZE is doubled by bench.py as MI355X_MICROARCH.md prescribes)"}
json.dump(meta, open("$OUT/${TAG}_${cfg}_pmc_meta.json", "w"), indent=1)
PY
