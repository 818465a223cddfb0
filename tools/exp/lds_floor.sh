#!/bin/bash
# counters of tools/exp/lds_floor (built here beforehand: the binary travels)
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $OUT/lds_floor -o c -- $REPO/tools/exp/lds_floor > $OUT/lds_floor.log 2>&1
cd $REPO
python3 - <<PY
import csv, glob, collections
names = {0: "ds_write_b64, consecutive slots", 1: "ds_read_b64, consecutive slots", 2: "ds_write_b64, first radix-8 pass (slot 8 i + (i >> 2) + r)",
         3: "ds_write_b64, second pass (72 g + k + 8 r)", 4: "ds_read_b32 gather, 1.14 words a lane", 5: "two ds_read_b32 (k, k + 1), 1.14 words a lane", 6: "ds_read_b32 gather, 0.88 words a lane", 7: "one ds_read_b64 at a 4-byte-aligned address (k, k + 1), 1.14 words a lane"}
dur = {}
for f in glob.glob("$OUT/lds_floor/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): dur[r["Kernel_Name"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for f in glob.glob("$OUT/lds_floor/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)): agg[r["Kernel_Name"]][r["Counter_Name"]] = float(r["Counter_Value"])
    for k in sorted(agg):
        m = int(k.split("<")[1].split(">")[0]); c = agg[k]
        print("%-62s conflict/active %.3f  (cycles per LDS instruction: %.2f active, %.2f conflict)  %.0f us" % (names[m], c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"],
              c["SQ_LDS_IDX_ACTIVE"] / c["SQ_INSTS_LDS"], c["SQ_LDS_BANK_CONFLICT"] / c["SQ_INSTS_LDS"], dur.get(k, 0)))
PY
