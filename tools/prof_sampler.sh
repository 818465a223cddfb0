#!/bin/bash
# rocprofv3 kernel stats of the end-to-end sampler run (device proposals)
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sampler -o s -- python3 $REPO/tools/sampler_bench.py --maxcall 400000 --modes device_chunks > $OUT/prof_sampler.log 2>&1
echo rc=$?
cd $REPO
python3 - <<PY
import glob
for f in glob.glob("$OUT/prof_sampler/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:2500])
PY
