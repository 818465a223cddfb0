#!/bin/bash
# round-6 evidence in one session: the default bench (200-step blocks), rocprofv3 stats + PMC passes for every measured configuration,
# cycle stamps, the transform probes.  Copies nothing: gpurun_out/r6_* -> profiles/ by hand afterwards.
TAG=${1:-r6}
OUT=$PWD/gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
python bench.py > $OUT/${TAG}_bench_default.out 2> $OUT/${TAG}_bench_default.err; cp bench_detail.json $OUT/${TAG}_bench_default.json; tail -c 600 $OUT/${TAG}_bench_default.out
bash tools/gpu_profiles_all.sh $TAG 2>&1 | tail -40
timeout 600 python tools/post_stamps.py > $OUT/${TAG}_c2_cycle_stamps.txt 2>&1; tail -40 $OUT/${TAG}_c2_cycle_stamps.txt
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o tools/exp/wave_local_probe tools/exp/wave_local_probe.hip 2>/dev/null; timeout 120 tools/exp/wave_local_probe > $OUT/${TAG}_wave_local_probe.txt 2>&1; cat $OUT/${TAG}_wave_local_probe.txt
