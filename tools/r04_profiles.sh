#!/bin/bash
# Round-4 evidence on the GPU box, reduced on the box into gpurun_out/r04/: the default bench line (short line + side file), kernel-trace
# summaries of every BASELINE shape at HEAD (c5, ns64, c4 and the launch-bound c3, c2, c1), the fused kernels' durations for bench.py.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04
mkdir -p $OUT
PREFIX=r04 WORKLOADS="${WORKLOADS:-c5 ns64 c4 c3 c2 c1}" BENCH_EXTRA="--no-all-cores --also=" LATENCY_SPECS="" bash tools/refresh_profiles.sh > $OUT/refresh.log 2>&1
mv gpurun_out/profiles_new/* $OUT/ 2>/dev/null
python tools/kernel_us.py $OUT r04 > $OUT/kernel_us.json
rm -rf gpurun_out/prof_* gpurun_out/profiles_new
ls -la $OUT
