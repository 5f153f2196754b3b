#!/bin/bash
# Round-5 evidence on the GPU box, reduced on the box into gpurun_out/r05/ (copy into profiles/): the driver-style default bench line (short
# line + side file), kernel-trace summaries of every timed shape at HEAD (c5, ns64, c4, one rank's share of strong-scaled c4 in both layouts —
# c4s8, c4g — and the launch-bound c3, c2, c1), the ladder's kernel summary, the fused kernels' durations for bench.py.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05
mkdir -p $OUT
python bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{' > $OUT/r05_c5_bench_default.json
cp gpurun_out/bench_details.json $OUT/r05_c5_bench_default_details.json 2>/dev/null
PREFIX=r05 WORKLOADS="${WORKLOADS:-c5 ns64 c4 c4s8 c4g c3 c2 c1}" BENCH_EXTRA="--no-all-cores --also= --extended-ratio 0" LATENCY_SPECS="" bash tools/refresh_profiles.sh > $OUT/refresh.log 2>&1
for f in gpurun_out/profiles_new/bench_*.json; do b=$(basename $f .json); mv $f $OUT/r05_${b#bench_}_bench.json; done
mv gpurun_out/profiles_new/* $OUT/ 2>/dev/null
python tools/kernel_us.py $OUT r05 > $OUT/kernel_us.json
# the extended ladder as the headline, under the kernel trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/prof_ladder; rm -rf $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python bench.py --workload c5 --tail-ratio 8 --steps 128 --warmup 8 --also "" --no-cpu-baseline --batched-block 0 --extended-ratio 0 --realtime-block 0 --no-self-check 2>/dev/null | grep '^{' > $OUT/r05_c5_ladder_bench_under_rocprof.json
T=$(find $D -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python tools/prof_summary.py "$T" 0.5 40 > $OUT/r05_c5_ladder_kernel_summary.txt
python bench.py --workload c5 --tail-ratio 8 --steps 128 --warmup 8 --also "" --no-all-cores --batched-block 0 --extended-ratio 0 --realtime-block 0 2>/dev/null | grep '^{' > $OUT/r05_c5_ladder_bench.json
rm -rf gpurun_out/prof_* gpurun_out/profiles_new
ls -la $OUT
