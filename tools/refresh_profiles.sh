#!/bin/bash
# Run on the GPU box: regenerates the judged evidence under gpurun_out/profiles_new/ (copy into profiles/ afterwards).
#   - default bench line (no profiler)                      -> bench_<w>.json
#   - rocprofv3 --kernel-trace --stats of the same command  -> <w>_kernel_stats.csv, <w>_kernel_summary.txt, <w>_bench_under_rocprof.json
set -u
OUT=gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in ${WORKLOADS:-c5 ns64 c4}; do
  python bench.py --workload $w ${BENCH_EXTRA:-} 2>/dev/null | grep '^{' > $OUT/bench_$w.json
  D=gpurun_out/prof_$w
  rm -rf $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python bench.py --workload $w --also "" --no-cpu-baseline --batched-block 0 --extended-ratio 0 --realtime-block 0 2>/dev/null | grep '^{' > $OUT/${PREFIX:-r02}_${w}_bench_under_rocprof.json
  T=$(find $D -name "*kernel_trace.csv" | head -1)
  S=$(find $D -name "*kernel_stats.csv" | head -1)
  [ -n "$S" ] && head -40 "$S" > $OUT/${PREFIX:-r02}_${w}_kernel_stats.csv
  [ -n "$T" ] && python tools/prof_summary.py "$T" 0.5 40 > $OUT/${PREFIX:-r02}_${w}_kernel_summary.txt
  rm -rf $D
done
for spec in ${LATENCY_SPECS:-"ns64 128 3" "ns64 64 3" "c4 128 6" "c5 256 2"}; do echo "latency $spec (SWAP_EVERY=37): $(SWAP_EVERY=37 python tools/latency.py $spec 2>&1 | grep -v amdgpu.ids | tr "\n" " ")" >> $OUT/latency.txt; done
