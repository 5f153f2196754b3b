#!/bin/bash
# Round-6 evidence on the GPU box, reduced on the box into gpurun_out/r06/ (copy into profiles/): the driver-style default bench line (short
# line + side file), kernel-trace summaries of the timed shapes at HEAD (c5, ns64, c4, c4s8, c4g), and the OFFLINE leg's kernel summary
# (64-hop calls: spectral_mac_mfma_kernel) for c5, ns64 and c4.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06
mkdir -p $OUT
python bench.py --steps 20 --warmup 5 2>/dev/null | grep '^{' > $OUT/r06_c5_bench_default.json
cp gpurun_out/bench_details.json $OUT/r06_c5_bench_default_details.json 2>/dev/null
PREFIX=r06 WORKLOADS="${WORKLOADS:-c5 ns64 c4 c4s8 c4g}" BENCH_EXTRA="--no-all-cores --also= --extended-ratio 0 --offline-hops 0" LATENCY_SPECS="" bash tools/refresh_profiles.sh > $OUT/refresh.log 2>&1
for f in gpurun_out/profiles_new/bench_*.json; do b=$(basename $f .json); mv $f $OUT/r06_${b#bench_}_bench.json; done
mv gpurun_out/profiles_new/* $OUT/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in ${OFFLINE_WORKLOADS:-c5 ns64 c4}; do
  D=gpurun_out/prof_off_$w; rm -rf $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python bench.py --workload $w --steps 10 --warmup 2 --also "" --no-cpu-baseline --batched-block 0 --extended-ratio 0 --realtime-block 0 --no-self-check --offline-hops 64 2>/dev/null | grep '^{' > $OUT/r06_${w}_offline_bench_under_rocprof.json
  T=$(find $D -name "*kernel_trace.csv" | head -1)
  [ -n "$T" ] && python tools/prof_summary.py "$T" 0.5 0 > $OUT/r06_${w}_offline_kernel_summary.txt
  rm -rf $D
done
cp $OUT/r06_c5_offline_kernel_summary.txt $OUT/r06_offline_kernel_summary.txt 2>/dev/null
rm -rf gpurun_out/prof_* gpurun_out/profiles_new
ls -la $OUT
