#!/bin/bash
# Round-2 evidence on the GPU box: bench lines, rocprofv3 kernel-trace summaries, PMC traffic (separate passes).  Everything is
# reduced on the box (raw traces are far larger than what gpurun brings back); results land in gpurun_out/r02/.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02
mkdir -p $OUT
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' > $OUT/r02_c5_bench_default.json
WORKLOADS="${WORKLOADS:-c5 ns64 c4 c3 c2}" BENCH_EXTRA="--no-all-cores" LATENCY_SPECS="" bash tools/refresh_profiles.sh > $OUT/refresh.log 2>&1
mv gpurun_out/profiles_new/* $OUT/ 2>/dev/null
mkdir -p profiles_tmp && cp profiles/*.json profiles_tmp/ 2>/dev/null
for w in ${PMC_WORKLOADS:-c5 c4 ns64}; do
  bash tools/pmc_traffic.sh $w > $OUT/pmc_$w.log 2>&1
  python tools/pmc_parse.py $w > $OUT/pmc_parse_$w.log 2>&1 && cp profiles/traffic_$w.json $OUT/
  rm -rf gpurun_out/pmc_$w
done
PMC_BATCHED=65536 bash tools/pmc_traffic.sh c5 > $OUT/pmc_c5_batched.log 2>&1
python tools/pmc_parse.py c5 gpurun_out/pmc_c5 batched > $OUT/pmc_parse_c5_batched.log 2>&1 && cp profiles/traffic_c5_batched.json $OUT/
rm -rf gpurun_out/pmc_c5 gpurun_out/prof_* gpurun_out/profiles_new
du -sh gpurun_out
ls $OUT
