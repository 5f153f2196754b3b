#!/bin/bash
# Round-3 evidence on the GPU box, reduced on the box into gpurun_out/r03/: kernel-trace summaries of the default line and of the
# launch-bound workloads, PMC traffic of the headline kernel (separate passes), host enqueue time of the sharded object.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03
mkdir -p $OUT
PREFIX=r03 WORKLOADS="${WORKLOADS:-c5 c3 c2 c1}" BENCH_EXTRA="--no-all-cores --also=" LATENCY_SPECS="" bash tools/refresh_profiles.sh > $OUT/refresh.log 2>&1
mv gpurun_out/profiles_new/* $OUT/ 2>/dev/null
for w in c4 c3; do timeout 300 python tools/shard_enqueue.py --workload $w --shards 8 > $OUT/r03_shard_enqueue_$w.json 2> $OUT/shard_enqueue_$w.err; done
for w in ${PMC_WORKLOADS:-c5}; do
  bash tools/pmc_traffic.sh $w > $OUT/pmc_$w.log 2>&1
  python tools/pmc_parse.py $w > $OUT/pmc_parse_$w.log 2>&1 && cp profiles/traffic_$w.json $OUT/
  rm -rf gpurun_out/pmc_$w
done
rm -rf gpurun_out/prof_* gpurun_out/profiles_new
ls -la $OUT
