#!/bin/bash
# HBM traffic of one 8192-sample step of the extended far-tail ladder (bench.py --workload $1 --tail-ratio 8) from PMC counters, separate
# passes.  Counter collection serialises the dispatches of all queues; the ladder's rungs run on stage streams with cross-stream event
# waits, and under that serialisation the run hung (rounds 2 and 3).  Here the engine runs on ONE stream for the passes (HCV_SERIAL=1:
# same kernels, same bytes), every pass under `timeout`.  PMC_STREAMS=1 tries the stage streams as well, under a short timeout, to
# record whether the hang is still there.  Usage on the GPU box: tools/pmc_ladder.sh c5
w=${1:-c5}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_ladder_$w
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  HCV_SERIAL=1 timeout ${PMC_TIMEOUT:-400} rocprofv3 --pmc $c --kernel-trace -d $out/$c -o run --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload $w --tail-ratio 8 --steps 64 --warmup 4 --batched-block 0 --realtime-block 0 --also "" --no-self-check > $out/$c.log 2>&1
  echo "$c pass (one stream): rc $?"
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $out/calib_$c -o run --output-format csv -- python $GRAFT_REPO_ROOT/tools/pmc_calib.py > $out/calib_$c.log 2>&1
done
if [ -n "$PMC_STREAMS" ]; then
  HCV_SERIAL=0 timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/streams -o run --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload $w --tail-ratio 8 --steps 64 --warmup 4 --batched-block 0 --realtime-block 0 --also "" --no-self-check > $out/streams.log 2>&1
  echo "FETCH_SIZE pass on the stage streams: rc $? (124 = killed by the timeout: hung)"
fi
cd $GRAFT_REPO_ROOT && python tools/pmc_ladder_parse.py $w $out
