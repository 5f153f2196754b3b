#!/bin/bash
# HBM traffic of the tail spectral_mac launch from PMC counters (PMC_BATCHED=65536: of the hop-tiled launch of batched calls too; PMC_OFFLINE=64: of the
# matrix-core launch of 64-hop offline calls), collected in separate passes (no tracing domains
# beyond --kernel-trace; the headline leg only — with the extended-ladder leg in the same process counter collection crashed or
# hung on this ROCm stack, so it is switched off here and every pass runs under `timeout`).  Usage on the GPU box: tools/pmc_traffic.sh <workload> ; writes gpurun_out/pmc_<workload>/
w=${1:-c5}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$w
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout ${PMC_TIMEOUT:-600} rocprofv3 --pmc $c --kernel-trace -d $out/$c -o run --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload $w --steps 4 --warmup 1 --batched-block ${PMC_BATCHED:-0} --offline-hops ${PMC_OFFLINE:-0} --extended-ratio 0 --realtime-block 0 --no-self-check > $out/$c.log 2>&1
  timeout ${PMC_TIMEOUT:-300} rocprofv3 --pmc $c --kernel-trace -d $out/calib_$c -o run --output-format csv -- python $GRAFT_REPO_ROOT/tools/pmc_calib.py > $out/calib_$c.log 2>&1
done
ls -R $out | head -40
