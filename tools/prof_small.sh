#!/bin/bash
# Kernel-trace summary of a launch-bound workload under the given environment: tools/prof_small.sh <workload> <tag> [ENV=val ...]
# writes gpurun_out/prof_<workload>_<tag>.txt (per-kernel launch counts and durations of a 200-step bench run)
w=$1; tag=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/prof_small_tmp; rm -rf $D
env "$@" rocprofv3 --kernel-trace --output-format csv -d $D -- python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --extended-ratio 0 --realtime-block 0 --batched-block 0 --also "" > gpurun_out/prof_${w}_${tag}.log 2>&1
T=$(find $D -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py "$T" 0.05 > gpurun_out/prof_${w}_${tag}.txt
rm -rf $D
grep '^{' gpurun_out/prof_${w}_${tag}.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$w $tag', d['ms_per_step'], 'ms/step')
"
head -12 gpurun_out/prof_${w}_${tag}.txt
