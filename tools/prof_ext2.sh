#!/bin/bash
# Kernel-trace summary of the headline workload on the extended far-tail ladder: tools/prof_ext2.sh [workload] [steps]
w=${1:-c5}; steps=${2:-128}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/prof_ext_tmp; rm -rf $D
rocprofv3 --kernel-trace --output-format csv -d $D -- python bench.py --workload $w --tail-ratio 8 --steps $steps --warmup 8 --no-cpu-baseline --extended-ratio 0 --realtime-block 0 --batched-block 0 --also "" > gpurun_out/prof_ext_${w}.log 2>&1
T=$(find $D -name "*kernel_trace.csv" | head -1)
python tools/prof_summary.py "$T" 0.5 > gpurun_out/prof_ext_${w}.txt
rm -rf $D
grep '^{' gpurun_out/prof_ext_${w}.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$w extended', d['ms_per_step'], 'ms/step', d['value'], 'Msamples/s', [(s, round(v,3)) for s,v in d['roofline']['all_stage_mac_ms'].items()])
"
head -30 gpurun_out/prof_ext_${w}.txt
