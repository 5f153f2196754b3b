#!/bin/bash
# sweep the spectral_mac launch plan on the GPU box: tools/sweep_mac.sh <workload> "<ots>" "<blocks>" "<nt modes>"
w=${1:-c5}; ots=${2:-"16 8"}; bl=${3:-"512 1024 2048"}; nts=${4:-"0"}
for nt in $nts; do for ot in $ots; do for blocks in $bl; do
  echo -n "nt=$nt ot=$ot blocks=$blocks  "
  HCV_MAC_NT=$nt HCV_MAC_OT=$ot HCV_MAC_BLOCKS=$blocks python bench.py --no-cpu-baseline --workload $w --steps 30 --warmup 3 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['value'], 'Msamples/s', d['ms_per_step'],'ms/step  mac', r['avg_launch_ms'],'ms', r['achieved'],'GB/s', r['kernel'][33:])
"
done; done; done
