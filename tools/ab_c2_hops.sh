#!/bin/bash
# The fused multi-hop block of config 2 (fused_block_hops_kernel) against the four separate launches, one box, two rounds
for rep in 1 2; do
  for v in 1 0; do
    echo -n "[HCV_COOP=$v] "
    HCV_COOP=$v python bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline --no-all-cores --extended-ratio 0 --realtime-block 0 --batched-block 0 --also= 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step  self-check', d['config'].get('max_rel_err'))
"
  done
done
