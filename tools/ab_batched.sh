#!/bin/bash
# A/B of scheduler knobs on the multi-hop legs (65536-sample calls: 8 hops per launch; 64-hop offline calls), one box, two rounds:
#   tools/ab_batched.sh <workload> "<ENV=val ...>" ...
w=$1; shift
for rep in 1 2; do
  for envs in "$@"; do
    echo -n "[$envs] "
    env $envs python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-all-cores --extended-ratio 0 --realtime-block 0 --batched-block 65536 --offline-hops 64 --also= 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print(d['value'], 'Msamples/s | batched', c.get('batched_msamples_per_s'), c.get('batched_mac_hbm_frac'), '| offline', c.get('offline_msamples_per_s'), c.get('offline_frac'))
"
  done
done
