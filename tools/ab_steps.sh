#!/bin/bash
# The driver's short run (20 steps after 5) against longer ones and longer warm-ups, new default against HCV_SERIAL=0, one box, two rounds (c5).
for rep in 1 2; do
for cfg in "X=0 20 5" "HCV_SERIAL=0 20 5" "X=0 45 5" "X=0 20 30" "HCV_SERIAL=0 20 30"; do
  set -- $cfg
  echo -n "[$1 steps $2 warmup $3] "
  env $1 python bench.py --workload c5 --steps $2 --warmup $3 --no-cpu-baseline --no-all-cores --extended-ratio 0 --realtime-block 0 --batched-block 0 --offline-hops 0 --also= 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step  mac', d['roofline']['avg_launch_ms'])
"
done
done
