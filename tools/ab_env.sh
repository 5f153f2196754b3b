#!/bin/bash
# A/B of scheduler knobs on ONE box: tools/ab_env.sh <workload> "<ENV=val ...>" "<ENV=val ...>" ...   (two rounds, short bench lines)
w=$1; shift
for rep in 1 2; do
  for envs in "$@"; do
    echo -n "[$envs] "
    env $envs python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline --extended-ratio 0 --realtime-block 0 --batched-block 0 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step  mac', d['roofline']['avg_launch_ms'])
"
  done
done
