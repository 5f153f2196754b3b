#!/bin/bash
# A/B of environment settings on the launch-bound workloads, one box, self-check on: tools/ab_small.sh "<workloads>" "<ENV=val ...>" "<ENV=val ...>" ...
# (two rounds; prints ms per step bare, ms per step with the MAC's HIP events, max error against the reference)
ws=$1; shift
for rep in 1 2; do
  for w in $ws; do
    for envs in "$@"; do
      echo -n "$w [$envs] "
      env $envs timeout 200 python bench.py --workload $w --steps 200 --warmup 20 --no-all-cores --extended-ratio 0 --realtime-block 0 --batched-block 0 --also "" 2>/dev/null | python -c "
import json,sys
ok=False
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; ok=True
        print(d['ms_per_step'], 'ms/step', d['value'], 'Msamples/s | with events', r.get('profiled_ms_per_step'), '| mac', r['avg_launch_ms'], '| err', d['config']['max_rel_err'])
if not ok: print('FAILED')
"
    done
  done
done
