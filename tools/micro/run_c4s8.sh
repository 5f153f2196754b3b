#!/bin/bash
# one rank's share of c4 strong-scaled over 8 GPUs (64 inputs x 8 output rows) beside the whole matrix on one GPU, per call size
for b in 8192 16384 32768; do
  for w in c4s8 c4; do
  timeout 200 python bench.py --workload $w --block $b --steps 200 --warmup 20 --also "" --no-cpu-baseline --batched-block 0 --realtime-block 0 --extended-ratio 0 --no-self-check 2>/dev/null < /dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$w block $b:', r['value'], 'Msamples/s', r['ms_per_step'], 'ms per call')"
  done
done
