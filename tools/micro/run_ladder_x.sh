#!/bin/bash
for v in "HCV_X_MINK=8" "HCV_X_MINK=4" "HCV_X_MINK=2" "HCV_X_MAXSPLIT=8" "HCV_X_MAXSPLIT=4" "HCV_X_MINK=8"; do
  env $v timeout 200 python bench.py --workload c5 --tail-ratio 8 --steps 256 --warmup 16 --also "" --no-cpu-baseline --batched-block 0 --realtime-block 0 --extended-ratio 0 --no-self-check 2>/dev/null < /dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$v', r['value'], r['ms_per_step'])"
done
