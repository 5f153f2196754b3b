#!/bin/bash
# A/B of library builds on the launch-bound workloads, one box, two rounds: tools/ab_small_libs.sh <variant> ...  (tools/micro/build/lib_<variant>.so)
for rep in 1 2; do
  for v in "$@"; do
    cp tools/micro/build/lib_$v.so hisstools_library_amd/libhisstools_amd.so
    for w in c1 c2 c3; do
      echo -n "[$v] $w: "
      python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --no-all-cores --extended-ratio 0 --realtime-block 0 --batched-block 0 --also= 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step')
"
    done
  done
done
