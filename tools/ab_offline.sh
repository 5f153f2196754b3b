#!/bin/bash
# A/B of library builds on the offline leg (64-hop calls on the matrix cores), one box, two rounds:
#   tools/ab_offline.sh <variant> ...   (gpurun_ab/lib_<variant>.so; workloads in $WL, default "c5 ns64 c4")
WL=${WL:-"c5 ns64 c4"}
for rep in 1 2; do
  for v in "$@"; do
    cp gpurun_ab/lib_$v.so hisstools_library_amd/libhisstools_amd.so
    for w in $WL; do
      echo -n "[$v] $w: "
      HCV_AB_OLD_LIBRARY=1 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-all-cores --batched-block 0 --extended-ratio 0 --realtime-block 0 --offline-hops 64 --also= 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        c=json.loads(l)['config']; print(c.get('offline_msamples_per_s'), 'Msamples/s', c.get('offline_bound'), c.get('offline_frac'), 'err', c.get('offline_max_rel_err'))
"
    done
  done
done
