#!/bin/bash
# A/B of bench workloads across library builds on one box: tools/micro/run_bench_ab.sh "<workloads>" <variant>...
WL=$1; shift
for rep in 1 2; do
for v in "$@"; do
  cp tools/micro/build/lib_$v.so hisstools_library_amd/libhisstools_amd.so
  for w in $WL; do
    python bench.py --workload $w --no-cpu-baseline --batched-block 0 --extended-ratio 0 --realtime-block 0 --no-self-check --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', '$w', d['value'], d['ms_per_step'], d['roofline'].get('frac'))"
  done
done
done
