#!/bin/bash
# A/B builds of the library on the same box: tools/ab.sh <workload> <variant>...   (gpurun_ab/lib_<variant>.so)
W=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    cp gpurun_ab/lib_$v.so hisstools_library_amd/libhisstools_amd.so
    echo -n "$v: "; python tools/bench_line.py --workload $W --batched-block 0 --extended-ratio 0 2>&1 | cut -c1-150
  done
done
