#!/bin/bash
# A/B builds of the library on the same box: tools/ab.sh "<bench args>" <variant>...   (gpurun_ab/lib_<variant>.so)
ARGS=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    cp gpurun_ab/lib_$v.so hisstools_library_amd/libhisstools_amd.so
    echo -n "$v: "; HCV_AB_OLD_LIBRARY=1 python tools/bench_line.py $ARGS 2>&1 | cut -c1-150
  done
done
