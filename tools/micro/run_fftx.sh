#!/bin/bash
# A/B of the transform surface's rows across library builds: tools/micro/run_fftx.sh "<rows>" <variant>...  (tools/micro/build/lib_<variant>.so)
ROWS=$1; shift
for rep in 1 2; do
for v in "$@"; do
  cp tools/micro/build/lib_$v.so hisstools_library_amd/libhisstools_amd.so
  echo "== $v"
  python tests/perf/bench_fft.py --reps 3 --only "$ROWS" 2>&1 | cut -c1-118
done
done
