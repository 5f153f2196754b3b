#!/bin/bash
# A/B of the single-workgroup transforms across library builds: tools/micro/run_fft.sh <variant>...  (tools/micro/build/lib_<variant>.so)
for v in "$@"; do
  cp tools/micro/build/lib_$v.so hisstools_library_amd/libhisstools_amd.so
  echo "== $v"
  for spec in "14 8" "14 1" "14 64" "12 4" "12 1" "8 64" "10 16"; do tools/micro/build/fft_bench $spec 300; done
done
