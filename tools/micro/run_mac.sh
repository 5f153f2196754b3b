#!/bin/bash
# A/B of spectral_mac builds on the GPU box: tools/micro/build/lib_<variant>.so, tools/micro/build/mac_bench (built by the caller)
B=tools/micro/build/mac_bench
for v in "$@"; do
  cp tools/micro/build/lib_$v.so hisstools_library_amd/libhisstools_amd.so
  echo "== $v"
  $B 16 16 703 8 10
  HCV_MAC_PREFETCH=0 $B 16 16 703 8 10
  $B 64 64 58 8 10
  $B 64 64 11 8 20
  HCV_MAC_BLOCKS=512 $B 16 16 703 8 10
  $B 16 16 703 16 6
  $B 16 16 703 4 10
  $B 16 16 703 2 10
  $B 16 16 703 1 20
  $B 64 64 58 1 20
  $B 64 64 11 1 40
done
