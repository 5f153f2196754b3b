#!/bin/bash
# A/B of the hop-tiled multiply-accumulate tiles on one box: 4 x 8 two-bin tile (HCV_MAC_OT8=0) against the one-bin 8 x 8 tile at
# prefetch distances 1..3 (HCV_MAC_DIST), c5 / ns64-like shapes, two rounds.   tools/micro/ab_ot8.sh [outfile]
B=tools/micro/build/mac_bench
out=${1:-gpurun_out/ab_ot8.txt}
: > $out
for rep in 1 2; do
  for shape in "16 16 703 8 10" "64 64 58 8 10" "16 16 200 8 20" "32 32 130 8 10"; do
    echo "== $shape" >> $out
    HCV_MAC_OT8=0 $B $shape >> $out
    for d in 1 2 3; do HCV_MAC_OT8_MIN_P=32 HCV_MAC_DIST=$d $B $shape >> $out; done
  done
done
cat $out
