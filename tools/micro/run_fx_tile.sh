#!/bin/bash
# A/B of the four-step passes: LDS-staged (HCV_FX_TILE=0) against register tiles, per size and scratch chunk
mkdir -p gpurun_out/fxtile
O=gpurun_out/fxtile
ROWS="fft:f32:16,fft:f32:17,fft:f32:18,fft:f32:19,fft:f32:20,fft:f32:21,rfft:f32:20,rifft:f32:20,rfft:f32:21"
(HCV_FX_TILE=0 timeout 300 python tools/micro/fx_tile_check.py --sizes 15,18,20 --batch 2 2>&1 | tail -4) > $O/check_old.txt
(timeout 600 python tools/micro/fx_tile_check.py 2>&1 | tail -60) > $O/check_new.txt
for t in 0 1; do
  for c in 64 128; do
    (HCV_FX_TILE=$t HCV_FX_CHUNK_MB=$c timeout 300 python tests/perf/bench_fft.py --only $ROWS --json $O/bench_t${t}_c${c}.json 2>&1 | tail -12) > $O/bench_t${t}_c${c}.txt
  done
done
tail -3 $O/check_old.txt $O/check_new.txt
for f in $O/bench_t*.txt; do echo "== $f"; cat $f; done
