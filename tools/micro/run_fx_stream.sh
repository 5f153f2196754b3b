#!/bin/bash
# the one-launch four-step batch: correctness (against torch.fft in float64) and throughput beside the two-launch forms
O=gpurun_out/fxtile; mkdir -p $O
(timeout 200 python tools/micro/fx_tile_check.py --sizes ${SIZES:-16,17,18,19,20} --min-tiles 1100 2>&1 | grep -E "FFT  |IFFT|worst|rror|fault" ) > $O/stream_check.txt 2>&1 < /dev/null
cat $O/stream_check.txt
ROWS="fft:f32:16,fft:f32:17,fft:f32:18,fft:f32:19,fft:f32:20"
for v in ${VARIANTS:-"HCV_FX_LAG=8" "HCV_FX_LAG=12" "HCV_FX_LAG=16" "HCV_FX_STREAM=0 HCV_FX_TILE=1 HCV_FX_CHUNK_MB=1024" "HCV_FX_STREAM=0 HCV_FX_TILE=0 HCV_FX_CHUNK_MB=1024"}; do
  echo "== $v"
  env $v timeout 120 python tests/perf/bench_fft.py --only $ROWS 2>$O/err.txt < /dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('  2^%d %.4f ms %.0f GB/s' % (r['log2n'], r['ms'], r['achieved_GBps']))"
  grep -v amdgpu.ids $O/err.txt | tail -3
done
