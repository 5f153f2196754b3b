#!/bin/bash
ROWS="fft:f32:15,fft:f32:16,fft:f32:18,fft:f32:20,fft:f32:22,fft:f64:14,fft:f64:16,fft:f64:18,fft:f64:20,fft:f64:22,rfft:f32:16,rfft:f32:20,rifft:f32:16,rifft:f32:20,rfft:f64:18"
for c in 64 256 512 1024 2048; do
  echo "== chunk $c MiB"
  HCV_FX_STREAM=0 HCV_FX_TILE=0 HCV_FX_CHUNK_MB=$c timeout 200 python tests/perf/bench_fft.py --only $ROWS 2>/dev/null < /dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('  %s %s 2^%d %.4f ms %.0f GB/s' % (r['op'], r['precision'], r['log2n'], r['ms'], r['achieved_GBps']))"
done
