#!/bin/bash
# quick FFT kernel tuning run on the GPU box: prints one compact line per case
python tests/perf/bench_fft.py --quick --reps 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: continue
    print(f\"{r['op']:5s} {r['precision']} 2^{r['log2n']:<2d} {r['ms']:8.3f} ms {r['achieved_GBps']:8.1f} GB/s {r['path']}\")
"
