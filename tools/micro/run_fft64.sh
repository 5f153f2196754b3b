#!/bin/bash
for v in "$@"; do cp tools/micro/build/lib_$v.so hisstools_library_amd/libhisstools_amd.so; echo "== $v"; python tests/perf/bench_fft.py --reps 3 2>&1 | grep -E '"f64".*four-step' | cut -c1-120; done
python -m pytest tests/test_fft_surface.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "extended or rfft" 2>&1 | grep -E "passed|failed" | tail -2
