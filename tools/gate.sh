#!/bin/bash
for w in ns64 c4 c5 c3; do for g in auto 0 auto 0; do echo -n "$w gate=$g: "; if [ $g = auto ]; then python tools/bench_line.py --workload $w --extended-ratio 0 2>&1 | cut -c1-120; else HCV_TAIL_GATE=$g python tools/bench_line.py --workload $w --extended-ratio 0 2>&1 | cut -c1-120; fi; done; done
