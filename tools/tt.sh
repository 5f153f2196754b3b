#!/bin/bash
for w in ns64 c4; do for t in 0 8 0 8; do echo -n "$w TT=$t: "; if [ $t = 0 ]; then python tools/bench_line.py --workload $w --extended-ratio 0 2>&1 | cut -c1-250; else HCV_MAC_TT=$t python tools/bench_line.py --workload $w --extended-ratio 0 2>&1 | cut -c1-250; fi; done; done
