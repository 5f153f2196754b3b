#!/bin/bash
for w in c2 c3; do for m in 0 1 0 1; do echo -n "$w one_stream=$m: "; HCV_ONE_STREAM=$m python tools/bench_line.py --workload $w --batched-block 0 --extended-ratio 0 2>&1 | cut -c1-120; done; done
