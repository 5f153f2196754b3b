#!/bin/bash
# quick hot-path check on the GPU box: one digest line per workload
for w in c5 ns64 c2 c3 c4; do python tools/bench_line.py --workload $w 2>&1 | cut -c1-260; done
