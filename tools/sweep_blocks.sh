#!/bin/bash
for b in 768 1024 1536 2304 3072; do for rep in 1 2; do echo -n "blocks=$b: "; HCV_MAC_BLOCKS=$b python tools/bench_line.py --workload ${1:-ns64} --extended-ratio 0 2>&1 | cut -c1-130; done; done
