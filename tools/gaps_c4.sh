#!/bin/bash
# kernel trace of a short bench run + the launches between two tail MACs (tools/trace_gaps.py)
w=${1:-c4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/prof_gaps_$w
rm -rf $D
rocprofv3 --kernel-trace --output-format csv -d $D -- python bench.py --workload $w --no-cpu-baseline --batched-block 0 --extended-ratio 0 --realtime-block 0 --steps 20 --warmup 5 2>/dev/null | grep '^{' | cut -c1-200
T=$(find $D -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py "$T" 8
python tools/prof_summary.py "$T" 0.2 20 | head -30
rm -rf $D
