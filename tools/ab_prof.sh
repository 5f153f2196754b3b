#!/bin/bash
# kernel-trace summaries of two library builds on the same box: tools/ab_prof.sh "<bench args>" <variant>...
ARGS=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  cp gpurun_ab/lib_$v.so hisstools_library_amd/libhisstools_amd.so
  D=gpurun_out/prof_ab_$v; rm -rf $D
  HCV_AB_OLD_LIBRARY=1 rocprofv3 --kernel-trace --output-format csv -d $D -- python bench.py --no-cpu-baseline $ARGS > gpurun_out/ab_prof_$v.log 2>&1
  T=$(find $D -name "*kernel_trace.csv" | head -1)
  python tools/prof_summary.py "$T" 0.3 > gpurun_out/ab_prof_$v.txt
  rm -rf $D
done
