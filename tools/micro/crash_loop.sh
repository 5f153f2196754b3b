#!/bin/bash
# how often does tests/test_gpu_parity.py (helping path forced) die, under the environments given: crash_loop.sh "A=1" "HCV_QUEUE_PROBE=0" ...   (N runs each)
cd "$GRAFT_REPO_ROOT"
N=${N:-6}
for e in "$@"; do
  bad=0
  for i in $(seq $N); do
    rm -f gpurun_out/crash_bt.txt
    env HCV_NATIVE_BACKTRACE=$PWD/gpurun_out/crash_bt.txt HCV_COOP_SPIN=0 $e python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "create_destroy or mono or partitioned or ntomono" > /tmp/cl.log 2>&1; rc=$?
    if [ $rc -ne 0 ]; then bad=$((bad+1)); echo "   $e run $i rc $rc: $(tail -1 /tmp/cl.log | cut -c1-90) | $(grep -m3 'hisstools_amd.so(_Z\|hisstools_amd.so(hcv' gpurun_out/crash_bt.txt 2>/dev/null | sed 's/.*so(//; s/+0x.*//' | tr '\n' ' ')"; fi
  done
  echo "$e: $bad of $N runs died"
done
