#!/bin/bash
# the several-engines worker of tests/test_fused_nxm_gpu.py with the helping path forced: N runs for each engine count given (K=4 8 12 ...)
cd "$GRAFT_REPO_ROOT"
N=${N:-6}
for K in "$@"; do
  bad=0
  for i in $(seq $N); do
    r=$(env HCV_COOP_SPIN=0 $MANY_ENV timeout 600 python tests/_fused_nxm_worker.py 16 8 48000 16 many $K 2>/dev/null | tail -1)
    echo "$r" | grep -q '"all_same": true' || { bad=$((bad+1)); echo "   K=$K run $i: $(echo $r | cut -c1-160)"; }
  done
  echo "K=$K: $bad of $N runs wrong"
done
