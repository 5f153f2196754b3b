#!/bin/bash
# a workload on the extended ladder as the process's ONLY engine, under the environments given: ladder_ab.sh <workload> <steps> "ENV=.." ...
cd "$GRAFT_REPO_ROOT"
W=$1; S=$2; shift; shift
for e in "$@"; do
    env $e python bench.py --workload $W --tail-ratio 8 --steps $S --warmup 8 --also "" --no-all-cores --no-cpu-baseline --batched-block 0 --extended-ratio 0 --realtime-block 0 --no-self-check 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W ladder', '$e', d['value'], d['ms_per_step'])"
done
