#!/bin/bash
# a workload on the extended ladder at several rung ratios: ladder_ratio.sh <workload> <steps> <ratio> ...
cd "$GRAFT_REPO_ROOT"
W=$1; S=$2; shift; shift
for r in "$@"; do
    python bench.py --workload $W --tail-ratio $r --steps $S --warmup 8 --also "" --no-all-cores --no-cpu-baseline --batched-block 0 --extended-ratio 0 --realtime-block 0 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$W ratio $r:', d['value'], 'Msamples/s', d['ms_per_step'], 'ms', c['workload'][c['workload'].find('stages'):][:120], 'err', c.get('max_rel_err'))"
done
