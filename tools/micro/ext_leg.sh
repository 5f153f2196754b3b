#!/bin/bash
# the extended ladder as the SECOND engine of a bench process (the default line's in-process leg), under the environments given:
#   ext_leg.sh "HCV_X=1" "HCV_NXM_LADDER=1"        (STEPS=20 by default)
cd "$GRAFT_REPO_ROOT"
for e in "$@"; do
    env $e python bench.py --workload c5 --extended-ratio 8 --steps ${STEPS:-20} --warmup 5 --also "" --no-all-cores --no-cpu-baseline --batched-block 0 --realtime-block 0 > /dev/null 2>&1
    python - "$e" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_details.json"))
ex = d["config"]["extended_layout"]
print(sys.argv[1], "headline", d["ms_per_step"], "extended", ex.get("ms_per_step"), ex.get("msamples_per_s"), "pivot mac", ex["all_stage_mac_ms_per_step"].get("16384"), "frac", ex["roofline_step"]["frac"])
PY
done
