#!/bin/bash
# Short bench lines of the launch-bound workloads (c1, c2, c3): tools/small_lines.sh [outdir]
out=${1:-gpurun_out/small}; mkdir -p $out
for w in c2 c3 c1; do
  timeout 150 python bench.py --workload $w --no-all-cores --extended-ratio 0 --realtime-block 0 --also "" --steps 200 --warmup 20 2>/dev/null | grep '^{' > $out/${w}.json
  python - $out/${w}.json <<'P'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(d["config"]["workload"][:3], "ms/step", d["ms_per_step"], "value", d["value"], "profiled", r.get("profiled_ms_per_step"), "mac", r["avg_launch_ms"], "err", d["config"]["max_rel_err"])
P
done
