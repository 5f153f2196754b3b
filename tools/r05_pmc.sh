#!/bin/bash
# Round 5: the HBM-traffic counter passes of every timed shape at ONE commit, stamped.
#   from the build container:  PMC_COMMIT=$(git rev-parse --short HEAD) gpurun --timeout 2400 -- "PMC_COMMIT=$PMC_COMMIT bash tools/r05_pmc.sh"
# writes profiles/traffic_{c5,ns64,c4,c4s8,c5_batched,c5_ladder}.json on the box under gpurun_out/pmc_json/ (copy them into profiles/).
cd "$GRAFT_REPO_ROOT"
export PMC_COMMIT=${PMC_COMMIT:-unknown}
mkdir -p gpurun_out/pmc_json
for w in ${PMC_WORKLOADS:-c5 ns64 c4 c4s8}; do
  bash tools/pmc_traffic.sh $w > gpurun_out/pmc_$w.log 2>&1
  python tools/pmc_parse.py $w > gpurun_out/pmc_json/parse_$w.log 2>&1 && cp profiles/traffic_$w.json gpurun_out/pmc_json/ && echo "$w ok" || { echo "$w FAILED"; tail -3 gpurun_out/pmc_json/parse_$w.log; }
done
if [ -z "$PMC_SKIP_BATCHED" ]; then
  PMC_BATCHED=65536 bash tools/pmc_traffic.sh c5 > gpurun_out/pmc_c5_batched.log 2>&1
  python tools/pmc_parse.py c5 gpurun_out/pmc_c5 batched > gpurun_out/pmc_json/parse_c5_batched.log 2>&1 && cp profiles/traffic_c5_batched.json gpurun_out/pmc_json/ && echo "c5 batched ok" || echo "c5 batched FAILED"
  # (the batched passes overwrite gpurun_out/pmc_c5: the single-hop figure was parsed and copied above)
fi
if [ -z "$PMC_SKIP_LADDER" ]; then
  bash tools/pmc_ladder.sh c5 > gpurun_out/pmc_ladder.log 2>&1 && cp profiles/traffic_c5_ladder.json gpurun_out/pmc_json/ && echo "c5 ladder ok" || echo "c5 ladder FAILED"
fi
for f in gpurun_out/pmc_json/traffic_*.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d.get("kernel","")[:60], d.get("commit"), d.get("date"), d.get("hbm_bytes_per_launch", d.get("hbm_bytes_per_step")))
PY
done
# (the raw counter dumps are tens of MB per workload: what travels back is the JSON and the logs' tails)
for d in gpurun_out/pmc_*; do [ -d "$d" ] && [ "$d" != gpurun_out/pmc_json ] && rm -rf "$d"; done
rm -rf gpurun_out/pmc_ladder_* 2>/dev/null
du -sh gpurun_out
