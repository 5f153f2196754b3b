#!/bin/bash
# Round 6: the HBM-traffic counter passes of the timed shapes at ONE commit, stamped, plus the offline (matrix-core) launch of c5.
#   from the build container:  PMC_COMMIT=$(git rev-parse --short HEAD) gpurun --timeout 2400 -- "PMC_COMMIT=$PMC_COMMIT bash tools/r06_pmc.sh"
# writes traffic_{c5,ns64,c4,c4s8,c5_offline}.json under gpurun_out/pmc_json/ (copy them into profiles/).
cd "$GRAFT_REPO_ROOT"
export PMC_COMMIT=${PMC_COMMIT:-unknown}
mkdir -p gpurun_out/pmc_json
for w in ${PMC_WORKLOADS:-c5 ns64 c4 c4s8}; do
  bash tools/pmc_traffic.sh $w > gpurun_out/pmc_$w.log 2>&1
  python tools/pmc_parse.py $w > gpurun_out/pmc_json/parse_$w.log 2>&1 && cp profiles/traffic_$w.json gpurun_out/pmc_json/ && echo "$w ok" || { echo "$w FAILED"; tail -3 gpurun_out/pmc_json/parse_$w.log; }
done
for w in ${PMC_OFFLINE_WORKLOADS:-c5}; do
  PMC_OFFLINE=64 bash tools/pmc_traffic.sh $w > gpurun_out/pmc_${w}_offline.log 2>&1
  python tools/pmc_parse.py $w gpurun_out/pmc_$w offline > gpurun_out/pmc_json/parse_${w}_offline.log 2>&1 && cp profiles/traffic_${w}_offline.json gpurun_out/pmc_json/ && echo "$w offline ok" || { echo "$w offline FAILED"; tail -3 gpurun_out/pmc_json/parse_${w}_offline.log; }
done
for f in gpurun_out/pmc_json/traffic_*.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d.get("kernel","")[:60], d.get("commit"), d.get("date"), d.get("hbm_bytes_per_launch", d.get("hbm_bytes_per_step")))
PY
done
for d in gpurun_out/pmc_*; do [ -d "$d" ] && [ "$d" != gpurun_out/pmc_json ] && rm -rf "$d"; done
du -sh gpurun_out
