#!/usr/bin/env python3
"""Turn the PMC passes of tools/pmc_traffic.sh into profiles/traffic_<workload>.json (read by bench.py).

HBM bytes per launch of the tail spectral_mac = FETCH_SIZE[KB] * 1024 * fetch_factor + WRITE_SIZE[KB] * 1024 * write_factor,
with the factors calibrated in the same run on a 1 GiB float4-coalesced copy (MI355X_MICROARCH.md §HBM: on gfx950
FETCH_SIZE reports half the bytes of a wide coalesced stream)."""
import collections
import csv
import json
import re
import sys

w = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else f"gpurun_out/pmc_{w}"
batched = len(sys.argv) > 3 and sys.argv[3] == "batched"       # the hop-tiled launch of 65536-sample calls instead of the single-hop one
offline = len(sys.argv) > 3 and sys.argv[3] == "offline"       # the matrix-core launch of 64-hop offline calls (hcv_mac_mfma.hip)
GiB = 1 << 30


def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[(re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return agg


out = {"workload": w, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/pmc_traffic.sh)"}
factors = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    cal = load(f"{src}/calib_{c}/run_counter_collection.csv")
    v = [x for (name, grid), vals in cal.items() if "copyBuffer" in name for x in vals]
    kb = sum(v) / len(v)
    factors[c] = GiB / (kb * 1024.0)
    out[f"calibration_{c}"] = {"known_bytes": GiB, "counter_kb": kb, "factor": round(factors[c], 4)}
tail = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    run = load(f"{src}/{c}/run_counter_collection.csv")
    # the steady-state tail launch: the unpredicated single-hop spectral_mac (<OT, 1, false, NT>) moving the most bytes
    # (... or the n x m block's multiply-accumulate launch, hcv_fused_nxm.hip, where the engine takes that block: c4s8, c4g)
    cand = {k: v for k, v in run.items() if ("spectral_mac_mfma_kernel" in k[0] if offline else "spectral_mac_tiled_kernel" in k[0] if batched else (("spectral_mac_kernel" in k[0] and ", 1, false," in k[0]) or "mac_meet_kernel" in k[0]))}
    # the head partition's MAC of whole-hop mode can share the tail's template variant and grid: the tail launches are the
    # ones near the largest value of the (name, grid) group that holds it
    key = max(cand, key=lambda k: max(cand[k]))
    # (offline: the ramp-up's launches run the same kernel over fewer partitions — only the launches with every partition live count)
    top = [v for v in cand[key] if v >= (0.97 if offline else 0.5) * max(cand[key])]
    tail[c] = sum(top) / len(top)
    if "mac_meet_kernel" in key[0]:
        # the n x m block: every launch of the group is the same steady-state block, but a launch whose forward launch came late does
        # the transforms itself (and polls while it waits) — under counter collection, which runs kernels one at a time, the first ones
        # do.  The typical launch is the median; the spread is kept beside it.
        vals = sorted(cand[key])
        top = vals
        tail[c] = vals[len(vals) // 2]
        out[f"{c}_kb_min_median_max"] = [vals[0], tail[c], vals[-1]]
    out[f"{c}_kb_per_launch"] = tail[c]
    out["kernel"] = key[0].replace("void ", "")
    if c == "FETCH_SIZE":
        out["launches_sampled"] = len(top)
out["hbm_read_bytes_per_launch"] = int(tail["FETCH_SIZE"] * 1024 * factors["FETCH_SIZE"])
out["hbm_write_bytes_per_launch"] = int(tail["WRITE_SIZE"] * 1024 * factors["WRITE_SIZE"])
out["hbm_bytes_per_launch"] = out["hbm_read_bytes_per_launch"] + out["hbm_write_bytes_per_launch"]
# which build the counters were read on: the commit (PMC_COMMIT, handed in by whoever starts the run: the GPU box has no .git) and the date
import datetime, os
out["commit"] = os.environ.get("PMC_COMMIT", "unknown")
out["date"] = datetime.date.today().isoformat()
json.dump(out, open(f"profiles/traffic_{w}{'_batched' if batched else '_offline' if offline else ''}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
