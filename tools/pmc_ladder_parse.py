#!/usr/bin/env python3
"""profiles/traffic_<w>_ladder.json from the passes of tools/pmc_ladder.sh: HBM bytes of ONE 8192-sample step of the extended ladder —
every kernel between two emit launches, averaged over the last 48 steps — with the FETCH_SIZE / WRITE_SIZE factors calibrated in the same
run on a 1 GiB copy (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced stream)."""
import csv, json, re, sys
w, src = sys.argv[1], sys.argv[2]
GiB = 1 << 30
out = {"workload": w + " on the extended ladder (tail ratio 8)", "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, engine on one stream (tools/pmc_ladder.sh)"}
per_step = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    cal = [float(r["Counter_Value"]) for r in csv.DictReader(open(f"{src}/calib_{c}/run_counter_collection.csv")) if "copyBuffer" in r["Kernel_Name"]]
    factor = GiB / (sum(cal) / len(cal) * 1024.0)
    rows = sorted(csv.DictReader(open(f"{src}/{c}/run_counter_collection.csv")), key=lambda r: int(r["Dispatch_Id"]))
    emits = [k for k, r in enumerate(rows) if "emit_kernel" in r["Kernel_Name"]]
    steps = 48
    a, b = emits[-steps - 1], emits[-1]
    total = sum(float(r["Counter_Value"]) for r in rows[a + 1:b + 1])
    per_step[c] = total * 1024.0 * factor / steps
    by = {}
    for r in rows[a + 1:b + 1]:
        n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "").replace("hcv::", "")[:40]
        by[n] = by.get(n, 0.0) + float(r["Counter_Value"]) * 1024.0 * factor / steps
    out[f"{c}_factor"] = round(factor, 4)
    out[f"{c}_bytes_per_step_by_kernel"] = {k: int(v) for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:8]}
out["hbm_read_bytes_per_step"] = int(per_step["FETCH_SIZE"])
out["hbm_write_bytes_per_step"] = int(per_step["WRITE_SIZE"])
out["hbm_bytes_per_step"] = out["hbm_read_bytes_per_step"] + out["hbm_write_bytes_per_step"]
import datetime, os
out["kernel"] = "every launch of one 8192-sample step (dominant: " + next(iter(out["FETCH_SIZE_bytes_per_step_by_kernel"]), "?") + ")"
out["commit"] = os.environ.get("PMC_COMMIT", "unknown")
out["date"] = datetime.date.today().isoformat()
json.dump(out, open(f"profiles/traffic_{w}_ladder.json", "w"), indent=1)
print(json.dumps(out, indent=1))
