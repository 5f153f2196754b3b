#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) launch count, avg / min / total duration.
Usage: prof_summary.py <kernel_trace.csv> [min_total_ms]"""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
floor = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
agg = defaultdict(list)
for r in csv.DictReader(open(path)):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("hcv::", "")
    grid = (r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"))
    wg = r.get("Workgroup_Size_X", "?")
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[(name[:48], grid, wg, r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "?"))].append(dur)
tot = sum(sum(v) for v in agg.values())
print(f"{'kernel':48s} {'grid(threads)':>22s} {'wg':>4s} {'vgpr':>5s} {'agpr':>5s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'total_ms':>10s} {'pct':>6s}")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    t = sum(v)
    if t / 1e3 < floor:
        continue
    print(f"{k[0]:48s} {'x'.join(k[1]):>22s} {k[2]:>4s} {k[3]:>5s} {k[4]:>5s} {len(v):6d} {t/len(v):10.2f} {min(v):10.2f} {t/1e3:10.3f} {100*t/tot:6.2f}")
