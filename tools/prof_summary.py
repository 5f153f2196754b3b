#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) launch count, avg / min / total duration.
Usage: prof_summary.py <kernel_trace.csv> [min_total_ms] [timed_launches]
With timed_launches = K the summary ends with the average of the LAST K launches of the kernel with the largest total time:
bench.py times exactly its last K steps, so that line is the one to compare with the bench line's roofline.avg_launch_ms
(the per-kernel rows above also contain the priming and warm-up launches)."""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
floor = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
last_k = int(sys.argv[3]) if len(sys.argv) > 3 else 0
agg = defaultdict(list)
starts = defaultdict(list)
for r in csv.DictReader(open(path)):
    # (kernels of an anonymous namespace demangle as hcv::(anonymous namespace)::name<...>(...): that parenthesis is not the argument list)
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "").replace("hcv::", "")
    grid = (r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"))
    wg = r.get("Workgroup_Size_X", "?")
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    key = (name[:48], grid, wg, r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "?"))
    agg[key].append(dur)
    starts[key].append(int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
print(f"{'kernel':48s} {'grid(threads)':>22s} {'wg':>4s} {'vgpr':>5s} {'agpr':>5s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'total_ms':>10s} {'pct':>6s}")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    t = sum(v)
    if t / 1e3 < floor:
        continue
    print(f"{k[0]:48s} {'x'.join(k[1]):>22s} {k[2]:>4s} {k[3]:>5s} {k[4]:>5s} {len(v):6d} {t/len(v):10.2f} {min(v):10.2f} {t/1e3:10.3f} {100*t/tot:6.2f}")

if last_k:
    # The tail stage's steady-state multiply-accumulate (spectral_mac_*, or the n x m block's mac_meet_kernel)
    macs = sorted((t, k, i) for k, v in starts.items() if ("spectral_mac" in k[0] or "mac_meet_kernel" in k[0]) for i, t in enumerate(v))
    if macs:
        # the kernel that owns most of the run's final K such launches (the timed steps are the last thing the bench does; a ramp-up's
        # checked launches — longer each — come before them), and its last K launches
        from collections import Counter
        top = Counter(m[1] for m in macs[-last_k:]).most_common(1)[0][0]
        sel = [d for _, d in sorted(zip(starts[top], agg[top]))][-last_k:]
        print(f"\ntimed region: last {len(sel)} launches of {top[0]} grid {'x'.join(top[1])}: avg {sum(sel)/len(sel):.2f} us, min {min(sel):.2f} us, max {max(sel):.2f} us")
