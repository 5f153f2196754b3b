#!/usr/bin/env python3
"""Show the last N launches of the biggest kernel in a rocprofv3 kernel trace with the gap to the previous one,
and what ran in a chosen gap.  Usage: trace_gaps.py <kernel_trace.csv> [N]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "").replace("hcv::", "")[:46]
rows.sort(key=lambda r: r["s"])
big = [r for r in rows if r["n"].startswith("spectral_mac_kernel<8, 1, false, true")]        # (the nontemporal single-hop tile: the timed launches)
big = big[-N:]
t0 = big[0]["s"]
prev = None
for r in big:
    gap = (r["s"] - prev["e"]) / 1e3 if prev else 0.0
    print(f"tail mac start {(r['s']-t0)/1e6:10.3f} ms dur {(r['e']-r['s'])/1e3:8.1f} us  gap-before {gap:10.1f} us  queue {r.get('Queue_Id')}")
    prev = r
# detail of the largest gap
gaps = [(big[i]["s"] - big[i-1]["e"], i) for i in range(1, len(big))]
g, i = max(gaps)
a, b = big[i-1]["e"], big[i]["s"]
print(f"--- kernels between tail mac {i-1} end and tail mac {i} start (gap {g/1e3:.1f} us):")
for r in rows:
    if r["e"] > a and r["s"] < b and r is not big[i] and r is not big[i-1]:
        print(f"   +{(r['s']-a)/1e3:9.1f} us  dur {(r['e']-r['s'])/1e3:8.1f} us  q{r.get('Queue_Id'):>3}  {r['n']}  grid {r.get('Grid_Size_X')}x{r.get('Grid_Size_Y')}x{r.get('Grid_Size_Z')}")
