#!/usr/bin/env python3
"""profiles/kernel_us.json from the kernel summaries of the launch-bound workloads (tools/prof_summary.py output): for each of c1, c2, c3
the fused one-launch block kernel's average duration under rocprofv3 --kernel-trace — what bench.py quotes as roofline.avg_launch_ms of
those workloads (a kernel's duration cannot be measured from inside the run without lengthening the launch-bound chain).
Usage: kernel_us.py <dir with <prefix>_<w>_kernel_summary.txt> <prefix> > kernel_us.json"""
import json, os, sys
d, prefix = sys.argv[1], sys.argv[2]
out = {}
for w in ("c1", "c2", "c3"):
    path = os.path.join(d, f"{prefix}_{w}_kernel_summary.txt")
    if not os.path.exists(path):
        continue
    best = None
    for line in open(path):
        f = line.split()
        if len(f) >= 10 and f[0].startswith("fused_block"):
            calls, avg = int(f[-5]), float(f[-4])
            if best is None or calls > best[1]:
                best = (f[0], calls, avg)
    if best:
        out[w] = {"kernel": best[0].split("<")[0], "us": round(best[2], 2), "calls": best[1], "source": f"profiles/{prefix}_{w}_kernel_summary.txt"}
print(json.dumps(out, indent=1))
