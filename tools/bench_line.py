#!/usr/bin/env python3
"""Run bench.py with the given args and print a one-line digest (value, ms/step, tail MAC, per-stage MAC ms)."""
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"] + sys.argv[1:], capture_output=True, text=True)
for l in out.stdout.splitlines():
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print(" ".join(sys.argv[1:]), "->", d["value"], "Msamples/s", d["ms_per_step"], "ms/step | tail mac", r["avg_launch_ms"], "ms", r["achieved"], "GB/s",
              r["kernel"][33:])
        break
else:
    print("FAILED", out.stdout[-2000:], out.stderr[-3000:])
