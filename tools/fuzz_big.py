#!/usr/bin/env python3
"""Which randomised cases come closest to the tolerance: tools/fuzz_big.py <first seed> <seconds> [threshold]  (prints the cases above it)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
import fuzz_parity as F
seed, secs = int(sys.argv[1]), float(sys.argv[2])
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 4e-6
t0, n = time.time(), 0
while time.time() - t0 < secs:
    kind, desc, e = F.one_case(seed)
    if e > thr: print(f"seed {seed} {kind} {desc.strip()} err {e:.2e}", flush=True)
    seed += 1; n += 1
print(n, "cases")
