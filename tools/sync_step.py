#!/usr/bin/env python3
"""Synchronous against asynchronous hop-sized steps on HBM-resident audio: tools/sync_step.py <workload>.
What a caller that waits for every block pays beyond the pipelined rate (launch chain latency, idle gaps)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hisstools_library_amd as H
import bench
w = sys.argv[1]
nin, nout, L, fs, layout = bench.WORKLOADS[w]
dev = torch.device("cuda", 0)
conv = H.Convolver(nin, nout, 0, device=0, maxBlock=8192, custom=(L, *layout))
g = torch.Generator(device=dev); g.manual_seed(1)
for o in range(nout):
    for i in range(nin):
        h = torch.rand(L, generator=g, device=dev) * 2 - 1
        torch.cuda.synchronize()
        assert conv.set_dev(i, o, h.data_ptr(), L, True) == 0
B = 8192
xs = torch.rand((nin, B), device=dev); ys = torch.zeros((nout, B), device=dev)
torch.cuda.synchronize()
for _ in range(L // B + 4):
    conv.process_dev(xs.data_ptr(), B, ys.data_ptr(), B, nin, nout, B)
conv.synchronize()
for mode in ("async", "sync", "async", "sync"):
    t0 = time.perf_counter()
    for _ in range(40):
        conv.process_dev(xs.data_ptr(), B, ys.data_ptr(), B, nin, nout, B, sync=(mode == "sync"))
    conv.synchronize()
    print(w, mode, round(1e3 * (time.perf_counter() - t0) / 40, 4), "ms/step")
