#!/usr/bin/env python3
"""Host enqueue time against completion time of paced small calls: tools/hostq_small.py <workload> <block>
For every call: t_enq = time inside process_dev(sync=False), t_done = until synchronize() returns.  Prints both for the
plain calls (median) and for the calls that carry stage boundaries (every 512 / 2048 samples)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import hisstools_library_amd as H
import bench

w, B = sys.argv[1], int(sys.argv[2])
nin, nout, L, fs, layout = bench.WORKLOADS[w]
dev = torch.device("cuda", 0)
conv = H.Convolver(nin, nout, 0, device=0, maxBlock=8192, custom=(L, *layout))
h = (torch.rand(L, device=dev) * 2 - 1) * 1e-3
for o in range(nout):
    for i in range(nin):
        torch.cuda.synchronize(); assert conv.set_dev(i, o, h.data_ptr(), L, True) == 0
xs = torch.rand((nin, B), device=dev); ys = torch.zeros((nout, B), device=dev)
big = torch.rand((nin, 8192), device=dev); bigy = torch.zeros((nout, 8192), device=dev)
for _ in range(L // 8192 + 2): conv.process_dev(big.data_ptr(), 8192, bigy.data_ptr(), 8192, nin, nout, 8192)
conv.synchronize()
n = 2 * 8192 // B
enq, done = [], []
t_start = time.perf_counter()
for k in range(n):
    while time.perf_counter() < t_start + k * B / fs: pass
    t0 = time.perf_counter()
    conv.process_dev(xs.data_ptr(), B, ys.data_ptr(), B, nin, nout, B)
    t1 = time.perf_counter()
    conv.synchronize()
    t2 = time.perf_counter()
    enq.append(1e6 * (t1 - t0)); done.append(1e6 * (t2 - t0))
enq, done = np.array(enq), np.array(done)
idx = np.arange(n)
for name, sel in (("plain calls", ((idx + 1) * B) % 128 != 0), ("128-boundaries only", (((idx + 1) * B) % 128 == 0) & (((idx + 1) * B) % 512 != 0)),
                  ("512-boundaries", (((idx + 1) * B) % 512 == 0) & (((idx + 1) * B) % 2048 != 0)), ("2048-boundaries", ((idx + 1) * B) % 2048 == 0)):
    sel = sel & (idx > 0)
    if sel.any():
        print(f"{w} B={B} {name:22s}: {int(sel.sum()):4d} calls, enqueue median {np.median(enq[sel]):7.1f} us, until done median {np.median(done[sel]):7.1f} us (max {done[sel].max():7.1f})")
