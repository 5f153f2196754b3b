#!/usr/bin/env python3
"""Synchronous hop-sized host-pointer steps (staged, and on registered caller memory): tools/host_step.py <workload> [steps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hisstools_library_amd as H
from hisstools_library_amd._lib import f32p
import bench
w = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
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
lib = H.load()
xh = np.random.rand(nin, B).astype(np.float32); yh = np.zeros((nout, B), np.float32)
ih = (f32p * nin)(*[xh[i].ctypes.data_as(f32p) for i in range(nin)])
oh = (f32p * nout)(*[yh[o].ctypes.data_as(f32p) for o in range(nout)])
for _ in range(L // B + 4):
    lib.hcv_convolver_process_f32(conv.h, ih, oh, nin, nout, B)
for mode in ("staged", "registered"):
    if mode == "registered":
        H.host_register(xh); H.host_register(yh)
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        lib.hcv_convolver_process_f32(conv.h, ih, oh, nin, nout, B)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print(w, mode, "p50", round(float(np.median(ts)), 4), "min", round(float(ts.min()), 4), "ms/step")
