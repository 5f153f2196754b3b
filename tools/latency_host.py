#!/usr/bin/env python3
"""Per-call latency of paced small-block calls through HOST pointers (hcv_convolver_process_f32: what HISSTools::Convolver::process
does): tools/latency_host.py <workload> <block> [seconds].  Prints p50 / p99 / max per call class (which stages' hops the call completes)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import hisstools_library_amd as H
from hisstools_library_amd._lib import f32p
import bench

w, B = sys.argv[1], int(sys.argv[2])
seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
nin, nout, L, fs, layout = bench.WORKLOADS[w]
dev = torch.device("cuda", 0)
conv = H.Convolver(nin, nout, 0, device=0, maxBlock=8192, custom=(L, *layout))
h = (torch.rand(L, device=dev) * 2 - 1) * 1e-3
for o in range(nout):
    for i in range(nin):
        torch.cuda.synchronize(); assert conv.set_dev(i, o, h.data_ptr(), L, True) == 0
lib = H.load()
xin = np.random.RandomState(1).uniform(-1, 1, (nin, B)).astype(np.float32)
yout = np.zeros((nout, B), np.float32)
ip = (f32p * nin)(*[xin[i].ctypes.data_as(f32p) for i in range(nin)])
op = (f32p * nout)(*[yout[o].ctypes.data_as(f32p) for o in range(nout)])
big = torch.rand((nin, 8192), device=dev); bigy = torch.zeros((nout, 8192), device=dev)
for _ in range(L // 8192 + 2): conv.process_dev(big.data_ptr(), 8192, bigy.data_ptr(), 8192, nin, nout, 8192)
conv.synchronize()
n = int(seconds * fs / B)
ts = np.zeros(n)
for _ in range(8): lib.hcv_convolver_process_f32(conv.h, ip, op, nin, nout, B)
t_start = time.perf_counter()
for k in range(n):
    while time.perf_counter() < t_start + k * B / fs: pass
    t0 = time.perf_counter()
    assert lib.hcv_convolver_process_f32(conv.h, ip, op, nin, nout, B) == 0
    ts[k] = time.perf_counter() - t0
ts *= 1e3
print(f"{w} block={B} host pointers: calls={n} mean={ts.mean():.3f} p50={np.percentile(ts,50):.3f} p99={np.percentile(ts,99):.3f} max={ts.max():.3f} ms | budget {1e3*B/fs:.3f} ms")
idx = (np.arange(n) + 8 + 1) * B
for name, sel in (("plain", idx % 128 != 0), ("128", (idx % 128 == 0) & (idx % 512 != 0)), ("512", (idx % 512 == 0) & (idx % 2048 != 0)), ("2048", idx % 2048 == 0)):
    if sel.any(): print(f"   {name:6s}: {int(sel.sum()):5d} calls p50 {np.percentile(ts[sel],50):.3f} p99 {np.percentile(ts[sel],99):.3f} max {ts[sel].max():.3f}")
print("   slowest:", [(int(i), round(float(ts[i]), 3)) for i in np.argsort(ts)[-10:]])
