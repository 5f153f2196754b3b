#!/usr/bin/env python3
"""Per-call latency of synchronous small-block process() calls (real-time usage): tools/latency.py <workload> <block> [hops]
Prints mean / p50 / p99 / max milliseconds per call and the real-time budget of the block at the workload's sample rate.
SWAP_EVERY=K replaces the IR of one (in, out) pair every K calls (a live IR swap: set_dev between two process calls) and
reports the set() time and the latency of the call that follows (retiring + ghost spectra of the exact restart)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import hisstools_library_amd as H
import bench

w, B = sys.argv[1], int(sys.argv[2])
hops = int(sys.argv[3]) if len(sys.argv) > 3 else 3
nin, nout, L, fs, layout = bench.WORKLOADS[w]
dev = torch.device("cuda", 0)
TAIL_RATIO = int(os.environ.get("TAIL_RATIO", "0"))       # the same workload on the extended far-tail ladder
conv = H.Convolver(nin, nout, 0, device=0, maxBlock=max(B, 8192), custom=(L, *layout), tailRatio=TAIL_RATIO)
g = torch.Generator(device=dev)
decay = torch.pow(torch.tensor(10.0, device=dev), -3.0 * torch.arange(L, device=dev, dtype=torch.float32) / L)
for o in range(nout):
    for i in range(nin):
        g.manual_seed(1000 * i + o + 1)
        h = (torch.rand(L, generator=g, device=dev) * 2 - 1) * decay
        h = h / torch.linalg.vector_norm(h)
        torch.cuda.synchronize()
        assert conv.set_dev(i, o, h.data_ptr(), L, True) == 0
tail = [s for s in layout[1:] if s][-1]
ncalls = min(hops * (tail // 2) // B, int(os.environ.get("MAX_CALLS", "1000000")))
xs = torch.rand((nin, B), device=dev) * 2 - 1
ys = torch.zeros((nout, B), device=dev)
torch.cuda.synchronize()
# prime: run a full IR length so every partition is live
prime = (L // (tail // 2) + 2) * (tail // 2) // 8192
big = torch.rand((nin, 8192), device=dev) * 2 - 1
bigy = torch.zeros((nout, 8192), device=dev)
for _ in range(prime):
    conv.process_dev(big.data_ptr(), 8192, bigy.data_ptr(), 8192, nin, nout, 8192)
conv.synchronize()
ts, set_ms, after = [], [], []
swap_every = int(os.environ.get("SWAP_EVERY", "0"))
paced = os.environ.get("PACED", "1") != "0"      # real-time pacing: call k is issued no earlier than k * B / fs
t_start = time.perf_counter()
for k in range(ncalls):
    if paced:
        while time.perf_counter() < t_start + k * B / fs:
            pass
    if swap_every and k and k % swap_every == 0:
        t0 = time.perf_counter()
        assert conv.set_dev((k // swap_every) % nin, (k // swap_every * 7) % nout, h.data_ptr(), L, True) == 0
        set_ms.append((time.perf_counter() - t0) * 1e3)
        after.append(k)
    t0 = time.perf_counter()
    conv.process_dev(xs.data_ptr(), B, ys.data_ptr(), B, nin, nout, B, sync=True)
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print(f"{w}{' ladder x' + str(TAIL_RATIO) if TAIL_RATIO else ''} block={B} paced={int(paced)} defer={os.environ.get('HCV_DEFER','1')}: calls={ncalls} mean={ts.mean():.3f} p50={np.percentile(ts,50):.3f} p99={np.percentile(ts,99):.3f} "
      f"max={ts.max():.3f} ms | budget {1e3*B/fs:.3f} ms | sum={ts.sum():.1f} ms for {1e3*ncalls*B/fs:.1f} ms of audio")
thr = float(os.environ.get("SLOW_MS", "0.6"))
slow = [(i, round(float(t), 3)) for i, t in enumerate(ts) if t > thr]
print(f"   calls > {thr} ms:", slow[:48])
if set_ms:
    a = ts[after]
    print(f"   live swaps: {len(set_ms)}; set() mean {np.mean(set_ms):.3f} max {np.max(set_ms):.3f} ms; the call after a swap mean {a.mean():.3f} max {a.max():.3f} ms")
