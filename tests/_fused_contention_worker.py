"""Worker of tests/test_fused_block_contention_gpu.py: K one-output engines, each driven by its own host thread with back-to-back
asynchronous hop-sized process_dev calls, all at once on the one GPU (optionally under a CU mask set by the parent).

argv: kind (nx1 | 1x1 | hops)  engines  seconds
Prints one JSON line: worst deviation from the oracle, whether every repetition of every engine gave the same bits, how many fused
launches ran, the time per block, and how long a bandwidth probe took (shows whether a CU mask is in force).
"""
import json
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import hisstools_library_amd as H
from oracle import oracle as O

kind, K, seconds = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
B, blocks = 8192, 32
n = B * blocks
dev = torch.device("cuda:0")

if kind == "nx1":               # BASELINE config 3's shape with 1 s impulse responses: 8 -> 1, zero latency
    nin, L = 8, 48000
    make = lambda: H.Convolver(nin, 1, 0, custom=(L, True, 256, 1024, 4096, 16384), maxBlock=B)
    ref = O.Convolver(nin, 1, 0)
    ref.setResetOffset(0)
elif kind == "1x1":             # zero-latency MonoConvolve-shaped engine, 1 s impulse response
    nin, L = 1, 48000
    make = lambda: H.Convolver(nin, 1, 0, custom=(L, True, 256, 1024, 4096, 16384), maxBlock=B)
    ref = O.Convolver(nin, 1, 0)
    ref.setResetOffset(0)
else:                           # BASELINE config 2's shape at 2 s: one 4096-point stage, four hops per block
    nin, L = 1, 96000
    make = lambda: H.Convolver(nin, 1, 0, custom=(L, False, 4096, 0, 0, 0), maxBlock=B)
    ref = O.PartitionedConvolve(4096, L, 0, 0)
    ref.setResetOffset(0)

xs = np.stack([O.synth_audio(i, n) for i in range(nin)])
irs = [O.synth_ir(i, 0, L) for i in range(nin)]
if kind == "hops":
    assert ref.set(irs[0]) == 0
    y_ref = ref.run(xs[0], 2048)
else:
    for i in range(nin):
        assert ref.set(i, 0, irs[i], True) == 0
    y_ref = ref.run(xs, 1, 2048)[0]

# bandwidth probe: a 256 MiB elementwise pass, warmed up (a CU mask shows here)
a = torch.zeros(1 << 26, device=dev)
for _ in range(3):
    a.add_(1.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    a.add_(1.0)
torch.cuda.synchronize()
probe_ms = (time.perf_counter() - t0) * 1e3 / 20
del a

xd = torch.from_numpy(xs).to(dev)
engines, outs = [], []
for k in range(K):
    c = make()
    for i in range(nin):
        assert c.set(i, 0, irs[i], True) == 0
    engines.append(c)
    outs.append(torch.zeros((1, n), device=dev))
torch.cuda.synchronize()

start = threading.Barrier(K)
result = [None] * K


def drive(k):
    c, ys = engines[k], outs[k]
    first, same, reps = None, True, 0
    start.wait()
    t_begin = time.perf_counter()
    while True:
        c.reset()
        for b in range(blocks):
            c.process_dev(xd.data_ptr() + 4 * b * B, n, ys.data_ptr() + 4 * b * B, n, nin, 1, B, sync=False)
        c.synchronize()
        y = ys[0].cpu().numpy()
        if first is None:
            first = y
        else:
            same = same and np.array_equal(first, y)
        reps += 1
        if time.perf_counter() - t_begin >= seconds:
            break
    dt = time.perf_counter() - t_begin
    fused = sum(s["fused_launches"] for s in c.stage_stats())
    result[k] = dict(err=float(np.abs(first - y_ref).max() / np.abs(y_ref).max()), same=bool(same), reps=reps, fused=int(fused),
                     ms_per_block=dt * 1e3 / (reps * blocks))


threads = [threading.Thread(target=drive, args=(k,)) for k in range(K)]
for t in threads:
    t.start()
for t in threads:
    t.join()
print(json.dumps(dict(kind=kind, engines=K, probe_ms=probe_ms, max_err=max(r["err"] for r in result), all_same=all(r["same"] for r in result),
                      reps=min(r["reps"] for r in result), fused_launches=min(r["fused"] for r in result),
                      ms_per_block=max(r["ms_per_block"] for r in result))))
