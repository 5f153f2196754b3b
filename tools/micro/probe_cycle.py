"""create / destroy cycles of a small several-output engine in a process that holds other streams: does the hardware-queue experiment's
replacement path (hcv_queue_probe.hip) survive them?   python tools/micro/probe_cycle.py [held streams] [cycles]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import hisstools_library_amd as H
from oracle import oracle as O

held_n, cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 5, int(sys.argv[2]) if len(sys.argv) > 2 else 200
held = [torch.cuda.Stream() for _ in range(held_n)]
for s in held:
    with torch.cuda.stream(s):
        torch.zeros(8, device="cuda").add_(1)
torch.cuda.synchronize()
h, x = O.synth_ir(1, 1, 30000), O.synth_audio(1, 4096)
xs = np.stack([x, x])
for k in range(cycles):
    c = H.Convolver(2, 2, k % 3)
    for i in range(2):
        assert c.set(i, i, h[: 1000 + 97 * k], True) == 0
    c.run(xs, 2, 512)
    assert c.set(0, 1, h, True) == 0
    c.run(xs, 2, 4096)
    del c
    if k % 7 == 0:
        held.append(torch.cuda.Stream())
        with torch.cuda.stream(held[-1]):
            torch.zeros(8, device="cuda").add_(1)
print("ok", cycles)
