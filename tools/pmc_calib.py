#!/usr/bin/env python3
"""Known-byte-count streaming kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
(MI355X_MICROARCH.md §HBM: FETCH_SIZE under-reports wide coalesced reads; calibrate in your own access pattern).
Runs a 1 GiB float32 copy (reads 1 GiB, writes 1 GiB) a few times through torch."""
import torch
x = torch.empty(1 << 28, dtype=torch.float32, device="cuda").uniform_()
y = torch.empty_like(x)
torch.cuda.synchronize()
for _ in range(3):
    y.copy_(x)
torch.cuda.synchronize()
print("calib done")
