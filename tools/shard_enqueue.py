#!/usr/bin/env python3
"""Host time per block of a sharded Convolver with and without its enqueue threads (hcv_shard_pool.h).

    python tools/shard_enqueue.py [--workload c4|c3] [--shards 8] [--steps 200]

Builds ONE object over `--shards` engines on the visible GPU(s) (device k % device_count: on the one-GPU test box all engines
share the GPU, which is the worst case for the threads — one runtime, one set of hardware queues), BASELINE config 4's matrix
(64x64, 2 s IRs: output rows split) or config 3's (8 -> 1, 5 s: inputs split + sum), streams 8192-sample device-pointer
blocks and reports, per mode: host microseconds per block spent inside process_dev (the enqueue), and the wall time per block.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4", choices=["c4", "c3"])
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    import numpy as np
    import torch
    import hisstools_library_amd as H

    nin, nout, L = {"c4": (64, 64, 96000), "c3": (8, 1, 240000)}[args.workload]
    B = 8192
    ndev = torch.cuda.device_count()
    devices = [k % ndev for k in range(args.shards)]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    xs = torch.rand((nin, 8 * B), generator=g, device=dev) * 2 - 1
    ys = torch.zeros((nout, 8 * B), device=dev)
    out = {"workload": args.workload, "matrix": [nin, nout], "ir_samples": L, "shards": args.shards, "devices": devices, "block": B, "steps": args.steps}
    for mode in ("0", "1"):
        os.environ["HCV_SHARD_THREADS"] = mode
        c = H.Convolver(nin, nout, 0, custom=(L, True, 256, 1024, 4096, 16384), maxBlock=B, devices=devices)
        h = torch.rand(L, generator=g, device=dev) * 2 - 1
        h = h / torch.linalg.vector_norm(h)
        torch.cuda.synchronize()
        for o in range(nout):
            for i in range(nin):
                assert c.set_dev(i, o, h.data_ptr(), L, True) == 0
        for k in range(L // B + 8):
            c.process_dev(xs.data_ptr() + 4 * (k % 8) * B, 8 * B, ys.data_ptr() + 4 * (k % 8) * B, 8 * B, nin, nout, B)
        c.synchronize()
        best = None
        for rep in range(3):
            t_enq = 0.0
            t0 = time.perf_counter()
            for k in range(args.steps):
                a = time.perf_counter()
                c.process_dev(xs.data_ptr() + 4 * (k % 8) * B, 8 * B, ys.data_ptr() + 4 * (k % 8) * B, 8 * B, nin, nout, B)
                t_enq += time.perf_counter() - a
            c.synchronize()
            wall = time.perf_counter() - t0
            r = {"enqueue_us_per_block": round(1e6 * t_enq / args.steps, 1), "wall_us_per_block": round(1e6 * wall / args.steps, 1)}
            if best is None or r["wall_us_per_block"] < best["wall_us_per_block"]:
                best = r
        # synchronous blocks (a caller that waits for every block): the enqueue is then on the critical path
        t0 = time.perf_counter()
        for k in range(50):
            c.process_dev(xs.data_ptr(), 8 * B, ys.data_ptr(), 8 * B, nin, nout, B, sync=True)
        best["sync_wall_us_per_block"] = round(1e6 * (time.perf_counter() - t0) / 50, 1)
        best["finite"] = bool(torch.isfinite(ys).all().item())
        out["threads" if mode == "1" else "serial"] = best
        del c
    print(json.dumps(out))


if __name__ == "__main__":
    main()
