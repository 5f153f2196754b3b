#!/usr/bin/env python3
"""Throughput of the spectral IR functions on HBM-resident spectra (hcv_ir_exec_dev), the GPU form of the reference's
IR_Manipulation_Tester ("- Test/IR_Manipulation_Tester/.../main.cpp": ir_phase at fft_log2 = 14 in the eight
Zero/Center x Mix/Min/Max/Lin modes), with the reference's CPU time beside it.

A batch of spectra big enough to defeat the 256 MiB Infinity Cache (`--gib` of operands) is processed in place in one
call; time = HIP events on the launch stream, best of --reps.  Algorithmic bytes = one read + one write of the spectra.

    python tests/perf/bench_ir.py [--json profiles/r01_ir_functions.json] [--cpu]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hisstools_library_amd.spectral_functions as S  # noqa: E402

MODES = [("Zero Mix", 0.1, True), ("Center Mix", 0.9, False), ("Zero Min", 0.0, True), ("Center Min", 0.0, False),
         ("Zero Max", 1.0, True), ("Center Max", 1.0, False), ("Zero Lin", 0.5, True), ("Center Lin", 0.5, False)]


def run(op, prec, l2, value, zero, gib, reps):
    real = 4 if prec == "f32" else 8
    tdt = torch.float32 if prec == "f32" else torch.float64
    half = (1 << l2) >> 1
    per = 2 * half * real
    batch = max(1, int(gib * (1 << 30)) // per)
    # plausible spectra: magnitudes away from zero so that log / exp stay in range
    re = torch.rand(batch * half, device="cuda", dtype=tdt) + 0.5
    im = torch.rand(batch * half, device="cuda", dtype=tdt) - 0.5
    src_r, src_i = re.clone(), im.clone()
    st = torch.cuda.current_stream().cuda_stream
    best = 1e30
    for _ in range(reps + 1):
        re.copy_(src_r)
        im.copy_(src_i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        S.exec_dev(op, 0 if prec == "f32" else 1, l2, batch, re.data_ptr(), im.data_ptr(), re.data_ptr(), im.data_ptr(), half, half, value, zero, st, False)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    gbs = (1.0 if op == S.IrOp.SPIKE else 2.0) * batch * per / (best * 1e-3) / 1e9      # a spike is written, not read
    return {"batch": batch, "ms": round(best, 4), "achieved_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / 8000.0, 4),
            "spectra_per_s": round(batch / (best * 1e-3), 1), "finite": bool(torch.isfinite(re).all() and torch.isfinite(im).all())}


def cpu_phase(prec, l2, value, zero, iters=20):
    from oracle import oracle as O
    if not O.have_ref_spectral():
        return None
    dt = np.float32 if prec == "f32" else np.float64
    half = (1 << l2) >> 1
    rng = np.random.default_rng(0)
    re, im = (rng.random(half) + 0.5).astype(dt), (rng.random(half) - 0.5).astype(dt)
    O.ir_op("phase", re, im, 1 << l2, value, zero, prec, "ref")
    t0 = time.perf_counter()
    for _ in range(iters):
        O.ir_op("phase", re, im, 1 << l2, value, zero, prec, "ref")
    return iters / (time.perf_counter() - t0)      # includes the reference's setup creation per call, as its tester does not


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    ap.add_argument("--gib", type=float, default=1.0)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--cpu", action="store_true")
    args = ap.parse_args()
    rows = []
    for prec in ("f64", "f32"):
        for name, phase, zero in MODES:
            for l2 in (14,):
                r = {"op": "ir_phase", "mode": name, "precision": prec, "log2n": l2, "phase": phase, "zero_center": zero}
                r.update(run(S.IrOp.PHASE, prec, l2, phase, zero, args.gib, args.reps))
                if args.cpu:
                    c = cpu_phase(prec, l2, phase, zero)
                    if c:
                        r["cpu_reference_spectra_per_s_1core"] = round(c, 1)
                rows.append(r)
                print(json.dumps(r), flush=True)
    for prec in ("f32", "f64"):
        for op, name, v in ((S.IrOp.DELAY, "ir_delay", 2.5), (S.IrOp.TIME_REVERSE, "ir_time_reverse", 0.0), (S.IrOp.SPIKE, "ir_spike", 100.25)):
            r = {"op": name, "precision": prec, "log2n": 14}
            r.update(run(op, prec, 14, v, False, args.gib, args.reps))
            rows.append(r)
            print(json.dumps(r), flush=True)
    for l2 in (10, 12, 15, 16, 18, 20):
        for prec in ("f32", "f64"):
            r = {"op": "ir_phase", "mode": "Center Min", "precision": prec, "log2n": l2, "phase": 0.0, "zero_center": False}
            r.update(run(S.IrOp.PHASE, prec, l2, 0.0, False, args.gib, args.reps))
            rows.append(r)
            print(json.dumps(r), flush=True)
    if args.json:
        with open(args.json, "w") as fh:
            json.dump({"peak_GBps": 8000.0, "operand_GiB": args.gib, "rows": rows}, fh, indent=1)


if __name__ == "__main__":
    main()
