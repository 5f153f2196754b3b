#!/usr/bin/env python3
"""Throughput of the hisstools_* FFT surface on HBM-resident data (hcv_fft_exec_dev), against the HBM roofline and with
the reference's CPU transform timed beside it.

For every (operation, precision, log2n) a batch big enough to defeat the 256 MiB Infinity Cache (1 GiB of operands) is
transformed in one call; time = HIP events on the launch stream, best of `--reps`.  Algorithmic bytes per transform = one
read + one write of the operands (in place: 2 x M complex values), so achieved GB/s = batch * bytes / time.  Four-step
sizes move the data through HBM twice more (scratch), which the figure deliberately does not credit.

    python tests/perf/bench_fft.py [--json profiles/r01_fft_surface.json] [--gib 1.0] [--cpu]
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
import hisstools_library_amd.fft as F  # noqa: E402

PEAK = 8000.0


def run_case(op, prec, l2, gib, reps):
    real = 4 if prec == "f32" else 8
    tdt = torch.float32 if prec == "f32" else torch.float64
    complex_op = op in ("fft", "ifft")
    m = (1 << l2) if complex_op else (1 << l2) >> 1             # split values per array
    per = 2 * m * real                                          # bytes of one transform's operand
    batch = max(1, int(gib * (1 << 30)) // per)
    a = torch.rand(batch * m, device="cuda", dtype=tdt) * 2 - 1
    b = torch.rand(batch * m, device="cuda", dtype=tdt) * 2 - 1
    st = torch.cuda.current_stream().cuda_stream
    P = F.Precision.F32 if prec == "f32" else F.Precision.F64
    if op == "rfft_zip":
        x = torch.rand(batch * 2 * m, device="cuda", dtype=tdt) * 2 - 1

        def call():
            F.exec_dev(F.Op.RFFT_ZIP, P, l2, batch, x.data_ptr(), 0, a.data_ptr(), b.data_ptr(), 2 * m, m, 2 * m, st, False)
    elif op == "rifft_zip":
        x = torch.empty(batch * 2 * m, device="cuda", dtype=tdt)

        def call():
            F.exec_dev(F.Op.RIFFT_ZIP, P, l2, batch, a.data_ptr(), b.data_ptr(), x.data_ptr(), 0, m, 2 * m, 0, st, False)
    else:
        code = {"fft": F.Op.FFT, "ifft": F.Op.IFFT, "rfft": F.Op.RFFT, "rifft": F.Op.RIFFT}[op]

        def call():
            F.exec_dev(code, P, l2, batch, a.data_ptr(), b.data_ptr(), a.data_ptr(), b.data_ptr(), m, m, 0, st, False)
    call()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        if op not in ("rfft_zip", "rifft_zip"):                 # keep in-place data bounded: re-randomise outside the timed region
            a.uniform_(-1, 1)
            b.uniform_(-1, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    ok = bool(torch.isfinite(a).all())
    gbs = 2.0 * batch * per / (best * 1e-3) / 1e9
    pts = (1 << l2)
    flops = 5.0 * (pts if complex_op else pts / 2) * (l2 if complex_op else l2 - 1) * batch
    return {"op": op, "precision": prec, "log2n": l2, "batch": batch, "ms": round(best, 4), "achieved_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / PEAK, 4),
            "transforms_per_s": round(batch / (best * 1e-3), 1), "gflops_5nlogn": round(flops / (best * 1e-3) / 1e9, 1), "finite": ok,
            "path": "lds" if (l2 if complex_op else l2 - 1) <= (14 if prec == "f32" else 13) else "four-step"}


def vendor_case(prec, l2, gib, reps):
    """Yardstick only (nothing of the product uses it): the vendor FFT that PyTorch-ROCm binds (rocFFT through torch.fft.fft) on the same
    box and the same bytes — a batch of complex transforms, interleaved layout, out of place (one read + one write of the operand, as
    the in-place split transform counts)."""
    cdt = torch.complex64 if prec == "f32" else torch.complex128
    per = 2 * (1 << l2) * (4 if prec == "f32" else 8)
    batch = max(1, int(gib * (1 << 30)) // per)
    x = torch.view_as_complex(torch.rand((batch, 1 << l2, 2), device="cuda", dtype=torch.float32 if prec == "f32" else torch.float64) * 2 - 1)
    y = torch.fft.fft(x, dim=1)
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = torch.fft.fft(x, dim=1)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    gbs = 2.0 * batch * per / (best * 1e-3) / 1e9
    return {"op": "fft", "precision": prec, "log2n": l2, "batch": batch, "ms": round(best, 4), "achieved_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / PEAK, 4),
            "library": "rocFFT via torch.fft.fft (yardstick, interleaved, out of place)", "dtype": str(cdt)}


def cpu_case(op, prec, l2, budget=1.5):
    """The compiled reference (oracle/_ref) on one host core: transforms per second for the same operation."""
    from oracle import oracle as O
    if not O.have_ref():
        return None
    dt = np.float32 if prec == "f32" else np.float64
    n = 1 << l2
    m = n if op in ("fft", "ifft") else n >> 1
    rng = np.random.default_rng(0)
    a, b = rng.uniform(-1, 1, m).astype(dt), rng.uniform(-1, 1, m).astype(dt)
    f = O._surface_lib("ref")
    sfx = prec
    ptr = (lambda v: v.ctypes.data_as(O._f32p)) if dt == np.float32 else (lambda v: v.ctypes.data_as(O._f64p))
    if op in ("fft", "ifft"):
        fn = lambda: f["fft_" + sfx](ptr(a), ptr(b), l2, int(op == "ifft"))
    else:
        fn = lambda: f["rfft_inplace_" + sfx](ptr(a), ptr(b), l2, int(op == "rifft"))
    fn()
    count, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget:
        a[:] = 0.5
        b[:] = 0.25
        fn()
        count += 1
    return count / (time.perf_counter() - t0)       # includes the reference's per-call setup creation (tables), as its tester does not


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    ap.add_argument("--gib", type=float, default=1.0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--vendor", action="store_true", help="also time rocFFT (through torch.fft.fft) on the complex rows: a yardstick, never used by the product")
    ap.add_argument("--quick", action="store_true", help="a handful of complex sizes only (kernel tuning)")
    ap.add_argument("--only", default="", help="comma-separated op:precision:log2n rows (e.g. fft:f32:16,rfft:f32:20) instead of the plan")
    args = ap.parse_args()
    rows = []
    plan = [("fft", "f32", l) for l in (4, 6, 8, 10, 12, 14, 15, 16, 18, 20, 22)]
    plan += [("fft", "f64", l) for l in (4, 6, 8, 10, 12, 13, 14, 16, 18, 20, 22)]
    plan += [("rfft", "f32", l) for l in (8, 12, 15, 16, 20)] + [("rifft", "f32", l) for l in (8, 12, 15, 16, 20)]
    plan += [("rfft_zip", "f32", l) for l in (10, 14)] + [("rifft_zip", "f32", l) for l in (10, 14)] + [("rfft", "f64", l) for l in (12, 14, 18)]
    if args.quick:
        plan = [("fft", "f32", l) for l in (6, 8, 10, 12, 13, 14)] + [("fft", "f64", l) for l in (10, 12, 13)] + [("fft", "f32", 16), ("fft", "f32", 20)]
    if args.only:
        plan = [(o, p, int(l)) for o, p, l in (item.split(":") for item in args.only.split(","))]
    for op, prec, l2 in plan:
        r = run_case(op, prec, l2, args.gib, args.reps)
        if args.cpu and op in ("fft", "rfft", "rifft") and l2 <= 20:
            c = cpu_case(op, prec, l2)
            if c:
                r["cpu_reference_transforms_per_s_1core"] = round(c, 1)
        if args.vendor and op == "fft":
            try:
                v = vendor_case(prec, l2, args.gib, args.reps)
                r["vendor_rocfft_GBps"] = v["achieved_GBps"]
            except Exception as e:          # (a yardstick must not take the measurement down)
                r["vendor_rocfft_GBps"] = f"error: {e}"
        rows.append(r)
        print(json.dumps(r), flush=True)
    if args.json:
        with open(args.json, "w") as fh:
            json.dump({"peak_GBps": PEAK, "operand_GiB": args.gib, "rows": rows}, fh, indent=1)


if __name__ == "__main__":
    main()
