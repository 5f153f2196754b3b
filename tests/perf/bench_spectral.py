#!/usr/bin/env python3
"""One-shot FFT convolution (spectral_processor::convolve) on HBM-resident operands, with the reference's
CPU time beside it.  Time = HIP events on the launch stream around one call, best of --reps.  A call is two forward real
FFTs, one bin-wise product, one inverse FFT and the edge arrangement; algorithmic bytes = the operands read + the result
written once (4 bytes per sample), so the GB/s figure is small by construction — the useful numbers are milliseconds per
call and output samples per second.

    python tests/perf/bench_spectral.py [--json profiles/r01_spectral_convolve.json] [--cpu]
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
from hisstools_library_amd import spectral_processor, EdgeMode  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu", action="store_true")
    args = ap.parse_args()
    sp = spectral_processor()
    st = torch.cuda.current_stream().cuda_stream
    rows = []
    cases = [(1000, 1000, EdgeMode.Linear), (16000, 16000, EdgeMode.Linear), (48000, 480, EdgeMode.Linear), (262144, 262144, EdgeMode.Linear),
             (480000, 48000, EdgeMode.Linear), (480000, 48000, EdgeMode.Wrap), (480000, 48000, EdgeMode.Fold), (500000, 500000, EdgeMode.Linear)]
    for n1, n2, mode in cases:
        a = torch.rand(n1, device="cuda") * 2 - 1
        b = torch.rand(n2, device="cuda") * 2 - 1
        n = sp.convolved_size(n1, n2, mode)
        out = torch.zeros(n, device="cuda")
        sp.convolve_dev(a.data_ptr(), n1, b.data_ptr(), n2, out.data_ptr(), mode, False, st, True)
        best = 1e30
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            sp.convolve_dev(a.data_ptr(), n1, b.data_ptr(), n2, out.data_ptr(), mode, False, st, False)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        span = (n1 + n2 - 1) if mode not in (EdgeMode.Fold, EdgeMode.FoldRepeat) else max(n1, n2) + 2 * (min(n1, n2) // 2) + min(n1, n2) - 1
        fft = 1 << max(5, int(np.ceil(np.log2(span))))
        r = {"size1": n1, "size2": n2, "mode": mode.name, "fft_size": fft, "out_samples": n, "ms": round(best, 4),
             "out_msamples_per_s": round(n / (best * 1e-3) / 1e6, 1), "algorithmic_GBps": round(4.0 * (n1 + n2 + n) / (best * 1e-3) / 1e9, 2),
             "fft_traffic_GBps": round(8.0 * fft * 3 / (best * 1e-3) / 1e9, 1), "finite": bool(torch.isfinite(out).all())}
        if args.cpu:
            from oracle import oracle as O
            if O.have_ref_spectral():
                ha, hb = a.cpu().numpy(), b.cpu().numpy()
                O.spectral_convolve(ha, hb, int(mode), "ref")
                t0 = time.perf_counter()
                O.spectral_convolve(ha, hb, int(mode), "ref")
                r["cpu_reference_ms_1core"] = round(1e3 * (time.perf_counter() - t0), 2)
        rows.append(r)
        print(json.dumps(r), flush=True)
    # the remaining overloads (double real, complex float / double) through the host-pointer entry points: wall time of one call,
    # PCIe both ways included, best of --reps, with the reference on one host core beside it
    over = []
    rng = np.random.default_rng(3)
    for n1, n2 in ((16000, 16000), (262144, 262144), (1000000, 1000000)):
        for kind in ("real f64", "complex f32", "complex f64"):
            dt = np.float32 if kind.endswith("f32") else np.float64
            ins = [rng.uniform(-1, 1, n).astype(dt) for n in ((n1, n2) if kind.startswith("real") else (n1, n1, n2, n2))]
            call = (lambda: sp.convolve(*ins, EdgeMode.Linear)) if kind.startswith("real") else (lambda: sp.convolve_complex(*ins, EdgeMode.Linear))
            call()
            best = 1e30
            for _ in range(args.reps):
                t0 = time.perf_counter()
                y = call()
                best = min(best, time.perf_counter() - t0)
            r = {"overload": kind, "size1": n1, "size2": n2, "ms_host_call": round(1e3 * best, 3),
                 "out_msamples_per_s": round((n1 + n2 - 1) / best / 1e6, 1)}
            if args.cpu:
                from oracle import oracle as O
                if O.have_ref_spectral():
                    ref = (lambda: O.spectral_convolve(*ins, 0, "ref")) if kind.startswith("real") else (lambda: O.spectral_convolve_complex(*ins, 0, "ref"))
                    ref()
                    t0 = time.perf_counter()
                    ref()
                    r["cpu_reference_ms_1core"] = round(1e3 * (time.perf_counter() - t0), 2)
            over.append(r)
            print(json.dumps(r), flush=True)
    if args.json:
        with open(args.json, "w") as fh:
            json.dump({"rows": rows, "overloads_host_pointers": over, "note": "fft_traffic_GBps = three transforms of fft_size, each read + written once (8 bytes per sample) / time"}, fh, indent=1)


if __name__ == "__main__":
    main()
