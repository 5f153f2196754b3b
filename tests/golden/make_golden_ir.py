#!/usr/bin/env python3
"""Generate tests/golden/golden_ir_v1.npz from the UNMODIFIED reference (oracle/_ref/libhisstools_ref_spectral.so, compiled
from /root/reference by oracle/Makefile): ir_copy / ir_spike / ir_delay / ir_time_reverse / ir_phase
(SpectralFunctions.hpp:365-413) and spectral_processor::change_phase (SpectralProcessor.hpp:188-208), float and double.
Run in the build container only:   python tests/golden/make_golden_ir.py

Keys: spec_<prec>_<log2n>_{re,im}                  the input spectrum (real FFT of a decaying noise burst)
      <op>_<prec>_<log2n>_<case>_{re,im}           result; the cases are listed in CASES below
      cp_<prec>_<k>_{x,y}                          change_phase input / output, parameters in CP_CASES
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

SIZES = (3, 5, 8, 11)
# (op, value, zero_center); spike/delay positions are scaled by the fft size where marked "n*"
CASES = [("copy", 0.0, 0), ("time_reverse", 0.0, 0), ("spike", 3.25, 0), ("spike", "n*0.5+0.5", 0), ("delay", 1.5, 0), ("delay", -7.25, 0),
         ("delay", 0.0, 0), ("phase", 0.0, 0), ("phase", 0.0, 1), ("phase", 1.0, 0), ("phase", 1.0, 1), ("phase", 0.5, 0), ("phase", 0.5, 1),
         ("phase", 0.1, 1), ("phase", 0.9, 0), ("phase", 0.3, 0)]
CP_CASES = [(100, 0.0, 1.0), (1000, 1.0, 1.0), (37, 0.5, 2.0), (1, 0.3, 1.0), (512, 0.25, 1.0), (700, 0.75, 1.5)]


def value_of(v, n):
    return n * 0.5 + 0.5 if v == "n*0.5+0.5" else v


if __name__ == "__main__":
    G = {}
    rng = np.random.default_rng(20260930)
    for prec in ("f32", "f64"):
        dt = np.float32 if prec == "f32" else np.float64
        for l2 in SIZES:
            n = 1 << l2
            x = (rng.uniform(-1, 1, n) * np.exp(-np.arange(n) / (n / 8))).astype(dt)
            re, im = O.fft_surface("rfft_zip", prec, l2, x, backend="ref")
            G[f"spec_{prec}_{l2}_re"], G[f"spec_{prec}_{l2}_im"] = re, im
            for ci, (op, v, zc) in enumerate(CASES):
                ro, io = O.ir_op(op, re, im, n, value_of(v, n), bool(zc), prec, "ref")
                G[f"{op}_{prec}_{l2}_{ci}_re"], G[f"{op}_{prec}_{l2}_{ci}_im"] = ro, io
        for k, (size, ph, tm) in enumerate(CP_CASES):
            x = (rng.uniform(-1, 1, size) * np.exp(-np.arange(size) / (size / 6 + 1))).astype(dt)
            G[f"cp_{prec}_{k}_x"], G[f"cp_{prec}_{k}_y"] = x, O.change_phase(x, ph, tm, prec, "ref")
    path = os.path.join(ROOT, "tests", "golden", "golden_ir_v1.npz")
    np.savez_compressed(path, **G)
    print(f"wrote {path}: {len(G)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")
