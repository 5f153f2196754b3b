#!/usr/bin/env python3
"""Generate tests/golden/golden_fft_v1.npz from the UNMODIFIED reference (oracle/_ref/libhisstools_ref.so, compiled
from /root/reference by oracle/Makefile): every operation of the hisstools_* FFT surface (HISSTools_FFT.h:87-369), in
float and double, at small and medium sizes.  Run in the build container only:   python tests/golden/make_golden_fft.py

Keys: <op>_<precision>_<log2n>_{a,b}   inputs (b only for split operands; for the sample operands a holds in_length values)
      <op>_<precision>_<log2n>_{o0,o1} outputs (o1 only for split results)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

SIZES = (0, 1, 2, 3, 4, 5, 6, 8, 10)
G = {}
rng = np.random.default_rng(20260929)
for prec in O.FFT_PRECISIONS:
    dt = np.float32 if prec == "f32" else np.float64
    for l2 in SIZES:
        n = 1 << l2
        half = n >> 1
        for op in O.FFT_OPS:
            if prec == "f32_to_f64" and op not in ("rfft_zip", "unzip"):
                continue
            key = f"{op}_{prec}_{l2}"
            if op in ("fft", "ifft"):
                a, b = rng.uniform(-1, 1, n).astype(dt), rng.uniform(-1, 1, n).astype(dt)
                out = O.fft_surface(op, prec, l2, a, b, backend="ref")
            elif op in ("rfft", "rifft", "rifft_zip", "zip"):
                if not half:
                    continue
                a, b = rng.uniform(-1, 1, half).astype(dt), rng.uniform(-1, 1, half).astype(dt)
                out = O.fft_surface(op, prec, l2, a, b, backend="ref")
            else:                                               # rfft_zip / unzip: short, odd-length input where the size allows
                if not half:
                    continue
                in_len = n if l2 < 3 else n - 3
                a = rng.uniform(-1, 1, in_len).astype(np.float32 if prec != "f64" else np.float64)
                b = None
                out = O.fft_surface(op, prec, l2, a, in_length=in_len, backend="ref")
            G[key + "_a"] = a
            if b is not None:
                G[key + "_b"] = b
            if isinstance(out, tuple):
                G[key + "_o0"], G[key + "_o1"] = out
            else:
                G[key + "_o0"] = out

path = os.path.join(ROOT, "tests", "golden", "golden_fft_v1.npz")
np.savez_compressed(path, **G)
print(f"wrote {path}: {len(G)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")
