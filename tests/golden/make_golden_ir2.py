#!/usr/bin/env python3
"""Generate tests/golden/golden_ir2_v1.npz from the UNMODIFIED reference (oracle/_ref/libhisstools_ref_spectral.so, compiled from
/root/reference by oracle/Makefile): the IR products ir_convolve_complex / ir_convolve_real / ir_correlate_complex / ir_correlate_real
(SpectralFunctions.hpp:415-436), float and double.  Run in the build container only:   python tests/golden/make_golden_ir2.py

Keys: in_<prec>_<n>_{a,b,c,d}                         the operands (in1 = a + i b, in2 = c + i d), n values per array
      <op>_<prec>_<n>_<k>_{re,im}                     result of case k of SCALES with `n` values per array (complex forms: fft_size = n,
                                                      real forms: fft_size = 2 n)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

COUNTS = (1, 2, 4, 16, 128, 512)
SCALES = (1.0, 0.5 / 4096, -3.7)

if __name__ == "__main__":
    G = {}
    rng = np.random.default_rng(20261001)
    for prec in ("f32", "f64"):
        dt = np.float32 if prec == "f32" else np.float64
        for n in COUNTS:
            ops = [rng.uniform(-1, 1, n).astype(dt) * np.exp(-np.arange(n) / (n / 4 + 1)).astype(dt) for _ in range(4)]
            ops[0][0] = dt(-0.0) if n > 1 else ops[0][0]           # a signed zero through bin 0
            for name, v in zip("abcd", ops):
                G[f"in_{prec}_{n}_{name}"] = v
            for op in O.IR_PRODUCTS:
                fs = n if op.endswith("complex") else 2 * n
                for k, sc in enumerate(SCALES):
                    re, im = O.ir_product(op, *ops, fs, sc, prec, "ref")
                    G[f"{op}_{prec}_{n}_{k}_re"], G[f"{op}_{prec}_{n}_{k}_im"] = re.copy(), im.copy()
    path = os.path.join(ROOT, "tests", "golden", "golden_ir2_v1.npz")
    np.savez_compressed(path, **G)
    print(f"wrote {path}: {len(G)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")
