#!/usr/bin/env python3
"""Golden vectors for the rest of spectral_processor<T>::convolve / correlate: the real overloads in double and the complex
overloads in float and double, produced by the UNMODIFIED reference (oracle/_ref/libhisstools_ref_spectral.so, built by
`make -C oracle ref` from /root/reference).  Run in the build container; the .npz travels, the reference does not.

The complex overloads are recorded for Linear, Fold and FoldRepeat only: in Wrap and WrapCentre the reference's complex
instantiation reads past its result (SpectralProcessor.hpp:401-408 against :429-435), so there is nothing to pin there —
tests/test_spectral_overloads.py checks those modes against the real overloads by linearity instead."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

REAL_CASES = [(10, 4), (4, 10), (7, 7), (1, 1), (1, 9), (100, 33), (33, 100), (512, 512), (1000, 129), (2, 3), (3000, 2047)]
# (r1, i1, r2, i2) lengths: equal, ragged, missing planes, single samples
COMPLEX_CASES = [(10, 10, 4, 4), (4, 4, 10, 10), (7, 5, 7, 7), (1, 1, 1, 1), (1, 0, 9, 9), (100, 90, 33, 0), (0, 33, 100, 100), (512, 512, 512, 512),
                 (1000, 1000, 129, 100), (2, 2, 3, 1), (1500, 1400, 900, 1000)]
COMPLEX_MODES = (0, 3, 4)


def main():
    assert O.have_ref_spectral(), "build the reference first: make -C oracle ref"
    S = {}
    for ci, (n1, n2) in enumerate(REAL_CASES):
        a, b = O.synth_audio(400 + ci, n1).astype(np.float64), O.synth_audio(500 + ci, n2).astype(np.float64)
        a, b = a + 1e-9 * np.arange(n1), b - 1e-9 * np.arange(n2)              # not representable in float32
        S[f"rd{ci}_a"], S[f"rd{ci}_b"] = a, b
        for mode in range(5):
            S[f"rd{ci}_conv{mode}"] = O.spectral_convolve(a, b, mode, "ref")
            S[f"rd{ci}_corr{mode}"] = O.spectral_correlate(a, b, mode, "ref")
    for ci, sizes in enumerate(COMPLEX_CASES):
        for tag, dt in (("cf", np.float32), ("cd", np.float64)):
            ins = [O.synth_audio(600 + 4 * ci + k, max(n, 1))[:n].astype(dt) for k, n in enumerate(sizes)]
            for k, a in enumerate(ins):
                S[f"{tag}{ci}_in{k}"] = a
            for mode in COMPLEX_MODES:
                for name, corr in (("conv", False), ("corr", True)):
                    r, i = O.spectral_convolve_complex(*ins, mode, "ref", corr)
                    S[f"{tag}{ci}_{name}{mode}_r"], S[f"{tag}{ci}_{name}{mode}_i"] = r, i
    out = os.path.join(ROOT, "tests", "golden", "golden_spectral2_v1.npz")
    np.savez_compressed(out, **S)
    print("wrote", out, os.path.getsize(out), "bytes,", len(S), "arrays")


if __name__ == "__main__":
    main()
