#!/usr/bin/env python3
"""Golden vector for a host that varies numIns / numOuts between process calls, produced by the UNMODIFIED reference
(oracle/_ref/libhisstools_ref.so): it pins what the reference does there — an inactive pair's private state is FROZEN and
resumes later as if no time had passed — which is the one place the HIP engine deliberately differs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
from active_scenario import build, drive  # noqa: E402


def main():
    assert O.have_ref(), "build the reference first: make -C oracle ref"
    conv, xs, _ = build(O, backend="ref")
    y = drive(conv, xs, 1000)
    out = os.path.join(ROOT, "tests", "golden", "golden_active_v1.npz")
    np.savez_compressed(out, y=np.asarray(y, np.float32))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
