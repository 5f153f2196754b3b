#!/usr/bin/env python3
"""Golden vectors for live control calls — `set` / `reset` / `clear` of ONE (in, out) pair while the others keep running —
produced by the UNMODIFIED reference (oracle/_ref/libhisstools_ref.so, `make -C oracle ref` where /root/reference exists).
They pin what "exact per-pair restart" means: the pair forgets its input and drops its pending output at the sample of the
next process call (MonoConvolve.cpp:139-150 -> PartitionedConvolve.cpp:262-292, TimeDomainConvolve.cpp:91-98).  The
reference staggers its FFT phases at random, so its output is reproducible to rounding only; the vectors are compared at the
suite's tolerance.  Scenarios are described in tests/test_restart_golden.py: SCENARIOS (shared with this generator)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
from restart_scenarios import SCENARIOS, build, build_ntomono, drive, drive_ntomono  # noqa: E402


def main():
    assert O.have_ref(), "build the reference first: make -C oracle ref"
    S = {}
    for name, sc in SCENARIOS.items():
        conv, xs, script = build(O, sc, backend="ref")
        S[name] = drive(conv, xs, sc["nout"], script, 1024)
    conv, xs, script = build_ntomono(O, backend="ref")
    S["ntomono_3"] = drive_ntomono(conv, xs, script, 1024)
    out = os.path.join(ROOT, "tests", "golden", "golden_restart_v1.npz")
    np.savez_compressed(out, **{k: np.asarray(v, np.float32) for k, v in S.items()})
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
