"""Randomised differential parity (tests/perf/fuzz_parity.py): random matrices, IR lengths, latency modes, call-size patterns and
mid-stream control calls (IR swaps, clears and restarts of single pairs, full resets) against the CPU oracle.  The tool has
been run for 105000+ cases (65000 with the mid-stream control calls; worst relative error 6.5e-6 against the 1.5e-5 bound);
the suite runs a fixed slice of the same seeds so that a regression shows up with a seed to reproduce it.
Round 2 (whole-hop mode as one uniform convolution, direct input / output, staged IR loads): 20 661 more cases, also under
HCV_SERIAL=1 and HCV_MAX_BLOCK=8192, worst 6.5e-6; 6 292 cases with every Convolver sharded over two or three engines (HCV_DEVICES=0,0[,0]), worst 2.0e-6
(profiles/r02_fuzz.txt)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first", [1, 1001, 5001])
def test_random_cases_match_the_oracle(first):
    import fuzz_parity
    for seed in range(first, first + 80):
        kind, desc, err = fuzz_parity.one_case(seed)
        assert err <= fuzz_parity.TOL, (seed, kind, desc, err)
