"""numIns / numOuts changing between process calls, against a vector recorded from the UNMODIFIED reference
(tests/golden/make_golden_active.py).  The reference only processes the channels it is handed (Convolver.cpp:148-153,
NToMonoConvolve.cpp:41), so an inactive pair's private buffers FREEZE and the pair later resumes as if no time had passed.

CPU: the oracle restatement reproduces that.  GPU: the HIP engine agrees with the reference wherever channel counts are
constant, and differs in exactly the documented way (deviation 3b) where a pair comes back:
it restarts from silence, so what is missing from its output is the ringing of the input it saw BEFORE it dropped out —
computed here in float64 and added back, which pins the deviation to the sample."""
import os

import numpy as np
import pytest

from active_scenario import SC, build, drive

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 2e-6
TOL_SUM = 1e-5


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_active_v1.npz"))["y"]


def worst(y, ref, peak=None):
    peak = float(np.abs(ref).max()) if peak is None else peak
    return float(np.abs(y.astype(np.float64) - ref.astype(np.float64)).max()) / peak


def test_oracle_reproduces_the_reference_freeze(oracle, gold):
    conv, xs, _ = build(oracle)
    y = drive(conv, xs, 512)
    assert worst(y, gold) < TOL
    # the channel counts matter: with all channels active throughout the stream is a different one
    full, xs2, _ = build(oracle)
    assert worst(full.run(xs2, SC["nout"], 512), gold) > 1e-2


def test_reference_agrees_where_present(oracle, gold):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref/libhisstools_ref.so is not present")
    conv, xs, _ = build(oracle, backend="ref")
    assert worst(drive(conv, xs, 333), gold) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("block", [500, 4000])
def test_gpu_differs_from_the_reference_only_by_the_documented_restart(gold, block):
    from scipy.signal import fftconvolve
    import hisstools_library_amd as H
    assert H.load().hcv_device_count() > 0
    conv, xs, irs = build(H)
    y = drive(conv, xs, block)
    (_, _, _), (t_out, ni_mid, no_mid), (t_back, _, _) = SC["cuts"]
    peak = float(np.abs(gold).max())
    # constant counts, and the span with fewer channels (its untouched outputs stay zero on both sides): identical streams
    assert worst(y[:, :t_back], gold[:, :t_back], peak) < TOL_SUM
    # after the channels come back: the reference resumes each returning pair's frozen state, i.e. its output contains the
    # response to the input the pair saw before it dropped out, spliced in at the point it left off; here the pair restarts
    # from silence.  reference - engine = that ringing, per returning pair (in 2 or out 2 was inactive).
    n_tail = SC["S"] - t_back
    ring = np.zeros((SC["nout"], n_tail))
    for (i, o), h in irs.items():
        if i >= ni_mid or o >= no_mid:
            pre = np.concatenate([xs[i, :t_out].astype(np.float64), np.zeros(n_tail)])
            ring[o] += fftconvolve(pre, h.astype(np.float64))[t_out:t_out + n_tail]
    assert worst(y[:, t_back:] + ring, gold[:, t_back:], peak) < TOL_SUM
    # and the deviation is real: without the correction the returning outputs are far from the reference
    assert worst(y[:, t_back:], gold[:, t_back:], peak) > 1e-2
