"""Live control calls against the UNMODIFIED reference: `set` / `reset` / `clear` of one (in, out) pair while the others keep
running, recorded by tests/golden/make_golden_restart.py (golden_restart_v1.npz) and replayed here sample by sample.

CPU: the oracle restatement reproduces the reference's streams (to rounding: the reference staggers its FFT phases at
random) — so "exact per-pair restart" in tests/test_pair_restart_gpu.py, which compares the HIP path with the oracle over
many more scenarios, means exactly what the reference does.  GPU (-m gpu): the HIP path against the same vectors."""
import os

import numpy as np
import pytest

from restart_scenarios import SCENARIOS, build, build_ntomono, drive, drive_ntomono

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 2e-6          # of the stream's peak: rounding-level agreement between differently staggered FFT phases
TOL_SUM = 1e-5


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_restart_v1.npz"))


def worst(y, ref):
    return float(np.abs(y.astype(np.float64) - ref.astype(np.float64)).max()) / float(np.abs(ref).max())


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_oracle_reproduces_the_reference_through_live_control_calls(oracle, gold, name):
    sc = SCENARIOS[name]
    conv, xs, script = build(oracle, sc)
    y = drive(conv, xs, sc["nout"], script, 512)
    assert y.shape == gold[name].shape
    assert worst(y, gold[name]) < TOL
    # the events matter: without them the stream is a different one
    plain, xs2, _ = build(oracle, sc)
    assert worst(drive(plain, xs2, sc["nout"], [], 512), gold[name]) > 1e-2


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_reference_agrees_where_present(oracle, gold, name):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref/libhisstools_ref.so is not present")
    sc = SCENARIOS[name]
    conv, xs, script = build(oracle, sc, backend="ref")
    assert worst(drive(conv, xs, sc["nout"], script, 333), gold[name]) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("block", [128, [700, 64, 8192, 3000]])
@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_gpu_reproduces_the_reference_through_live_control_calls(gold, name, block):
    import hisstools_library_amd as H
    assert H.load().hcv_device_count() > 0
    sc = SCENARIOS[name]
    conv, xs, script = build(H, sc)
    assert worst(drive(conv, xs, sc["nout"], script, block), gold[name]) < TOL_SUM


def test_oracle_ntomono_live_control_calls(oracle, gold):
    conv, xs, script = build_ntomono(oracle)
    assert worst(drive_ntomono(conv, xs, script, 512), gold["ntomono_3"]) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("block", [256, [1000, 37, 4096]])
def test_gpu_ntomono_live_control_calls(gold, block):
    import hisstools_library_amd as H
    conv, xs, script = build_ntomono(H)
    assert worst(drive_ntomono(conv, xs, script, block), gold["ntomono_3"]) < TOL_SUM
