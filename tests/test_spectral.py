"""spectral_processor<float>::convolve / correlate, real overloads (first "next" row of SURVEY.md §8f).

CPU: the oracle restatement is bit-identical to the golden vectors produced by the unmodified reference
(tests/golden/golden_spectral_v1.npz) and agrees with numpy.  GPU (-m gpu): the HIP path through the C ABI matches the
oracle, the golden vectors and float64 numpy for every edge mode."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(10, 4), (4, 10), (7, 7), (1, 1), (1, 9), (100, 33), (33, 100), (512, 512), (1000, 129), (2, 3), (3000, 2047)]
TOL = 2e-6          # of the output peak, as for the streaming path


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_spectral_v1.npz"))


def truth(a, b, mode, correlate):
    """float64 statement of the five edge modes built from the linear result only."""
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    n1, n2 = a.size, b.size
    mn, mx = min(n1, n2), max(n1, n2)
    if not correlate:
        lin = np.convolve(a64, b64)
        if mode == 0:
            return lin
        if mode in (1, 2):                                      # circular over max(n1, n2)
            out = np.zeros(mx)
            np.add.at(out, np.arange(lin.size) % mx, lin)
            return np.roll(out, -((mn - 1) >> 1)) if mode == 2 else out
        return None                                             # fold modes: checked against the oracle only
    lin = np.correlate(a64, b64, "full")                        # lags -(n2-1) .. n1-1
    if mode == 0:
        return np.concatenate([lin[n2 - 1:], lin[: n2 - 1]])
    return None


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_oracle_matches_reference_vectors(oracle, gold, ci):
    a, b = gold[f"sp{ci}_a"], gold[f"sp{ci}_b"]
    assert (a.size, b.size) == CASES[ci]
    for mode in range(5):
        assert np.array_equal(oracle.spectral_convolve(a, b, mode), gold[f"sp{ci}_conv{mode}"])
        assert np.array_equal(oracle.spectral_correlate(a, b, mode), gold[f"sp{ci}_corr{mode}"])
        assert oracle.spectral_size(a.size, b.size, mode) == gold[f"sp{ci}_conv{mode}"].size


def test_vectors_match_numpy(gold):
    for ci in range(len(CASES)):
        a, b = gold[f"sp{ci}_a"], gold[f"sp{ci}_b"]
        for mode in range(5):
            for corr in (False, True):
                t = truth(a, b, mode, corr)
                if t is None:
                    continue
                y = gold[f"sp{ci}_{'corr' if corr else 'conv'}{mode}"]
                assert np.abs(y - t).max() <= 5e-6 * max(1.0, np.abs(t).max())


def test_reference_agrees_where_present(oracle, gold):
    if not oracle.have_ref_spectral():
        pytest.skip("oracle/_ref/libhisstools_ref_spectral.so is not present")
    rng = np.random.RandomState(11)
    for n1, n2 in ((5, 5), (64, 17), (17, 64), (1500, 300)):
        a, b = rng.uniform(-1, 1, n1).astype(np.float32), rng.uniform(-1, 1, n2).astype(np.float32)
        for mode in range(5):
            assert np.array_equal(oracle.spectral_convolve(a, b, mode), oracle.spectral_convolve(a, b, mode, "ref"))
            assert np.array_equal(oracle.spectral_correlate(a, b, mode), oracle.spectral_correlate(a, b, mode, "ref"))


def test_empty_inputs(oracle):
    assert oracle.spectral_size(0, 5, 0) == 0 and oracle.spectral_size(5, 0, 1) == 0


# ------------------------------------------------------------------------------------------- GPU

@pytest.fixture(scope="module")
def sp():
    import hisstools_library_amd as H
    assert H.load().hcv_device_count() > 0
    return H.spectral_processor()


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(CASES)))
def test_gpu_matches_golden_and_oracle(sp, oracle, gold, ci):
    a, b = gold[f"sp{ci}_a"], gold[f"sp{ci}_b"]
    for mode in range(5):
        for corr in (False, True):
            y = sp.correlate(a, b, mode) if corr else sp.convolve(a, b, mode)
            ref = gold[f"sp{ci}_{'corr' if corr else 'conv'}{mode}"]
            assert y.shape == ref.shape
            peak = max(np.abs(ref).max(), 1e-30)
            assert np.abs(y - ref).max() <= TOL * max(peak, 1.0), (CASES[ci], mode, corr)
            t = truth(a, b, mode, corr)
            if t is not None:
                assert np.abs(y - t).max() <= TOL * max(np.abs(t).max(), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("n1,n2", [(48000, 20000), (20000, 48000), (300000, 100000), (65536, 65536)])
def test_gpu_large_sizes(sp, oracle, n1, n2):
    a, b = oracle.synth_audio(1, n1), oracle.synth_ir(1, 1, n2)
    for mode in (0, 2, 4):
        for corr in (False, True):
            y = sp.correlate(a, b, mode) if corr else sp.convolve(a, b, mode)
            ref = oracle.spectral_correlate(a, b, mode) if corr else oracle.spectral_convolve(a, b, mode)
            assert np.abs(y - ref).max() <= 1e-5 * np.abs(ref).max()


@pytest.mark.gpu
def test_gpu_sizes_and_limits(sp):
    assert sp.convolved_size(10, 4, 0) == 13 and sp.convolved_size(10, 4, 1) == 10 and sp.correlated_size(4, 10, 3) == 10
    assert sp.convolved_size(0, 4, 0) == 0
    assert sp.convolved_size(1 << 22, 2, 0) == 0                  # would need a 2^23-point FFT: beyond the engine's maximum
    assert sp.convolved_size(1 << 20, 2, 0) == (1 << 20) + 1
    assert sp.convolve(np.zeros(0, np.float32), np.ones(3, np.float32), 0).size == 0


@pytest.mark.gpu
def test_gpu_device_resident_operands(oracle):
    """hcv_spectral_*_f32_dev: operands and result in HBM, on a side stream, every edge mode, sizes on both sides of the
    LDS limit; repeated calls re-use the cached scratch."""
    torch = pytest.importorskip("torch")
    from hisstools_library_amd import spectral_processor
    sp = spectral_processor()
    rng = np.random.default_rng(4)
    st = torch.cuda.Stream()
    for n1, n2 in ((1000, 129), (129, 1000), (40000, 30000), (7, 7)):
        a, b = rng.uniform(-1, 1, n1).astype(np.float32), rng.uniform(-1, 1, n2).astype(np.float32)
        da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        for mode in range(5):
            for corr in (False, True):
                want = oracle.spectral_correlate(a, b, mode) if corr else oracle.spectral_convolve(a, b, mode)
                out = torch.full((want.size,), 7.0, device="cuda")
                torch.cuda.synchronize()
                sp.convolve_dev(da.data_ptr(), n1, db.data_ptr(), n2, out.data_ptr(), mode, corr, st.cuda_stream, True)
                got = out.cpu().numpy()
                peak = float(np.abs(want).max()) or 1.0
                assert float(np.abs(got - want).max()) / peak <= TOL * (1 if max(n1, n2) < 10000 else 4), (n1, n2, mode, corr)
