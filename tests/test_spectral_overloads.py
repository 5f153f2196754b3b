"""The rest of spectral_processor<T>::convolve / correlate: the real overloads in double and the complex
overloads in float and double (SpectralProcessor.hpp:164-184).

CPU: the oracle restatement against the golden vectors of the unmodified reference (tests/golden/golden_spectral2_v1.npz,
generator make_golden_spectral2.py) — bit-identical in float, to 1e-13 in double — and, for the complex overloads in every
edge mode, against the REAL overloads by linearity ((r1 + j i1) * (r2 + j i2) = (r1*r2 - i1*i2) + j (r1*i2 + i1*r2)); that
is also what pins the two wrap modes, where the reference's complex instantiation reads past its result.
GPU (-m gpu): the HIP path through the C ABI against the oracle and the golden vectors."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden_spectral2 import COMPLEX_CASES, COMPLEX_MODES, REAL_CASES  # noqa: E402

TOL32, TOL64 = 2e-6, 1e-12          # of the output peak


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_spectral2_v1.npz"))


def close(y, ref, tol):
    assert y.shape == ref.shape and y.dtype == ref.dtype
    peak = max(float(np.abs(ref).max()) if ref.size else 0.0, 1.0)
    return (float(np.abs(y - ref).max()) if ref.size else 0.0) <= tol * peak


def pad(a, n):
    return np.concatenate([a, np.zeros(n - a.size, a.dtype)])


def by_linearity(real_op, ins, mode, corr):
    """the complex overload out of four calls of the real one (operands zero-padded to the complex operand's size first)"""
    r1, i1, r2, i2 = ins
    n1, n2 = max(r1.size, i1.size), max(r2.size, i2.size)
    r1, i1, r2, i2 = pad(r1, n1), pad(i1, n1), pad(r2, n2), pad(i2, n2)
    f = lambda a, b: real_op(a, b, mode, corr).astype(np.float64)             # noqa: E731
    if not corr or (n1 == 1 and n2 == 1):      # two single samples: the reference's shortcut is the plain product for both (:590-595)
        return f(r1, r2) - f(i1, i2), f(r1, i2) + f(i1, r2)
    return f(r1, r2) + f(i1, i2), f(i1, r2) - f(r1, i2)                        # in1 x conj(in2)


# ------------------------------------------------------------------------------------------- CPU: the oracle

@pytest.mark.parametrize("ci", range(len(REAL_CASES)))
def test_oracle_real_double_matches_reference_vectors(oracle, gold, ci):
    a, b = gold[f"rd{ci}_a"], gold[f"rd{ci}_b"]
    assert a.dtype == np.float64 and (a.size, b.size) == REAL_CASES[ci]
    for mode in range(5):
        assert close(oracle.spectral_convolve(a, b, mode), gold[f"rd{ci}_conv{mode}"], 1e-13)
        assert close(oracle.spectral_correlate(a, b, mode), gold[f"rd{ci}_corr{mode}"], 1e-13)


@pytest.mark.parametrize("ci", range(len(COMPLEX_CASES)))
def test_oracle_complex_matches_reference_vectors(oracle, gold, ci):
    for tag, dt in (("cf", np.float32), ("cd", np.float64)):
        ins = [gold[f"{tag}{ci}_in{k}"] for k in range(4)]
        assert tuple(a.size for a in ins) == COMPLEX_CASES[ci] and all(a.dtype == dt for a in ins)
        for mode in COMPLEX_MODES:
            for name, corr in (("conv", False), ("corr", True)):
                r, i = oracle.spectral_convolve_complex(*ins, mode, correlate=corr)
                if dt == np.float32:
                    # bit-identical up to the sign of zero / denormal dust in a plane whose exact value is zero
                    assert close(r, gold[f"{tag}{ci}_{name}{mode}_r"], 1e-7) and close(i, gold[f"{tag}{ci}_{name}{mode}_i"], 1e-7)
                else:
                    assert close(r, gold[f"{tag}{ci}_{name}{mode}_r"], 1e-13) and close(i, gold[f"{tag}{ci}_{name}{mode}_i"], 1e-13)


@pytest.mark.parametrize("dt,tol", [(np.float32, 4e-6), (np.float64, 1e-12)])
def test_oracle_complex_is_the_real_overloads_by_linearity(oracle, dt, tol):
    rng = np.random.default_rng(5)
    for sizes in COMPLEX_CASES[:7] + [(40, 37, 40, 40), (64, 64, 17, 17), (17, 17, 64, 60)]:
        ins = [rng.uniform(-1, 1, n).astype(dt) for n in sizes]
        for mode in range(5):
            for corr in (False, True):
                r, i = oracle.spectral_convolve_complex(*ins, mode, correlate=corr)
                tr, ti = by_linearity(lambda a, b, m, c: oracle.spectral_convolve(a, b, m, correlate=c), ins, mode, corr)
                peak = max(np.abs(tr).max(), np.abs(ti).max(), 1.0)
                assert np.abs(r - tr).max() <= tol * peak and np.abs(i - ti).max() <= tol * peak, (sizes, mode, corr)


def test_reference_agrees_where_present(oracle):
    if not oracle.have_ref_spectral():
        pytest.skip("oracle/_ref/libhisstools_ref_spectral.so is not present")
    rng = np.random.default_rng(12)
    for n1, n2 in ((5, 5), (64, 17), (17, 64), (700, 300)):
        a, b = rng.uniform(-1, 1, n1), rng.uniform(-1, 1, n2)
        for mode in range(5):
            assert close(oracle.spectral_convolve(a, b, mode), oracle.spectral_convolve(a, b, mode, "ref"), 1e-13)
            assert close(oracle.spectral_correlate(a, b, mode), oracle.spectral_correlate(a, b, mode, "ref"), 1e-13)
        for dt, tol in ((np.float32, 1e-7), (np.float64, 1e-13)):
            ins = [rng.uniform(-1, 1, n).astype(dt) for n in (n1, n1 - 1, n2, n2)]
            for mode in COMPLEX_MODES:
                for corr in (False, True):
                    p, r = oracle.spectral_convolve_complex(*ins, mode, correlate=corr), oracle.spectral_convolve_complex(*ins, mode, "ref", corr)
                    assert close(p[0], r[0], tol) and close(p[1], r[1], tol)


# ------------------------------------------------------------------------------------------- GPU

@pytest.fixture(scope="module")
def sp():
    import hisstools_library_amd as H
    assert H.load().hcv_device_count() > 0
    return H.spectral_processor()


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(REAL_CASES)))
def test_gpu_real_double(sp, oracle, gold, ci):
    a, b = gold[f"rd{ci}_a"], gold[f"rd{ci}_b"]
    for mode in range(5):
        for name, corr in (("conv", False), ("corr", True)):
            y = sp.correlate(a, b, mode) if corr else sp.convolve(a, b, mode)
            assert close(y, gold[f"rd{ci}_{name}{mode}"], TOL64), (REAL_CASES[ci], mode, corr)
            assert close(y, oracle.spectral_convolve(a, b, mode, correlate=corr), TOL64)


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(COMPLEX_CASES)))
def test_gpu_complex(sp, oracle, gold, ci):
    for tag, tol in (("cf", TOL32), ("cd", TOL64)):
        ins = [gold[f"{tag}{ci}_in{k}"] for k in range(4)]
        for mode in range(5):
            for name, corr in (("conv", False), ("corr", True)):
                r, i = sp.correlate_complex(*ins, mode) if corr else sp.convolve_complex(*ins, mode)
                pr, pi = oracle.spectral_convolve_complex(*ins, mode, correlate=corr)
                assert close(r, pr, tol) and close(i, pi, tol), (COMPLEX_CASES[ci], tag, mode, corr)
                if mode in COMPLEX_MODES:
                    assert close(r, gold[f"{tag}{ci}_{name}{mode}_r"], tol) and close(i, gold[f"{tag}{ci}_{name}{mode}_i"], tol)


@pytest.mark.gpu
@pytest.mark.parametrize("n1,n2", [(48000, 20000), (20000, 48000), (300000, 100000)])
def test_gpu_large_sizes(sp, oracle, n1, n2):
    a, b = oracle.synth_audio(1, n1).astype(np.float64), oracle.synth_ir(1, 1, n2).astype(np.float64)
    c, d = oracle.synth_audio(2, n1 - 7), oracle.synth_ir(2, 2, n2)
    for mode in (0, 2, 4):
        for corr in (False, True):
            y = sp.correlate(a, b, mode) if corr else sp.convolve(a, b, mode)
            assert close(y, oracle.spectral_convolve(a, b, mode, correlate=corr), 1e-11)
            ins = [a.astype(np.float32), c, b.astype(np.float32), d]
            r, i = sp.correlate_complex(*ins, mode) if corr else sp.convolve_complex(*ins, mode)
            pr, pi = oracle.spectral_convolve_complex(*ins, mode, correlate=corr)
            peak = max(np.abs(pr).max(), np.abs(pi).max())
            assert np.abs(r - pr).max() <= 1e-5 * peak and np.abs(i - pi).max() <= 1e-5 * peak


@pytest.mark.gpu
def test_gpu_limits_and_empty_operands(sp):
    z32, z64 = np.zeros(0, np.float32), np.zeros(0)
    assert sp.convolve(z64, np.ones(3), 0).size == 0
    r, i = sp.convolve_complex(z32, z32, np.ones(3, np.float32), z32, 0)
    assert r.size == 0 and i.size == 0
    # a purely real and a purely imaginary operand: (a)(j b) = j (a b)
    a, b = np.arange(1, 6, dtype=np.float64), np.array([1.0, -2.0, 0.5])
    r, i = sp.convolve_complex(a, z64, z64, b, 0)
    assert np.abs(r).max() < 1e-12 and np.allclose(i, np.convolve(a, b), atol=1e-12)
    assert sp.convolve(np.ones(1 << 22), np.ones(2), 0).size == 0            # would need a 2^23-point FFT


@pytest.mark.gpu
def test_gpu_sizes_beyond_the_engine_fft(sp, oracle):
    """circular sizes of 2^21 and 2^22: the float real overloads leave the convolution engine's kernels (<= 2^20) for the
    general FFT surface; checked against the oracle and, sparsely, against the defining sum in float64"""
    n1, n2 = 1_500_000, 700_000                                  # linear size 2.2 M -> 2^22
    a, b = oracle.synth_audio(3, n1), oracle.synth_ir(3, 3, n2)
    for corr in (False, True):
        y = sp.correlate(a, b, 0) if corr else sp.convolve(a, b, 0)
        ref = oracle.spectral_convolve(a, b, 0, correlate=corr)
        assert y.dtype == np.float32 and y.shape == ref.shape
        assert np.abs(y - ref).max() <= 2e-5 * np.abs(ref).max()
    y = sp.convolve(a, b, 0)
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    for k in (0, 5, 699_999, 1_000_000, 2_199_998):
        lo, hi = max(0, k - n2 + 1), min(k, n1 - 1)
        t = float(np.dot(a64[lo:hi + 1], b64[k - hi:k - lo + 1][::-1]))
        assert abs(y[k] - t) <= 2e-5 * np.abs(y).max()
    yd = sp.convolve(a64[:1_100_000], b64, 2)                    # double, WrapCentre, 2^21
    assert np.abs(yd - oracle.spectral_convolve(a64[:1_100_000], b64, 2)).max() <= 1e-10 * np.abs(yd).max()
