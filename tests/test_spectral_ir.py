"""Spectral IR functions (fourth "next" row of SURVEY.md §8f; SpectralFunctions.hpp:365-413) and
spectral_processor::change_phase (SpectralProcessor.hpp:188-208).

CPU: the oracle restatement against golden vectors produced by the unmodified reference (float: bit-identical; double:
1e-11 of the peak — the reference's double FFT passes order the arithmetic differently and the linear-phase factors
amplify that) and against first-principles numpy.  GPU (-m gpu): the HIP path through hcv_ir_exec against the oracle,
the golden vectors and the same first-principles properties, for every mode of ir_phase the reference's
IR_Manipulation_Tester times ("Zero/Center" x "Mix/Min/Max/Lin")."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden_ir import CASES, CP_CASES, SIZES, value_of  # noqa: E402  (case tables only; the generator itself needs the reference)

TOL32 = 5e-6            # of the output peak: the minimum-phase path is two transforms plus log / exp
TOL64 = 1e-11


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_ir_v1.npz"))


def err(got, want):
    got = got if isinstance(got, tuple) else (got,)
    want = want if isinstance(want, tuple) else (want,)
    peak = max(float(np.abs(np.asarray(w, np.float64)).max()) for w in want) or 1.0
    return max(float(np.abs(np.asarray(g, np.float64) - np.asarray(w, np.float64)).max()) for g, w in zip(got, want)) / peak


def unpack(re, im):
    """packed half spectrum -> complex bins 0 .. N/2 (the reference doubles its spectra; keep that scale)"""
    z = np.zeros(re.size + 1, complex)
    z[: re.size] = np.asarray(re, np.float64) + 1j * np.asarray(im, np.float64)
    z[0] = re[0]
    z[re.size] = im[0]
    return z


# ------------------------------------------------------------------------------------------------ CPU: oracle pinned

def test_oracle_matches_reference_vectors(oracle, gold):
    for prec in ("f32", "f64"):
        for l2 in SIZES:
            n = 1 << l2
            re, im = gold[f"spec_{prec}_{l2}_re"], gold[f"spec_{prec}_{l2}_im"]
            for ci, (op, v, zc) in enumerate(CASES):
                got = oracle.ir_op(op, re, im, n, value_of(v, n), bool(zc), prec)
                want = (gold[f"{op}_{prec}_{l2}_{ci}_re"], gold[f"{op}_{prec}_{l2}_{ci}_im"])
                if prec == "f32":
                    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (op, v, zc, l2)
                else:
                    assert err(got, want) <= TOL64, (op, v, zc, l2, err(got, want))
        for k, (size, ph, tm) in enumerate(CP_CASES):
            got = oracle.change_phase(gold[f"cp_{prec}_{k}_x"], ph, tm, prec)
            want = gold[f"cp_{prec}_{k}_y"]
            assert got.size == want.size
            assert np.array_equal(got, want) if prec == "f32" else err(got, want) <= TOL64


def check_first_principles(run, prec, t):
    """Properties that define the functions, independent of any implementation."""
    dt = np.float32 if prec == "f32" else np.float64
    rng = np.random.default_rng(11)
    n, l2 = 1024, 10
    x = (rng.uniform(-1, 1, n) * np.exp(-np.arange(n) / 60.0)).astype(dt)
    X = np.fft.rfft(x.astype(np.float64)) * 2.0
    re, im = X.real[: n // 2].astype(dt), X.imag[: n // 2].astype(dt)
    im[0] = X.real[n // 2]
    # spike: spectrum of a unit impulse (not doubled: the functor writes exp(i k i) as is)
    z = unpack(*run("spike", None, None, n, 5.0, False))
    d = np.zeros(n)
    d[5] = 1.0
    assert np.abs(z - np.fft.rfft(d)).max() <= t
    # delay by whole samples = circular shift; the packed Nyquist slot keeps only the real part
    z = unpack(*run("delay", re, im, n, 3.0, False))
    assert np.abs(z - np.fft.rfft(np.roll(x.astype(np.float64), 3)) * 2.0).max() <= t * np.abs(X).max()
    # time reverse = conjugate = spectrum of x[-n mod N]
    z = unpack(*run("time_reverse", re, im, n, 0.0, False))
    assert np.abs(z - np.fft.rfft(np.roll(x[::-1].astype(np.float64), 1)) * 2.0).max() <= t * np.abs(X).max()
    # minimum phase: same magnitude, and its impulse response is causal with the energy packed to the front
    zr, zi = run("phase", re, im, n, 0.0, False)
    zmin = unpack(zr, zi)
    mag = np.abs(unpack(re, im))
    assert np.abs(np.abs(zmin)[1:-1] - mag[1:-1]).max() <= 20 * t * mag.max()
    h = np.fft.irfft(zmin, n)
    e = np.cumsum(h ** 2)
    e0 = np.cumsum(np.fft.irfft(unpack(re, im), n) ** 2)
    assert (e[: n // 2] >= e0[: n // 2] * (1 - 1e-3) - 1e-6).all()
    # maximum phase (zero centred) is the conjugate of minimum phase; linear phase keeps only the magnitude
    mr, mi = run("phase", re, im, n, 1.0, True)
    assert err((mr[1:], mi[1:]), (zr[1:], -zi[1:])) <= 4 * t
    lr, li = run("phase", re, im, n, 0.5, True)
    assert np.abs(lr[1:] - mag[1:-1]).max() <= 4 * t * mag.max() and not li[1:].any()


def test_oracle_first_principles(oracle):
    for prec, t in (("f32", 2e-6), ("f64", 1e-12)):
        check_first_principles(lambda op, re, im, n, v, zc: oracle.ir_op(op, re, im, n, v, zc, prec), prec, t)


# ------------------------------------------------------------------------------------------------ GPU

def hip_ir(op, re, im, n, v, zc, prec):
    import hisstools_library_amd.spectral_functions as S
    dt = np.float32 if prec == "f32" else np.float64
    if op == "spike":
        return S.ir_spike(n, v, dtype=dt)
    re, im = np.asarray(re, dt), np.asarray(im, dt)
    if op == "copy":
        return S.ir_copy(re, im, n)
    if op == "delay":
        return S.ir_delay(re, im, n, v)
    if op == "time_reverse":
        return S.ir_time_reverse(re, im, n)
    return S.ir_phase(re, im, n, v, zc)


@pytest.mark.gpu
def test_gpu_matches_reference_vectors(gold):
    from hisstools_library_amd import spectral_processor
    sp = spectral_processor()
    for prec in ("f32", "f64"):
        t = TOL32 if prec == "f32" else TOL64
        for l2 in SIZES:
            n = 1 << l2
            re, im = gold[f"spec_{prec}_{l2}_re"], gold[f"spec_{prec}_{l2}_im"]
            for ci, (op, v, zc) in enumerate(CASES):
                got = hip_ir(op, re, im, n, value_of(v, n), bool(zc), prec)
                want = (gold[f"{op}_{prec}_{l2}_{ci}_re"], gold[f"{op}_{prec}_{l2}_{ci}_im"])
                if op in ("copy", "time_reverse") or (op == "delay" and v == 0.0):
                    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (op, l2)
                else:
                    assert err(got, want) <= t, (op, v, zc, l2, prec, err(got, want))
        for k, (size, ph, tm) in enumerate(CP_CASES):
            got = sp.change_phase(gold[f"cp_{prec}_{k}_x"], ph, tm)
            want = gold[f"cp_{prec}_{k}_y"]
            assert got.size == want.size and got.dtype == want.dtype
            assert err(got, want) <= t, (k, prec, err(got, want))


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_gpu_first_principles(prec):
    check_first_principles(lambda op, re, im, n, v, zc: hip_ir(op, re, im, n, v, zc, prec), prec, 2e-6 if prec == "f32" else 1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("l2", [3, 4, 7, 12, 14, 15, 16, 18])
def test_gpu_ir_phase_every_mode_against_oracle(oracle, prec, l2):
    """The eight ir_phase cases of the reference's IR_Manipulation_Tester (main.cpp:127-147: Zero/Center x Mix/Min/Max/Lin),
    at sizes on both sides of the LDS limit (one fused kernel below it, four-step transforms above)."""
    dt = np.float32 if prec == "f32" else np.float64
    t = (TOL32 if prec == "f32" else TOL64) * (1 if l2 <= 14 else 4)
    rng = np.random.default_rng(l2)
    n = 1 << l2
    x = (rng.uniform(-1, 1, n) * np.exp(-np.arange(n) / (n / 16))).astype(dt)
    re, im = oracle.fft_surface("rfft_zip", prec, l2, x)
    for phase, zero in ((0.1, True), (0.9, False), (0.0, True), (0.0, False), (1.0, True), (1.0, False), (0.5, True), (0.5, False)):
        want = oracle.ir_op("phase", re, im, n, phase, zero, prec)
        got = hip_ir("phase", re, im, n, phase, zero, prec)
        assert err(got, want) <= t, (phase, zero, err(got, want))


@pytest.mark.gpu
def test_gpu_batched_and_strided_rows(oracle):
    """2-D input = one launch; rows must not leak into each other."""
    import hisstools_library_amd.spectral_functions as S
    rng = np.random.default_rng(2)
    for l2, batch in ((6, 300), (10, 33), (13, 5)):
        n = 1 << l2
        re = rng.uniform(-1, 1, (batch, n // 2)).astype(np.float32)
        im = rng.uniform(-1, 1, (batch, n // 2)).astype(np.float32)
        for op, fn, v in (("delay", lambda: S.ir_delay(re, im, n, 2.5), 2.5), ("phase", lambda: S.ir_phase(re, im, n, 0.0), 0.0),
                          ("phase", lambda: S.ir_phase(re, im, n, 0.7, True), 0.7)):
            gr, gi = fn()
            for r in (0, batch // 2, batch - 1):
                want = oracle.ir_op(op, re[r], im[r], n, v, op == "phase" and v == 0.7, "f32")
                assert err((gr[r], gi[r]), want) <= TOL32, (op, l2, r)


@pytest.mark.gpu
def test_gpu_far_bins_of_a_spike_stay_accurate(oracle):
    """Bin 2^21 of a spike near the end of a 2^23-point frame: the phase is ~1e7 rad; the reduction must not lose it."""
    n, l2 = 1 << 23, 23
    pos = n - 1.375
    want = oracle.ir_op("spike", None, None, n, pos, False, "f64")
    got = hip_ir("spike", None, None, n, pos, False, "f64")
    assert err(got, want) <= 1e-11
    got32 = hip_ir("spike", None, None, n, pos, False, "f32")
    assert err(got32, want) <= 1e-7


@pytest.mark.gpu
def test_gpu_bad_calls_fail_loudly():
    import hisstools_library_amd.spectral_functions as S
    with pytest.raises(ValueError):
        S.ir_copy(np.zeros(8, np.float32), np.zeros(8, np.float32), 24)
    with pytest.raises(RuntimeError):
        S.ir_phase(np.zeros(2, np.float32), np.zeros(2, np.float32), 4, 0.0)            # minimum phase needs >= 8 samples


# ------------------------------------------------------------------------------------------------ the IR products (SpectralFunctions.hpp:415-436)
# ir_convolve_complex / ir_convolve_real / ir_correlate_complex / ir_correlate_real: golden_ir2_v1.npz is made by the UNMODIFIED reference
# (tests/golden/make_golden_ir2.py over oracle/ref_spectral_driver.cpp).  Products, sums and the scale are each rounded on their own in the
# reference's vector layer, in the oracle and in the HIP kernel: BIT-identical, float and double.

from make_golden_ir2 import COUNTS as P_COUNTS, SCALES as P_SCALES  # noqa: E402
P_OPS = ("convolve_complex", "convolve_real", "correlate_complex", "correlate_real")


@pytest.fixture(scope="module")
def gold2():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_ir2_v1.npz"))


def _product_cases(gold2):
    for prec in ("f32", "f64"):
        for n in P_COUNTS:
            ops = [gold2[f"in_{prec}_{n}_{name}"] for name in "abcd"]
            for op in P_OPS:
                fs = n if op.endswith("complex") else 2 * n
                for k, sc in enumerate(P_SCALES):
                    yield prec, n, op, fs, sc, ops, (gold2[f"{op}_{prec}_{n}_{k}_re"], gold2[f"{op}_{prec}_{n}_{k}_im"])


def test_oracle_ir_products_match_reference_vectors(oracle, gold2):
    count = 0
    for prec, n, op, fs, sc, ops, want in _product_cases(gold2):
        got = oracle.ir_product(op, *ops, fs, sc, prec)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (prec, n, op, sc)
        assert np.array_equal(np.signbit(got[0][:1]), np.signbit(want[0][:1]))         # (the signed zero through bin 0)
        count += 1
    assert count == 2 * len(P_COUNTS) * 4 * len(P_SCALES)


def test_oracle_ir_products_first_principles(oracle):
    rng = np.random.default_rng(3)
    n = 64
    a, b, c, d = (rng.uniform(-1, 1, n) for _ in range(4))
    z1, z2 = a + 1j * b, c + 1j * d
    re, im = oracle.ir_product("convolve_complex", a, b, c, d, n, 0.25, "f64")
    assert np.allclose(re + 1j * im, 0.25 * z1 * z2, rtol=0, atol=1e-15)
    re, im = oracle.ir_product("correlate_complex", a, b, c, d, n, 2.0, "f64")
    assert np.allclose(re + 1j * im, 2.0 * z1 * np.conj(z2), rtol=0, atol=1e-14)
    # the real forms: bin 0 carries (DC, Nyquist), each a real product; the other bins as the complex form
    re, im = oracle.ir_product("convolve_real", a, b, c, d, 2 * n, 1.0, "f64")
    assert re[0] == a[0] * c[0] and im[0] == b[0] * d[0]
    assert np.allclose((re + 1j * im)[1:], (z1 * z2)[1:], rtol=0, atol=1e-15)
    re, im = oracle.ir_product("correlate_real", a, b, c, d, 2 * n, 1.0, "f64")
    assert re[0] == a[0] * c[0] and im[0] == b[0] * d[0]
    assert np.allclose((re + 1j * im)[1:], (z1 * np.conj(z2))[1:], rtol=0, atol=1e-15)


@pytest.mark.gpu
def test_gpu_ir_products_match_reference_vectors_bit_for_bit(gold2):
    import hisstools_library_amd.spectral_functions as S
    fns = {"convolve_complex": S.ir_convolve_complex, "convolve_real": S.ir_convolve_real, "correlate_complex": S.ir_correlate_complex,
           "correlate_real": S.ir_correlate_real}
    count = 0
    for prec, n, op, fs, sc, ops, want in _product_cases(gold2):
        got = fns[op](*ops, fs, sc)
        assert got[0].dtype == want[0].dtype
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (prec, n, op, sc)
        assert np.array_equal(np.signbit(got[0][:1]), np.signbit(want[0][:1]))
        count += 1
    assert count == 2 * len(P_COUNTS) * 4 * len(P_SCALES)


@pytest.mark.gpu
def test_gpu_ir_products_batched_and_one_filter_for_many(oracle):
    """a batch of spectra against one spectrum each, and against ONE spectrum for all rows (a filter applied to many): row by row the oracle's"""
    import hisstools_library_amd.spectral_functions as S
    rng = np.random.default_rng(9)
    rows, n = 7, 256
    a, b = rng.uniform(-1, 1, (rows, n)).astype(np.float32), rng.uniform(-1, 1, (rows, n)).astype(np.float32)
    c, d = rng.uniform(-1, 1, (rows, n)).astype(np.float32), rng.uniform(-1, 1, (rows, n)).astype(np.float32)
    re, im = S.ir_convolve_real(a, b, c, d, 2 * n, 0.5)
    for r in range(rows):
        w = oracle.ir_product("convolve_real", a[r], b[r], c[r], d[r], 2 * n, 0.5, "f32")
        assert np.array_equal(re[r], w[0]) and np.array_equal(im[r], w[1]), r
    re, im = S.ir_correlate_complex(a, b, c[0], d[0], n, 1.0)
    for r in range(rows):
        w = oracle.ir_product("correlate_complex", a[r], b[r], c[0], d[0], n, 1.0, "f32")
        assert np.array_equal(re[r], w[0]) and np.array_equal(im[r], w[1]), r


@pytest.mark.gpu
def test_gpu_ir_products_bad_calls_fail_loudly():
    import hisstools_library_amd.spectral_functions as S
    z = np.zeros(24, np.float32)
    with pytest.raises(ValueError):
        S.ir_convolve_complex(z, z, z, z, 24)                   # not a power of two
    with pytest.raises(ValueError):
        S.ir_convolve_real(z[:4], z[:4], z[:4], z[:4], 16)      # too few values
