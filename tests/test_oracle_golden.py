"""The CPU oracle (oracle/hcv_oracle.c) against the golden vectors produced by the unmodified reference
(tests/golden/make_golden.py).  Runs without a GPU and without /root/reference.

Where the reference API lets the FFT phase be pinned the oracle is BIT-IDENTICAL to the reference; through the
NToMonoConvolve / Convolver API the reference draws random phases, so those compare to rounding."""
import numpy as np
import pytest

from conftest import rel_err


@pytest.mark.parametrize("l2", [5, 8, 10, 12, 14])
def test_fft_format_vectors(oracle, golden, l2):
    re, im = oracle.rfft(golden[f"fft{l2}_x"], l2)
    assert np.array_equal(re, golden[f"fft{l2}_re"]) and np.array_equal(im, golden[f"fft{l2}_im"])
    assert np.array_equal(oracle.rifft(re, im, l2), golden[f"fft{l2}_inv"])


def test_fft_odd_short_input(oracle, golden):
    re, im = oracle.rfft(golden["fft8_odd_x"], 8)
    assert np.array_equal(re, golden["fft8_odd_re"]) and np.array_equal(im, golden["fft8_odd_im"])


def test_fft_conventions_vs_numpy(oracle, golden):
    # realp[k] = 2 Re X[k], imagp[k] = 2 Im X[k], realp[0] = 2 X[0], imagp[0] = 2 X[N/2]; rifft(rfft(x)) = 2N x
    x = golden["fft10_x"]
    n = x.size
    X = np.fft.rfft(x.astype(np.float64)) * 2
    re, im = oracle.rfft(x, 10)
    tr, ti = X.real[: n // 2].copy(), X.imag[: n // 2].copy()
    ti[0] = X.real[n // 2]
    assert rel_err(re, tr) < 5e-7 and np.abs(im - ti).max() / np.abs(X).max() < 5e-7
    assert np.abs(oracle.rifft(re, im, 10) / (2 * n) - x).max() < 1e-6


@pytest.mark.parametrize("name,N,ro,blocks", [("part256", 256, 0, 512), ("part1024", 1024, 77, [1, 7, 333, 2000, 64])])
def test_partitioned(oracle, golden, name, N, ro, blocks):
    h, x = golden[f"{name}_ir"], golden[f"{name}_x"]
    p = oracle.PartitionedConvolve(N, h.size, 0, 0)
    p.setResetOffset(ro)
    assert p.set(h) == 0
    assert np.array_equal(p.run(x, blocks), golden[f"{name}_y"])
    # bit-identical for any call pattern at a fixed reset offset
    p.reset()
    assert np.array_equal(p.run(x, 64), golden[f"{name}_y"])


def test_partitioned_window(oracle, golden):
    p = oracle.PartitionedConvolve(256, 1024, 300, 500)
    p.setResetOffset(0)
    assert p.set(golden["partwin_ir"]) == 0
    assert np.array_equal(p.run(golden["partwin_x"], 256), golden["partwin_y"])


@pytest.mark.parametrize("Lh", [1, 16, 128, 2044])
def test_time_domain(oracle, golden, Lh):
    t = oracle.TimeDomainConvolve(0, Lh)
    t.set(golden["td_ir"])
    assert np.array_equal(t.run(golden["td_x"], 512), golden[f"td{Lh}_y"])


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_mono_latency_modes(oracle, golden, mode):
    m = oracle.MonoConvolve(16384, latency=mode)
    m.setResetOffset(0)
    assert m.set(golden["mono_ir"], True) == 0
    assert np.array_equal(m.run(golden["mono_x"], 512), golden[f"mono{mode}_y"])


def test_mono_custom(oracle, golden):
    m = oracle.MonoConvolve(11000, zeroLatency=False, A=512, B=2048)
    m.setResetOffset(0)
    assert m.set(golden["mono_ir"], False) == 0
    assert np.array_equal(m.run(golden["mono_x"], [100, 900, 2048]), golden["monoc_y"])


def test_ntomono(oracle, golden):
    c = oracle.NToMonoConvolve(3, 16384, 0)
    for i in range(3):
        assert c.set(i, golden["n2m_irs"][i], True) == 0
    assert rel_err(c.run(golden["n2m_x"], 512), golden["n2m_y"]) < 5e-6


def test_convolver_matrix_and_parallel(oracle, golden):
    irs = golden["conv_irs"]
    c = oracle.Convolver(2, 3, 0)
    for o in range(3):
        for i in range(2):
            assert c.set(i, o, irs[o, i], True) == 0
    y = c.run(golden["conv_x"], 3, 512)
    for o in range(3):
        assert rel_err(y[o], golden["conv_y"][o]) < 5e-6
    c = oracle.Convolver(3, None, 1)
    for o in range(3):
        assert c.set(o, o, golden["par_irs"][o], True) == 0
    y = c.run(golden["par_x"], 3, 256)
    for o in range(3):
        assert rel_err(y[o], golden["par_y"][o]) < 5e-6


def test_golden_matches_float64_truth(golden):
    # the fixtures themselves are sane: reference output == float64 linear convolution to ~3e-7 of peak
    from scipy.signal import fftconvolve
    h, x = golden["mono_ir"].astype(np.float64), golden["mono_x"].astype(np.float64)
    truth = fftconvolve(x, h)[: x.size]
    for mode, lat in ((0, 0), (1, 128), (2, 512)):
        t = np.concatenate([np.zeros(lat), truth])[: x.size]
        assert rel_err(golden[f"mono{mode}_y"], t) < 2e-6


def test_synth_generator_is_mt19937(oracle):
    # the C generator is bit-identical to numpy's MT19937 raw stream
    bg = np.random.MT19937(0)
    bg._legacy_seeding(777 + 5)
    raw = bg.random_raw(1000).astype(np.uint64)
    expect = (2.0 * ((raw >> np.uint64(8)).astype(np.float64) / 16777216.0) - 1.0).astype(np.float32)
    assert np.array_equal(oracle.synth_audio(5, 1000), expect)


def test_reference_defect_zero_latency_three_sizes_is_restated(oracle):
    """MonoConvolve(zeroLatency, A, B, C): mPart1 is null so part2 overwrites the head's output
    (MonoConvolve.cpp:195-197).  The oracle restates the reference bug-for-bug (and is bit-identical to it where the
    compiled reference is available); the HIP engine deliberately does not reproduce it."""
    from scipy.signal import fftconvolve
    h, x = oracle.synth_ir(2, 2, 9000), oracle.synth_audio(4, 12000)
    m = oracle.MonoConvolve(9000, zeroLatency=True, A=256, B=1024, C_=4096)
    m.setResetOffset(0)
    assert m.set(h, False) == 0
    y = m.run(x, 512)
    truth = fftconvolve(x.astype(np.float64), h.astype(np.float64))[: x.size]
    head = fftconvolve(x.astype(np.float64), h[:128].astype(np.float64))[: x.size]
    assert rel_err(y, truth) > 1e-3
    assert rel_err(y + head, truth) < 2e-6
    if oracle.have_ref():
        r = oracle.MonoConvolve(9000, zeroLatency=True, A=256, B=1024, C_=4096, backend="ref")
        r.setResetOffset(0)
        r.set(h, False)
        assert np.array_equal(r.run(x, 512), y)
