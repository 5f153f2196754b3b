"""Parity of the HIP path (through the C ABI) against the CPU oracle, the golden vectors produced by the
unmodified reference, and float64 ground truth.  Needs a real MI355X:  pytest -m gpu

Stated tolerance: per output channel  max|y - y_ref| <= TOL * max|y_ref|
    TOL = 2e-6  for IRs up to 2 s (what the reference itself achieves against float64 is ~3e-7)
    TOL = 1e-5  for many-input sums and the long-IR stress shapes
The GPU path shares one forward FFT per input and one inverse FFT per output and reduces (input, partition)
products in a different order than the reference, so results agree to rounding, not bitwise.
"""
import functools
import os
import types

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = 2e-6
TOL_SUM = 1e-5


@pytest.fixture(scope="module")
def H():
    import hisstools_library_amd as H
    assert H.load().hcv_device_count() > 0, "no GPU visible: the HIP path cannot run (and there is no fallback)"
    return H


def truth_conv(x, h, latency=0):
    from scipy.signal import fftconvolve
    y = fftconvolve(np.asarray(x, np.float64), np.asarray(h, np.float64))[: len(x)]
    return np.concatenate([np.zeros(latency), y])[: len(x)]


# ------------------------------------------------------------------------------------------- FFT (K2/K3)

@pytest.mark.parametrize("l2", list(range(5, 21)))
def test_rfft_rifft_vs_oracle(H, oracle, l2):
    n = 1 << l2
    x = np.stack([oracle.synth_audio(40 + b, n) for b in range(3)])
    re, im = H.hisstools_rfft(x, l2)
    for b in range(3):
        ore, oim = oracle.rfft(x[b], l2)
        scale = max(np.abs(ore).max(), np.abs(oim).max())
        assert np.abs(re[b] - ore).max() / scale < 1e-6 and np.abs(im[b] - oim).max() / scale < 1e-6
        # vDSP packing against numpy in float64
        X = np.fft.rfft(x[b].astype(np.float64)) * 2
        assert abs(re[b][0] - X.real[0]) / scale < 1e-6 and abs(im[b][0] - X.real[n // 2]) / scale < 1e-6
    inv = H.hisstools_rifft(re, im, l2)
    for b in range(3):
        assert np.abs(inv[b] / (2 * n) - x[b]).max() < 2e-6          # rifft(rfft(x)) = 2N x
        assert rel_err(inv[b], oracle.rifft(re[b], im[b], l2)) < 1e-6


def test_rfft_zero_padding_and_odd_length(H, oracle, golden):
    re, im = H.hisstools_rfft(golden["fft8_odd_x"], 8)
    scale = np.abs(golden["fft8_odd_re"]).max()
    assert np.abs(re - golden["fft8_odd_re"]).max() / scale < 1e-6
    assert np.abs(im - golden["fft8_odd_im"]).max() / scale < 1e-6


@pytest.mark.parametrize("l2", [5, 8, 10, 12, 14])
def test_fft_golden_vectors(H, golden, l2):
    re, im = H.hisstools_rfft(golden[f"fft{l2}_x"], l2)
    scale = max(np.abs(golden[f"fft{l2}_re"]).max(), np.abs(golden[f"fft{l2}_im"]).max())
    assert np.abs(re - golden[f"fft{l2}_re"]).max() / scale < 1e-6
    assert np.abs(im - golden[f"fft{l2}_im"]).max() / scale < 1e-6
    assert rel_err(H.hisstools_rifft(golden[f"fft{l2}_re"], golden[f"fft{l2}_im"], l2), golden[f"fft{l2}_inv"]) < 1e-6


# ------------------------------------------------------------------------------------------- golden vectors end to end

@pytest.mark.parametrize("name,N,blocks", [("part256", 256, 512), ("part1024", 1024, [1, 7, 333, 2000, 64])])
def test_golden_partitioned(H, golden, name, N, blocks):
    p = H.PartitionedConvolve(N, golden[f"{name}_ir"].size, 0, 0)
    assert p.set(golden[f"{name}_ir"]) == 0
    assert rel_err(p.run(golden[f"{name}_x"], blocks), golden[f"{name}_y"]) < TOL


def test_golden_partitioned_window(H, golden):
    p = H.PartitionedConvolve(256, 1024, 300, 500)
    assert p.set(golden["partwin_ir"]) == 0
    assert rel_err(p.run(golden["partwin_x"], 256), golden["partwin_y"]) < TOL


@pytest.mark.parametrize("Lh", [1, 16, 128, 2044])
def test_golden_time_domain(H, golden, Lh):
    t = H.TimeDomainConvolve(0, Lh)
    t.set(golden["td_ir"])
    # a 2044-term float32 dot product: the reference's own summation order is ~2e-6 of peak away from float64
    assert rel_err(t.run(golden["td_x"], 512), golden[f"td{Lh}_y"]) < (TOL if Lh <= 128 else TOL_SUM)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_golden_mono(H, golden, mode):
    m = H.MonoConvolve(16384, latency=mode)
    assert m.set(golden["mono_ir"], True) == 0
    assert rel_err(m.run(golden["mono_x"], 512), golden[f"mono{mode}_y"]) < TOL


def test_golden_mono_custom(H, golden):
    m = H.MonoConvolve(11000, zeroLatency=False, A=512, B=2048)
    assert m.set(golden["mono_ir"], False) == 0
    assert rel_err(m.run(golden["mono_x"], [100, 900, 2048]), golden["monoc_y"]) < TOL


def test_golden_ntomono(H, golden):
    c = H.NToMonoConvolve(3, 16384, 0)
    for i in range(3):
        assert c.set(i, golden["n2m_irs"][i], True) == 0
    assert rel_err(c.run(golden["n2m_x"], 512), golden["n2m_y"]) < TOL_SUM


def test_golden_convolver(H, golden):
    irs = golden["conv_irs"]
    c = H.Convolver(2, 3, 0)
    for o in range(3):
        for i in range(2):
            assert c.set(i, o, irs[o, i], True) == 0
    y = c.run(golden["conv_x"], 3, 512)
    for o in range(3):
        assert rel_err(y[o], golden["conv_y"][o]) < TOL_SUM
    c = H.Convolver(3, None, 1)
    for o in range(3):
        assert c.set(o, o, golden["par_irs"][o], True) == 0
    y = c.run(golden["par_x"], 3, 256)
    for o in range(3):
        assert rel_err(y[o], golden["par_y"][o]) < TOL_SUM


# ------------------------------------------------------------------------------------------- HIP vs oracle on seeded inputs

@pytest.mark.parametrize("N,L,S,block", [
    (32, 200, 1500, 64), (64, 1000, 3000, [5, 300, 17]), (256, 5000, 12000, 512), (1024, 20000, 30000, 2048),
    (4096, 48000, 40000, 4096), (16384, 48000, 70000, 8192), (32768, 70000, 100000, 5000),
    (65536, 150000, 250000, 16384), (1 << 18, 300000, 700000, 30000), (1 << 20, 600000, 1700000, 32768)])
def test_partitioned_vs_oracle(H, oracle, N, L, S, block):
    h, x = oracle.synth_ir(1, 2, L), oracle.synth_audio(3, S)
    ref = oracle.PartitionedConvolve(N, L, 0, 0)
    ref.setResetOffset(0)
    assert ref.set(h) == 0
    gpu = H.PartitionedConvolve(N, L, 0, 0)
    assert gpu.set(h) == 0
    y_ref, y = ref.run(x, 2048 if not isinstance(block, int) or block > 2048 else block), gpu.run(x, block)
    assert rel_err(y, y_ref) < TOL
    assert rel_err(y, truth_conv(x, h, N // 2)) < TOL
    assert (y[: N // 2] == 0).all()                              # latency exactly N/2


def test_partitioned_block_size_independence(H, oracle):
    h, x = oracle.synth_ir(5, 5, 6000), oracle.synth_audio(6, 20000)
    outs = []
    for block in (64, 512, [1, 7, 333, 5000, 4096], 20000):
        p = H.PartitionedConvolve(1024, 6000, 0, 0)
        p.set(h)
        outs.append(p.run(x, block))
    for y in outs[1:]:
        assert rel_err(y, outs[0]) < 1e-6                        # same maths, reduction split may differ with batch size


def test_partitioned_fft_size_change_and_reload(H, oracle):
    h, x = oracle.synth_ir(7, 0, 3000), oracle.synth_audio(7, 9000)
    p = H.PartitionedConvolve(4096, 4096, 0, 0)
    assert p.set(h) == 0
    y1 = p.run(x, 512)
    assert rel_err(y1, truth_conv(x, h, 2048)) < TOL
    assert p.setFFTSize(512) == 0
    assert p.process(x[:100])[0] is False                         # no partitions until the next set (.cpp:131-154)
    assert p.set(h) == 0
    assert rel_err(p.run(x, 512), truth_conv(x, h, 256)) < TOL


@pytest.mark.parametrize("Lh,block", [(1, 512), (16, 64), (128, 512), (128, 5000), (2044, 4096), (777, [3, 1000, 4099])])
def test_time_domain_vs_oracle(H, oracle, Lh, block):
    h, x = oracle.synth_ir(2, 1, 2044), oracle.synth_audio(9, 15000)
    ref = oracle.TimeDomainConvolve(0, Lh)
    ref.set(h)
    gpu = H.TimeDomainConvolve(0, Lh)
    gpu.set(h)
    y = gpu.run(x, block)
    tol = TOL if Lh <= 128 else TOL_SUM                          # long direct sums: float32 summation-order noise
    assert rel_err(y, ref.run(x, 512)) < tol
    assert rel_err(y, truth_conv(x, h[:Lh])) < tol               # plain causal FIR for any call size (no ring-wrap defect)


@pytest.mark.parametrize("mode,lat", [(0, 0), (1, 128), (2, 512)])
def test_mono_modes_vs_oracle_and_truth(H, oracle, mode, lat):
    h, x = oracle.synth_ir(0, 0, 48000), oracle.synth_audio(0, 60000)      # config-1 shape: 1 s @ 48 kHz
    ref = oracle.MonoConvolve(48000, latency=mode)
    ref.setResetOffset(0)
    assert ref.set(h, True) == 0
    gpu = H.MonoConvolve(48000, latency=mode)
    assert gpu.set(h, True) == 0
    y, y_ref = gpu.run(x, 512), ref.run(x, 512)
    assert rel_err(y, y_ref) < TOL
    assert rel_err(y, truth_conv(x, h, lat)) < TOL


def test_mono_single_stage_config1(H, oracle):
    # BASELINE config 1: MonoConvolve 1x1, one 16384-point stage, 1 s @ 48 kHz IR
    h, x = oracle.synth_ir(0, 0, 48000), oracle.synth_audio(0, 80000)
    ref = oracle.MonoConvolve(48000, zeroLatency=False, A=16384)
    ref.setResetOffset(0)
    ref.set(h, False)
    gpu = H.MonoConvolve(48000, zeroLatency=False, A=16384)
    assert gpu.set(h, False) == 0
    assert rel_err(gpu.run(x, 2048), ref.run(x, 2048)) < TOL


def test_mono_custom_large_fft_stage(H, oracle):
    # custom partitioning with a four-step-FFT tail: MonoConvolve(L, false, 256, 4096, 131072)
    h, x = oracle.synth_ir(2, 2, 200000), oracle.synth_audio(4, 330000)
    ref = oracle.MonoConvolve(200000, zeroLatency=False, A=256, B=4096, C_=131072)
    ref.setResetOffset(0)
    assert ref.set(h, False) == 0
    gpu = H.MonoConvolve(200000, zeroLatency=False, A=256, B=4096, C_=131072)
    assert gpu.set(h, False) == 0
    y = gpu.run(x, 8192)
    assert rel_err(y, ref.run(x, 2048)) < TOL_SUM
    assert rel_err(y, truth_conv(x, h, 128)) < TOL_SUM


def test_zero_latency_with_fewer_than_four_sizes_keeps_the_head(H, oracle):
    """Reference defect (not reproduced): with zeroLatency and fewer than four FFT sizes mPart1 is null, so
    processAndSum(mPart2, ..., accumulate || mPart1) OVERWRITES the time-domain head's output
    (MonoConvolve.cpp:195-197) and the first A/2 taps of the IR are lost.  The engine sums all stages."""
    h, x = oracle.synth_ir(2, 2, 30000), oracle.synth_audio(4, 50000)
    gpu = H.MonoConvolve(30000, zeroLatency=True, A=256, B=4096, C_=16384)
    assert gpu.set(h, False) == 0
    y = gpu.run(x, 2048)
    assert rel_err(y, truth_conv(x, h)) < TOL
    ref = oracle.MonoConvolve(30000, zeroLatency=True, A=256, B=4096, C_=16384)
    ref.setResetOffset(0)
    ref.set(h, False)
    y_ref = ref.run(x, 2048)
    head_only = truth_conv(x, np.concatenate([h[:128], np.zeros(h.size - 128, np.float32)]))
    assert rel_err(y_ref + head_only, truth_conv(x, h)) < TOL     # the reference is missing exactly the head


def test_reference_quirks_mode_loses_the_head_as_the_reference_does(H, oracle, monkeypatch):
    """HCV_REFERENCE_QUIRKS=1 (opt-in, read when the object is made): the same layout now gives the REFERENCE's stream — the head's output
    overwritten by the first FFT stage (MonoConvolve.cpp:195-197), as the oracle restates it bit for bit — within the float32 tolerance;
    an IR that ends inside the head keeps it (that stage holds no partitions and touches nothing), and accumulate = true sums everything."""
    monkeypatch.setenv("HCV_REFERENCE_QUIRKS", "1")
    for sizes in ((256, 4096, 16384), (1024, 16384, 0), (4096, 0, 0)):
        h, x = oracle.synth_ir(2, 2, 30000), oracle.synth_audio(4, 50000)
        gpu = H.MonoConvolve(30000, zeroLatency=True, A=sizes[0], B=sizes[1], C_=sizes[2])
        ref = oracle.MonoConvolve(30000, zeroLatency=True, A=sizes[0], B=sizes[1], C_=sizes[2])
        ref.setResetOffset(0)
        assert gpu.set(h, False) == 0 and ref.set(h, False) == 0
        y, y_ref = gpu.run(x, 2048), ref.run(x, 2048)
        assert rel_err(y, y_ref) < TOL, (sizes, rel_err(y, y_ref))
        assert rel_err(y, truth_conv(x, h)) > 1e-3                  # (and it really is missing the head)
    # an IR shorter than the head: no FFT stage is loaded, nothing overwrites the head
    h, x = oracle.synth_ir(1, 1, 100), oracle.synth_audio(2, 20000)
    gpu = H.MonoConvolve(30000, zeroLatency=True, A=256, B=4096, C_=16384)
    ref = oracle.MonoConvolve(30000, zeroLatency=True, A=256, B=4096, C_=16384)
    ref.setResetOffset(0)
    assert gpu.set(h, False) == 0 and ref.set(h, False) == 0
    assert rel_err(gpu.run(x, 1024), ref.run(x, 1024)) < TOL
    assert rel_err(gpu.run(x, 1024), truth_conv(x, h)[: x.size]) < 1.0
    # four sizes: mPart1 exists, nothing is lost, quirk mode or not
    h, x = oracle.synth_ir(2, 2, 30000), oracle.synth_audio(4, 50000)
    gpu = H.MonoConvolve(30000, zeroLatency=True, A=256, B=1024, C_=4096, D=16384)
    assert gpu.set(h, False) == 0
    assert rel_err(gpu.run(x, 2048), truth_conv(x, h)) < TOL


def test_mono_impulse_is_exact(H):
    # delta in -> the IR comes out sample-exact in the zero-latency chain
    h = np.zeros(40000, np.float32)
    for k, v in ((0, 0.5), (5, -0.25), (130, 0.75), (600, 0.3), (9000, -0.6), (30000, 0.9)):
        h[k] = v
    m = H.MonoConvolve(40000, latency=0)
    assert m.set(h, True) == 0
    x = np.zeros(45000, np.float32)
    x[0] = 1.0
    y = m.run(x, 512)
    assert np.abs(y[:40000] - h).max() < 2e-7


def test_mono_accumulate(H, oracle):
    h, x = oracle.synth_ir(1, 1, 300), oracle.synth_audio(2, 2000)
    m = H.MonoConvolve(16384, latency=0)
    m.set(h, False)
    y = m.process(x)
    m.reset()
    ya = m.process(x, out=np.full(x.size, 2.0, np.float32), accumulate=True)
    assert np.abs(ya - (y + 2.0)).max() < 1e-6


def test_ntomono_vs_oracle(H, oracle):
    nin, L, S = 8, 24000, 30000                                   # config-3 shape at 1/10 length
    irs = [oracle.synth_ir(i, 0, L) for i in range(nin)]
    xs = np.stack([oracle.synth_audio(i, S) for i in range(nin)])
    ref = oracle.NToMonoConvolve(nin, 16384, 0)
    ref.setResetOffset(0)
    gpu = H.NToMonoConvolve(nin, 16384, 0)
    for i in range(nin):
        assert ref.set(i, irs[i], True) == 0 and gpu.set(i, irs[i], True) == 0
    y = gpu.run(xs, 1024)
    assert rel_err(y, ref.run(xs, 1024)) < TOL_SUM
    truth = sum(truth_conv(xs[i], irs[i]) for i in range(nin))
    assert rel_err(y, truth) < TOL_SUM
    # activeIns < numIns uses only the leading inputs (NToMonoConvolve.cpp:41)
    gpu.reset(0)
    for i in range(nin):
        gpu.reset(i)
    y3 = gpu.run(xs, 1024, activeIns=3)
    assert rel_err(y3, sum(truth_conv(xs[i], irs[i]) for i in range(3))) < TOL_SUM


def test_convolver_matrix_vs_oracle(H, oracle):
    nin, nout, L, S = 5, 3, 12000, 20000                          # ragged: not a multiple of any output tile
    irs = {(i, o): oracle.synth_ir(i, o, L - 100 * i - 10 * o) for i in range(nin) for o in range(nout)}
    xs = np.stack([oracle.synth_audio(i, S) for i in range(nin)])
    ref = oracle.Convolver(nin, nout, 1)
    ref.setResetOffset(0)
    gpu = H.Convolver(nin, nout, 1)
    for (i, o), h in irs.items():
        assert ref.set(i, o, h, True) == 0 and gpu.set(i, o, h, True) == 0
    y, y_ref = gpu.run(xs, nout, 512), ref.run(xs, nout, 512)
    for o in range(nout):
        assert rel_err(y[o], y_ref[o]) < TOL_SUM
        truth = sum(truth_conv(xs[i], irs[(i, o)], 128) for i in range(nin))
        assert rel_err(y[o], truth) < TOL_SUM


def test_convolver_double_api_and_parallel(H, oracle):
    n, L, S = 4, 5000, 9000
    irs = [oracle.synth_ir(o, o, L).astype(np.float64) for o in range(n)]
    xs = np.stack([oracle.synth_audio(50 + o, S) for o in range(n)]).astype(np.float64)
    gpu = H.Convolver(n, None, 0)
    ref = oracle.Convolver(n, None, 0)
    ref.setResetOffset(0)
    for o in range(n):
        assert gpu.set(o, o, irs[o], True) == 0 and ref.set(o, o, irs[o], True) == 0
    y = gpu.run(xs, n, 256)
    assert y.dtype == np.float64
    y_ref = ref.run(xs, n, 256)
    for o in range(n):
        assert rel_err(y[o], y_ref[o]) < TOL
        assert rel_err(y[o], truth_conv(xs[o], irs[o])) < TOL


def test_head_paths_agree_on_mixed_block_sizes(H, oracle):
    """>= 16 pairs: hop-aligned blocks take the head through the first stage's FFTs, ragged blocks through the
    direct-form FIR kernel; any mixture must give the same stream (and match the oracle)."""
    nin = nout = 4
    L, S = 3000, 12000
    irs = {(i, o): oracle.synth_ir(i, o, L) for i in range(nin) for o in range(nout)}
    xs = np.stack([oracle.synth_audio(i, S) for i in range(nin)])
    ref = oracle.Convolver(nin, nout, 0)
    ref.setResetOffset(0)
    outs = []
    for blocks in (512, [512, 100, 412, 1024, 37, 91, 128], 128, [1000, 24]):
        c = H.Convolver(nin, nout, 0)
        for (i, o), h in irs.items():
            assert c.set(i, o, h, True) == 0
        outs.append(c.run(xs, nout, blocks))
    for (i, o), h in irs.items():
        assert ref.set(i, o, h, True) == 0
    y_ref = ref.run(xs, nout, 512)
    for y in outs:
        for o in range(nout):
            assert rel_err(y[o], y_ref[o]) < TOL_SUM
            assert rel_err(y[o], sum(truth_conv(xs[i], irs[(i, o)]) for i in range(nin))) < TOL_SUM


def test_small_blocks_use_deferred_tail_and_match(H, oracle):
    """Calls shorter than the tail hop: partitions 1..P-1 are accumulated in the background between hop boundaries
    (the engine's form of the reference's time-spread scheduler, PartitionedConvolve.cpp:321-348); the output must not
    depend on it."""
    h, x = oracle.synth_ir(6, 6, 70000), oracle.synth_audio(6, 120000)
    outs = []
    for block in (128, 1000, 8192, 30000):
        m = H.MonoConvolve(70000, latency=1)
        assert m.set(h, True) == 0
        outs.append(m.run(x, block))
    truth = truth_conv(x, h, 128)
    for y in outs:
        assert rel_err(y, truth) < TOL
    # a reset in the middle of a hop drops the pre-accumulated spectra
    m = H.MonoConvolve(70000, latency=1)
    m.set(h, True)
    m.run(x[:20000], 256)
    m.reset()
    assert rel_err(m.run(x, 256), truth) < TOL


def test_deferred_tail_slices_with_ragged_calls_and_live_changes(H, oracle):
    """A tail of 25 partitions (more than the 16 background slices) fed with call sizes that neither divide the hop nor
    stay constant, a switch between small and hop-sized calls, an IR swapped in mid-hop and a partial-matrix call: the
    slice schedule is an implementation detail and must never show in the output."""
    L = 200_000
    S = 420_000
    irs = [[oracle.synth_ir(i, o, L) for o in range(2)] for i in range(2)]
    xs = np.stack([oracle.synth_audio(50 + i, S) for i in range(2)])
    ref, gpu = oracle.Convolver(2, 2, 0), H.Convolver(2, 2, 0)
    for i in range(2):
        for o in range(2):
            assert ref.set(i, o, irs[i][o], True) == 0 and gpu.set(i, o, irs[i][o], True) == 0
    y_ref = ref.run(xs, 2, 2048)
    pattern = [100, 37, 128, 1000, 64, 3, 511, 4096, 8192, 20000, 77]
    y = gpu.run(xs, 2, pattern)
    for o in range(2):
        assert rel_err(y[o], y_ref[o]) < TOL_SUM
    # swap one IR in the middle of a hop while small calls are running: both sides do the same at the same sample
    ref2, gpu2 = oracle.Convolver(2, 2, 0), H.Convolver(2, 2, 0)
    for c in (ref2, gpu2):
        for i in range(2):
            for o in range(2):
                assert c.set(i, o, irs[i][o], True) == 0
    cut = 8192 * 9 + 3000
    new_ir = oracle.synth_ir(7, 7, L)
    a_ref, a_gpu = ref2.run(xs[:, :cut], 2, 500), gpu2.run(xs[:, :cut], 2, 500)
    assert ref2.set(1, 0, new_ir, True) == 0 and gpu2.set(1, 0, new_ir, True) == 0
    b_ref, b_gpu = ref2.run(xs[:, cut:], 2, 500), gpu2.run(xs[:, cut:], 2, 500)
    for o in range(2):
        assert rel_err(a_gpu[o], a_ref[o]) < TOL_SUM
        # a re-set pair restarts from silence at the sample on both sides
        assert rel_err(b_gpu[o], b_ref[o]) < TOL_SUM


def test_process_argument_edge_cases(H, oracle):
    """numIns / numOuts smaller than constructed, calls longer than the engine's internal block, zero-length calls."""
    nin, nout, L, S = 3, 3, 4000, 90000
    irs = {(i, o): oracle.synth_ir(i, o, L) for i in range(nin) for o in range(nout)}
    xs = np.stack([oracle.synth_audio(i, S) for i in range(nin)])
    c = H.Convolver(nin, nout, 1)
    for (i, o), h in irs.items():
        assert c.set(i, o, h, True) == 0
    # one call far longer than the internal block (32768): chunked internally, same stream
    y = np.zeros((nout, S), np.float32)
    c.process(xs, y)
    for o in range(nout):
        assert rel_err(y[o], sum(truth_conv(xs[i], irs[(i, o)], 128) for i in range(nin))) < TOL_SUM
    # a zero-length call is a no-op
    c.process(xs[:, :0], np.zeros((nout, 0), np.float32))
    # fewer active inputs / outputs than constructed (Convolver.cpp:148-153, NToMonoConvolve.cpp:41)
    c.reset()
    y2 = np.full((nout, 5000), 7.0, np.float32)
    c.process(np.ascontiguousarray(xs[:, :5000]), y2, numIns=2, numOuts=2)
    for o in range(2):
        assert rel_err(y2[o], sum(truth_conv(xs[i, :5000], irs[(i, o)], 128) for i in range(2))) < TOL_SUM
    assert (y2[2] == 7.0).all()                                   # rows beyond numOuts are not written
    # the double overload clamps to the constructed sizes and converts through float (Convolver.cpp:156-183)
    c.reset()
    xd = xs[:, :4096].astype(np.float64)
    yd = np.zeros((nout, 4096), np.float64)
    c.process(xd, yd, numIns=9, numOuts=9)
    for o in range(nout):
        assert rel_err(yd[o], sum(truth_conv(xs[i, :4096], irs[(i, o)], 128) for i in range(nin))) < TOL_SUM


def test_silent_and_cleared_pairs(H, oracle):
    xs = np.stack([oracle.synth_audio(i, 4000) for i in range(2)])
    h = oracle.synth_ir(0, 0, 2000)
    c = H.Convolver(2, 2, 0)
    assert c.set(0, 1, h, True) == 0
    y = c.run(xs, 2, 512)
    assert (y[0] == 0).all()                                      # no IR on output 0: zeros (NToMonoConvolve.cpp:39)
    assert rel_err(y[1], truth_conv(xs[0], h)) < TOL
    c.clear(0, 1, False)
    assert (c.run(xs, 2, 512) == 0).all()
    assert c.set(0, 1, oracle.synth_ir(0, 0, 20000), False) == 4  # too long without resize: silent (MonoConvolve.cpp:139,183)
    assert (c.run(xs, 2, 512) == 0).all()


def test_reset_restarts_history(H, oracle):
    h, x = oracle.synth_ir(3, 3, 9000), oracle.synth_audio(8, 12000)
    c = H.Convolver(1, 1, 0)
    c.set(0, 0, h, True)
    y1 = c.run(x[None, :], 1, 512)[0]
    c.reset()
    y2 = c.run(x[None, :], 1, 512)[0]
    assert rel_err(y2, y1) < 1e-6                                 # same input after reset() -> same output (no tail of run 1)
    y3 = c.run(x[None, :], 1, 512)[0]
    assert rel_err(y3, y1) > 1e-3                                 # without reset the previous tail rings on


def test_capacity_growth_keeps_running_pairs(H, oracle):
    # growing the tail for one pair (set(..., resize=True)) must not disturb another pair that is mid-stream
    xs = np.stack([oracle.synth_audio(i, 40000) for i in range(2)])
    xs[1, :20000] = 0.0                                           # input 1 is silent until its IR arrives
    h0, h1 = oracle.synth_ir(0, 0, 16000), oracle.synth_ir(1, 0, 60000)
    c = H.Convolver(2, 1, 2)
    assert c.set(0, 0, h0, True) == 0
    y_a = c.run(xs[:, :20000], 1, 1000)[0]
    assert c.set(1, 0, h1, True) == 0                             # re-strides the tail stage while input 0 is running
    y_b = c.run(xs[:, 20000:], 1, 1000)[0]
    y = np.concatenate([y_a, y_b])
    truth = truth_conv(xs[0], h0, 512) + truth_conv(xs[1], h1, 512)
    assert rel_err(y, truth) < TOL_SUM


def test_set_while_processing_is_safe(H, oracle):
    """Threading contract: one thread streams process(), another loads IRs concurrently.  Nothing may
    crash or go non-finite, and once the control thread is done and the engine is reset the output is exact."""
    import threading
    nin, nout, L = 3, 2, 30000
    xs = np.stack([oracle.synth_audio(i, 60000) for i in range(nin)])
    irs = {(i, o): oracle.synth_ir(i, o, L) for i in range(nin) for o in range(nout)}
    c = H.Convolver(nin, nout, 0)
    for (i, o), h in irs.items():
        assert c.set(i, o, h, True) == 0
    stop = threading.Event()
    errors = []

    def control():
        k = 0
        try:
            while not stop.is_set():
                i, o = k % nin, (k // nin) % nout
                h = irs[(i, o)] if k % 3 else oracle.synth_ir(i + 7, o, 20000 + 1000 * (k % 5))
                rc = c.set(i, o, h, True)
                assert rc == 0
                if k % 4 == 0:
                    assert c.reset(i, o) == 0
                k += 1
        except Exception as e:  # pragma: no cover
            errors.append(e)

    th = threading.Thread(target=control)
    th.start()
    try:
        for _ in range(3):
            y = c.run(xs, nout, 512)
            assert np.isfinite(y).all()
    finally:
        stop.set()
        th.join()
    assert not errors, errors
    for (i, o), h in irs.items():
        assert c.set(i, o, h, True) == 0
    c.reset()
    y = c.run(xs, nout, 512)
    for o in range(nout):
        truth = sum(truth_conv(xs[i], irs[(i, o)]) for i in range(nin))
        assert rel_err(y[o], truth) < TOL_SUM


def test_per_pair_reset_mid_stream_is_exact(H, oracle):
    """Per-pair reset while other pairs run: like the reference (which restarts the pair's private history at the exact
    sample) the restarted pair forgets its input and drops its pending output at the sample — checked here against float64
    ground truth, and against the oracle over many more scenarios in test_pair_restart_gpu.py."""
    L, S, cut = 6000, 80000, 17000
    xs = np.stack([oracle.synth_audio(i, S) for i in range(2)])
    h0, h1 = oracle.synth_ir(0, 0, L), oracle.synth_ir(1, 0, L)
    c = H.Convolver(2, 1, 2)                                      # medium latency: stages 1024 / 4096 / 16384
    assert c.set(0, 0, h0, True) == 0 and c.set(1, 0, h1, True) == 0
    y_a = c.run(xs[:, :cut], 1, 500)[0]
    assert c.reset(1, 0) == 0                                     # restart pair (1,0) only
    y_b = c.run(xs[:, cut:], 1, 500)[0]
    y = np.concatenate([y_a, y_b])
    x1_after = np.concatenate([np.zeros(cut, np.float32), xs[1, cut:]])
    exact = truth_conv(xs[0], h0, 512) + np.concatenate([truth_conv(xs[1, :cut], h1, 512)[:cut], np.zeros(S - cut)]) \
        + truth_conv(x1_after, h1, 512)
    assert rel_err(y, exact) < TOL_SUM


def test_api_behaviour_checklist(H):
    from semantics import EXPECTED, checklist
    ns = types.SimpleNamespace(PartitionedConvolve=H.PartitionedConvolve, TimeDomainConvolve=H.TimeDomainConvolve, MonoConvolve=H.MonoConvolve,
                               NToMonoConvolve=H.NToMonoConvolve, Convolver=H.Convolver)
    obs, (y, ya) = checklist(ns)
    assert obs == EXPECTED
    assert np.allclose(ya, y + 1.0, atol=1e-6)


# ------------------------------------------------------------------------------------------- full-size shapes by property

def _sparse_matrix_case(H, nin, nout, L, S, block, latency, seed):
    """IRs that are a single scaled, delayed impulse: the exact answer is a gain-weighted sum of delayed inputs,
    which is cheap to compute in float64 at any size."""
    rng = np.random.RandomState(seed)
    delays = rng.randint(0, L, size=(nout, nin))
    gains = rng.uniform(-1, 1, size=(nout, nin))
    xs = rng.uniform(-1, 1, size=(nin, S)).astype(np.float32)
    c = H.Convolver(nin, nout, latency)
    h = np.zeros(L, np.float32)
    for o in range(nout):
        for i in range(nin):
            h[delays[o, i]] = gains[o, i]
            assert c.set(i, o, h, True) == 0
            h[delays[o, i]] = 0.0
    y = c.run(xs, nout, block)
    lat = {0: 0, 1: 128, 2: 512}[latency]
    for o in range(nout):
        t = np.zeros(S)
        for i in range(nin):
            d = delays[o, i] + lat
            if d < S:
                t[d:] += gains[o, i] * xs[i, : S - d].astype(np.float64)
        assert rel_err(y[o], t) < TOL_SUM, (o, rel_err(y[o], t))


def test_long_stream_wraps_every_ring_many_times(H, oracle):
    """Six million samples through a zero-latency MonoConvolve and a 2x2 Convolver in ragged calls: the input history,
    the input-spectrum rings and the stage timelines (a few blocks deep) wrap hundreds of times; the end of the stream
    must be as accurate as its start."""
    S = 6_000_000
    L = 5000
    h = oracle.synth_ir(2, 1, L)
    x = oracle.synth_audio(12, S)
    ref = oracle.MonoConvolve(L, 0)
    ref.setResetOffset(0)
    assert ref.set(h, True) == 0
    y_ref = ref.run(x, 2048)
    gpu = H.MonoConvolve(L, 0)
    assert gpu.set(h, True) == 0
    y = gpu.run(x, [64, 1000, 4096, 333, 8192, 5, 20000, 32768, 40001])
    for a, b in ((0, 100_000), (S // 2, S // 2 + 100_000), (S - 100_000, S)):
        assert rel_err(y[a:b], y_ref[a:b]) < TOL, (a, b)
    assert rel_err(y, y_ref) < TOL

    S2 = 1_500_000
    irs = [[oracle.synth_ir(i, o, 3000 + 500 * (i + o)) for o in range(2)] for i in range(2)]
    xs = np.stack([oracle.synth_audio(30 + i, S2) for i in range(2)])
    cref, cgpu = oracle.Convolver(2, 2, 0), H.Convolver(2, 2, 0)
    for i in range(2):
        for o in range(2):
            assert cref.set(i, o, irs[i][o], True) == 0 and cgpu.set(i, o, irs[i][o], True) == 0
    yr, yg = cref.run(xs, 2, 2048), cgpu.run(xs, 2, [4096, 100, 65536, 7])
    for o in range(2):
        assert rel_err(yg[o][-200_000:], yr[o][-200_000:]) < TOL_SUM


@pytest.mark.parametrize("ratio,latency_zero", [(8, True), (4, True), (8, False)])
def test_extended_tail_ladder_matches_reference_layout(H, oracle, ratio, latency_zero):
    """MI355X extension: past the reference's largest FFT the far tail is served by `ratio` times larger FFTs per
    rung (up to 2^20).  It must be the same convolution with the same latency as the reference partitioning."""
    nin, nout, L, S = 2, 2, 700000, 900000
    irs = {(i, o): oracle.synth_ir(i, o, L - 1000 * i - 10 * o) for i in range(nin) for o in range(nout)}
    xs = np.stack([oracle.synth_audio(i, S) for i in range(nin)])
    lat = 0 if latency_zero else 128
    c = H.Convolver(nin, nout, custom=(L, latency_zero, 256, 1024, 4096, 16384), tailRatio=ratio, maxBlock=32768)
    st = c.stage_stats()
    assert [s["fft_size"] for s in st][-1] == (1 << 20)           # the ladder reached the largest FFT
    for (i, o), h in irs.items():
        assert c.set(i, o, h, True) == 0
    y = c.run(xs, nout, [8192, 32768, 1000, 333])
    for o in range(nout):
        truth = sum(truth_conv(xs[i], irs[(i, o)], lat) for i in range(nin))
        assert rel_err(y[o], truth) < TOL_SUM
    # and the plain reference layout gives the same stream
    r = H.Convolver(nin, nout, custom=(L, latency_zero, 256, 1024, 4096, 16384), maxBlock=32768)
    for (i, o), h in irs.items():
        assert r.set(i, o, h, True) == 0
    y_r = r.run(xs[:, :200000], nout, 8192)
    for o in range(nout):
        assert rel_err(y[o][:200000], y_r[o]) < TOL_SUM


def test_config4_shape_64x64_2s_impulse_irs(H):
    _sparse_matrix_case(H, 64, 64, 96000, 3 * 8192 + 100, 8192, 0, seed=4)


def test_config3_shape_8to1_5s_impulse_irs(H):
    _sparse_matrix_case(H, 8, 1, 240000, 250000, 4096, 0, seed=3)


def test_config2_shape_partitioned_10s(H, oracle):
    # PartitionedConvolve 1x1, 4096-point partitions, 10 s @ 48 kHz IR (P = 235): linearity + impulse at full size
    L, N, S = 480000, 4096, 500000
    h = oracle.synth_ir(0, 0, L)
    p = H.PartitionedConvolve(N, L, 0, 0)
    assert p.set(h) == 0
    x = np.zeros(S, np.float32)
    x[0], x[1000] = 1.0, -0.5
    y = p.run(x, 8192)
    t = np.zeros(S)
    t[2048: 2048 + L] += h[: S - 2048]
    t[3048: 3048 + L] -= 0.5 * h[: S - 3048]
    assert np.abs(y - t).max() < 2e-6 * np.abs(h).max() * 10
    # linearity on noise: conv(a x1 + b x2) == a conv(x1) + b conv(x2)
    x1, x2 = oracle.synth_audio(1, 60000), oracle.synth_audio(2, 60000)
    outs = []
    for sig in (x1, x2, 0.5 * x1 - 2.0 * x2):
        p.reset()
        outs.append(p.run(sig.astype(np.float32), 4096).astype(np.float64))
    assert rel_err(outs[2], 0.5 * outs[0] - 2.0 * outs[1]) < TOL_SUM


def test_config5_shape_16x16_60s_device_resident(H):
    """16x16, 60 s @ 96 kHz IRs (11.8 GB of spectra): impulse IRs built in HBM, audio resident in HBM."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    nin = nout = 16
    L, S = 5760000, 4 * 8192
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(5)
    delays = rng.randint(0, 3 * 8192, size=(nout, nin))            # keep the responses inside the streamed span
    far = rng.randint(L - 100000, L, size=(nout, nin))             # plus one far tap per pair that must stay silent here
    gains = rng.uniform(-1, 1, size=(nout, nin))
    c = H.Convolver(nin, nout, 0, custom=(L, True, 256, 1024, 4096, 16384), maxBlock=8192)
    h = torch.zeros(L, dtype=torch.float32, device=dev)
    for o in range(nout):
        for i in range(nin):
            h[int(delays[o, i])] = float(gains[o, i])
            h[int(far[o, i])] = 1.0
            torch.cuda.synchronize()
            assert c.set_dev(i, o, h.data_ptr(), L, True) == 0
            h[int(delays[o, i])] = 0.0
            h[int(far[o, i])] = 0.0
    xs = torch.from_numpy(rng.uniform(-1, 1, size=(nin, S)).astype(np.float32)).to(dev)
    ys = torch.zeros((nout, S), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    for pos in range(0, S, 8192):
        c.process_dev(xs.data_ptr() + 4 * pos, S, ys.data_ptr() + 4 * pos, S, nin, nout, 8192)
    c.synchronize()
    y, x = ys.cpu().numpy(), xs.cpu().numpy()
    for o in range(nout):
        t = np.zeros(S)
        for i in range(nin):
            d = delays[o, i]
            t[d:] += gains[o, i] * x[i, : S - d].astype(np.float64)
        assert rel_err(y[o], t) < TOL_SUM
    st = c.stage_stats()
    assert st[-1]["fft_size"] == 16384 and st[-1]["partitions"] == 703


def test_create_destroy_cycles_do_not_leak(H, oracle):
    """300 engines created, loaded, run and destroyed (plus capacity growth and FFT-surface scratch): device memory in use
    must come back to where it started (within the allocator's granularity), and nothing may crash."""
    torch = pytest.importorskip("torch")
    import hisstools_library_amd.fft as F
    h, x = oracle.synth_ir(1, 1, 30000), oracle.synth_audio(1, 4096)
    xs = np.stack([x, x])

    def cycle(k):
        c = H.Convolver(2, 2, k % 3)
        for i in range(2):
            assert c.set(i, i, h[: 1000 + 97 * k], True) == 0
        c.run(xs, 2, 512)
        assert c.set(0, 1, h, True) == 0                         # grows the tail capacity
        c.run(xs, 2, 4096)
        del c
        m = H.MonoConvolve(5000, 0)
        m.set(h[:5000], True)
        m.run(x, 256)
        del m
        F.hisstools_fft(np.zeros(1 << 16, np.float32), np.zeros(1 << 16, np.float32), 16)

    for k in range(20):                                          # warm the caches (twiddles, FFT scratch, allocator pools)
        cycle(k)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for k in range(300):
        cycle(k)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, f"{(free0 - free1) >> 20} MiB of device memory not returned after 300 create/destroy cycles"


def test_process_rejects_buffers_the_c_side_would_misread(H, oracle):
    """Convolver.process hands raw row pointers to the C ABI: mixed precisions, integer data, transposed / strided views
    and over-long channel counts must be refused or clamped at the Python boundary, never reinterpreted."""
    c = H.Convolver(2, 2, 0)
    h = oracle.synth_ir(0, 0, 3000)
    for i in range(2):
        assert c.set(i, i, h, True) == 0
    x32 = np.stack([oracle.synth_audio(i, 1024) for i in range(2)])
    y32 = np.zeros((2, 1024), np.float32)
    with pytest.raises(TypeError):
        c.process(x32.astype(np.float64), y32)                       # f64 in, f32 out would overflow the f32 rows
    with pytest.raises(TypeError):
        c.process((x32 * 1000).astype(np.int32), y32)
    with pytest.raises(ValueError):
        c.process(np.asfortranarray(x32), y32)                        # samples of a row are not contiguous
    with pytest.raises(ValueError):
        c.process(x32[:, ::2], y32)
    with pytest.raises(ValueError):
        c.process(x32, np.zeros((2, 100), np.float32))
    c.process(x32, y32, numIns=7, numOuts=9)                          # clamped to the arrays' rows
    ref = oracle.Convolver(2, 2, 0)
    ref.setResetOffset(0)
    for i in range(2):
        ref.set(i, i, h, True)
    y_ref = ref.run(x32, 2, 1024)
    for o in range(2):
        assert rel_err(y32[o], y_ref[o]) < TOL
    # a strided ROW layout (rows far apart, samples contiguous) is fine: only row pointers cross the boundary
    big = np.zeros((4, 2048), np.float32)
    c.reset()
    c.process(x32, big[::2, :1024])
    for o in range(2):
        assert rel_err(big[2 * o, :1024], y_ref[o]) < TOL


def test_registered_host_memory_runs_in_place(H, oracle):
    """hcv_host_register: process() on evenly spaced rows inside registered memory runs on the caller's memory in place (no staging
    copies) and must give the staged path's stream — hop-sized calls (direct input / output of whole-hop mode), small and ragged
    calls, odd sample offsets (rows only 4-byte aligned), and a fallback to staging for rows outside the registration."""
    nin, nout, L, S = 3, 2, 20000, 6 * 8192
    irs = {(i, o): oracle.synth_ir(i, o, L) for i in range(nin) for o in range(nout)}
    xs = np.stack([oracle.synth_audio(i, S) for i in range(nin)])
    a, b = H.Convolver(nin, nout, 0, maxBlock=8192), H.Convolver(nin, nout, 0, maxBlock=8192)
    for c in (a, b):
        for (i, o), h in irs.items():
            assert c.set(i, o, h, True) == 0
    ya = a.run(xs, nout, [8192, 8192, 128, 1000, 333, 8192])             # staged
    xr, yr = xs.copy(), np.zeros((nout, S), np.float32)
    H.host_register(xr)
    H.host_register(yr)
    try:
        pos, k = 0, 0
        sizes = [8192, 8192, 128, 1000, 333, 8192]
        while pos < S:
            n = min(sizes[k % len(sizes)], S - pos)
            b.process(xr[:, pos:pos + n], yr[:, pos:pos + n])
            pos += n
            k += 1
        for o in range(nout):
            assert rel_err(yr[o], ya[o]) < TOL
        # rows outside the registered block: staged as before, same stream
        b.reset()
        other = np.zeros((nout, 4096), np.float32)
        b.process(np.ascontiguousarray(xs[:, :4096]), other)
        for o in range(nout):
            assert rel_err(other[o], ya[o][:4096]) < TOL
    finally:
        H.host_unregister(xr)
        H.host_unregister(yr)
    assert H.load().hcv_host_unregister(xr.ctypes.data) == -1             # not registered any more


_MUTE_SCRIPT = r"""
import sys, threading, time, json
import numpy as np
sys.path.insert(0, ".")
import hisstools_library_amd as H
from oracle import oracle as O
B, fs, L = 128, 48000, 20000
c = H.Convolver(2, 2, 0)
ha, hb = O.synth_ir(0, 0, L), O.synth_ir(3, 1, L)
for o in range(2):
    for i in range(2):
        assert c.set(i, o, ha, True) == 0
n_blocks = 300
x = np.zeros((2, n_blocks * B), np.float32)
x[0] = O.synth_audio(0, n_blocks * B)            # input 1 silent: output row 0 is pair (0, 0) alone
y = np.zeros((2, n_blocks * B), np.float32)
t_set = {}
def control():
    time.sleep(0.15)
    t_set["t0"] = time.perf_counter()
    assert c.set(0, 0, hb, True) == 0
    t_set["t1"] = time.perf_counter()
th = threading.Thread(target=control); th.start()
t_start = time.perf_counter(); stamps = []
for k in range(n_blocks):
    while time.perf_counter() < t_start + k * B / fs: pass
    stamps.append(time.perf_counter())
    c.process(x[:, k * B:(k + 1) * B], y[:, k * B:(k + 1) * B])
th.join()
peak = float(np.abs(y[0]).max())
quiet = [bool(np.abs(y[0, k * B:(k + 1) * B]).max() < 1e-5 * peak) for k in range(n_blocks)]
run = best = 0
for k in range(20, n_blocks):                    # (the stream's first blocks are quiet by themselves)
    run = run + 1 if quiet[k] else 0
    best = max(best, run)
print(json.dumps({"longest_quiet_run": best, "set_ms": 1e3 * (t_set["t1"] - t_set["t0"]), "peak": peak}))
"""


def test_reference_quirks_mode_mutes_the_pair_while_set_is_in_progress():
    """HCV_REFERENCE_QUIRKS=1, second quirk (round 6): the reference holds a pair's memory through the whole of set(), so the pair is SILENT for
    the blocks processed meanwhile and its pending output is dropped (MonoConvolve.cpp:118-140, 181-183); by default the pair here plays its
    previous IR until the swap.  A control thread replaces pair (0, 0) beside paced 128-sample calls with its hand-overs stalled 60 ms each
    (HCV_TEST_CTL_STALL_US): with the quirk the row fed by that pair alone goes exactly quiet for tens of blocks, without it it never does."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for quirks in ("1", "0"):
        env = dict(os.environ, HCV_TEST_CTL_STALL_US="60000")
        env.pop("HCV_REFERENCE_QUIRKS", None)
        if quirks == "1":
            env["HCV_REFERENCE_QUIRKS"] = "1"
        out = subprocess.run([sys.executable, "-c", _MUTE_SCRIPT], capture_output=True, text=True, timeout=300, cwd=root, env=env)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-2000:]
        res[quirks] = json.loads(out.stdout.strip().splitlines()[-1])
    print(res)
    assert res["1"]["longest_quiet_run"] >= 15, res          # >= 40 ms of silence on the replaced pair (the stall alone is 60 ms)
    assert res["0"]["longest_quiet_run"] <= 2, res           # ... and by default it keeps playing right through the set()
