"""One (in, out) pair restarted while the others keep running — `Convolver::set(in, out, …)` / `reset(in, out)` mid-stream,
the reference's live IR swap (MonoConvolve.cpp:139-150 -> PartitionedConvolve.cpp:262-292, TimeDomainConvolve.cpp:91-98).

The reference gives every pair private buffers, so the restarted pair forgets its input and drops its pending output AT THE
SAMPLE.  The engine shares input spectra per input and timelines per output and makes the restart exact with ghost spectra
and by retiring the pair's pending output (hcv_ghost.hip).  Every scenario drives the oracle and the HIP classes with the
same script and compares the WHOLE output, transient included, at the suite's tolerance for sums.  Needs a real MI355X.
"""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL_SUM = 1e-5


@pytest.fixture(scope="module")
def H():
    import hisstools_library_amd as H
    assert H.load().hcv_device_count() > 0, "no GPU visible: the HIP path cannot run (and there is no fallback)"
    return H


def drive(conv, xs, nout, script, default_block):
    """script: list of (sample position, callable(conv)) applied when the stream reaches that position; the stream is cut
    into calls at those positions and otherwise follows `default_block` (int or cyclic pattern)."""
    total = xs.shape[1]
    cuts = sorted({0, total} | {pos for pos, _ in script})
    out = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        for pos, fn in script:
            if pos == a:
                fn(conv)
        if b > a:
            out.append(conv.run(np.ascontiguousarray(xs[:, a:b]), nout, default_block))
    return np.concatenate(out, axis=1)


def both(H, oracle, make, xs, nout, script, block, tol=TOL_SUM):
    ref, gpu = make(oracle), make(H)
    y_ref = drive(ref, xs, nout, script, block)
    y_gpu = drive(gpu, xs, nout, script, block)
    peak = max(float(np.abs(y_ref).max()), 1e-12)
    for o in range(nout):
        err = float(np.abs(y_gpu[o].astype(np.float64) - y_ref[o].astype(np.float64)).max()) / peak
        assert err < tol, (o, err, int(np.abs(y_gpu[o] - y_ref[o]).argmax()))
    return y_ref, y_gpu


def loaded(ns, nin, nout, latency, irs):
    c = ns.Convolver(nin, nout, latency)
    for (i, o), h in irs.items():
        assert c.set(i, o, h, True) == 0
    return c


def test_reset_one_pair_is_exact_without_head(H, oracle):
    """medium latency (stages 1024 / 4096 / 16384, no FIR head): restart of (1,0) in the middle of every stage's hop"""
    L, S, cut = 6000, 80000, 17000
    xs = np.stack([oracle.synth_audio(i, S) for i in range(2)])
    irs = {(0, 0): oracle.synth_ir(0, 0, L), (1, 0): oracle.synth_ir(1, 0, L)}
    both(H, oracle, lambda ns: loaded(ns, 2, 1, 2, irs), xs, 1, [(cut, lambda c: c.reset(1, 0))], 500)


@pytest.mark.parametrize("block", [64, 500, [100, 37, 128, 1000, 64, 3, 511, 4096, 77]])
def test_ir_swap_mid_hop_small_calls(H, oracle, block):
    """zero latency, a tail of 25 partitions in deferred (time-spread) mode, one IR replaced at an odd sample"""
    L, S, cut = 200_000, 330_000, 8192 * 9 + 3001
    xs = np.stack([oracle.synth_audio(50 + i, S) for i in range(2)])
    irs = {(i, o): oracle.synth_ir(i, o, L) for i in range(2) for o in range(2)}
    new_ir = oracle.synth_ir(7, 7, L)
    both(H, oracle, lambda ns: loaded(ns, 2, 2, 0, irs), xs, 2, [(cut, lambda c: c.set(1, 0, new_ir, True))], block)


def test_ir_swap_with_whole_hop_calls(H, oracle):
    """calls of one tail hop (whole-hop mode: only the last stage runs); the swap lands on a hop boundary, the next on an odd
    sample after a ragged call, with whole-hop calls resuming afterwards"""
    L, S = 70_000, 8192 * 24
    xs = np.stack([oracle.synth_audio(20 + i, S) for i in range(2)])
    irs = {(i, o): oracle.synth_ir(i, o, L) for i in range(2) for o in range(2)}
    a, b = oracle.synth_ir(8, 1, L), oracle.synth_ir(9, 2, 50_000)
    script = [(8192 * 6, lambda c: c.set(0, 1, a, True)), (8192 * 12 + 777, lambda c: c.set(1, 1, b, True)), (8192 * 13, lambda c: None)]
    both(H, oracle, lambda ns: loaded(ns, 2, 2, 0, irs), xs, 2, script, 8192)


def test_restart_with_head_through_fft(H, oracle):
    """16 pairs: hop-aligned calls take the FIR head through the first stage's FFT; restarts at aligned and odd positions"""
    nin = nout = 4
    L, S = 9000, 60_000
    xs = np.stack([oracle.synth_audio(30 + i, S) for i in range(nin)])
    irs = {(i, o): oracle.synth_ir(i, o, L) for i in range(nin) for o in range(nout)}
    n1, n2 = oracle.synth_ir(11, 3, L), oracle.synth_ir(12, 4, 5000)
    script = [(128 * 100, lambda c: c.set(2, 1, n1, True)), (128 * 200 + 64, lambda c: c.reset(3, 3)), (128 * 201, lambda c: None),
              (128 * 300 + 5, lambda c: c.set(0, 0, n2, True))]
    both(H, oracle, lambda ns: loaded(ns, nin, nout, 0, irs), xs, nout, script, 128)


def test_overlapping_restarts(H, oracle):
    """several restarts inside one IR length: different pairs at different times, the same pair twice, two pairs of one input at
    the same time, a pair cleared and a pair reset without a new IR"""
    nin, nout, L, S = 3, 2, 40_000, 150_000
    xs = np.stack([oracle.synth_audio(40 + i, S) for i in range(nin)])
    irs = {(i, o): oracle.synth_ir(i, o, L) for i in range(nin) for o in range(nout)}
    v = [oracle.synth_ir(20 + k, k, L - 3000 * k) for k in range(5)]

    def two(c):
        assert c.set(1, 0, v[2], True) == 0 and c.set(1, 1, v[3], True) == 0

    script = [(20_011, lambda c: c.set(0, 0, v[0], True)), (26_500, lambda c: c.set(2, 1, v[1], True)), (31_000, two),
              (40_960, lambda c: c.set(0, 0, v[4], True)), (52_001, lambda c: c.reset(2, 0)), (70_000, lambda c: c.clear(1, 1, False)),
              (90_500, lambda c: c.set(1, 1, v[0], True))]
    both(H, oracle, lambda ns: loaded(ns, nin, nout, 0, irs), xs, nout, script, [256, 1000, 8192, 50, 3000])


def test_restart_in_parallel_mode(H, oracle):
    """three parallel channels (Convolver(numIO, latency)): one channel's IR replaced while the others run"""
    L, S = 30_000, 90_000
    xs = np.stack([oracle.synth_audio(60 + i, S) for i in range(3)])
    new_ir = oracle.synth_ir(33, 3, 25_000)

    def make(ns):
        c = ns.Convolver(3, None, 0)
        for k in range(3):
            assert c.set(k, k, oracle.synth_ir(k, k, L), True) == 0
        return c

    both(H, oracle, make, xs, 3, [(33_333, lambda c: c.set(1, 1, new_ir, True)), (50_000, lambda c: c.reset(2, 2))], 700)


def test_restart_then_full_reset_and_regrow(H, oracle):
    """a restart followed closely by a reset of everything, and by a set that grows the tail (resize): neither may leave ghost
    corrections or retired output behind"""
    L, S = 20_000, 120_000
    xs = np.stack([oracle.synth_audio(70 + i, S) for i in range(2)])
    irs = {(i, o): oracle.synth_ir(i, o, L) for i in range(2) for o in range(2)}
    long_ir = oracle.synth_ir(5, 5, 45_000)
    script = [(10_007, lambda c: c.set(0, 0, irs[(1, 1)], True)), (14_000, lambda c: c.reset()), (30_001, lambda c: c.set(1, 0, long_ir, True)),
              (31_000, lambda c: c.set(0, 1, long_ir, True))]
    both(H, oracle, lambda ns: loaded(ns, 2, 2, 0, irs), xs, 2, script, 333)


def test_many_restarts_soak(H, oracle):
    """300 live swaps / restarts / clears in one stream, several of them alive at any time (IR 12000 samples, an event every
    1000): ghost-spectrum blocks are pooled and re-used, expired entries pruned, tables rebuilt — the output stays exact and
    device memory stops growing once the pool has reached its working size."""
    import torch
    nin, nout, L, S = 3, 3, 12_000, 300_000
    rng = np.random.default_rng(9)
    xs = np.stack([oracle.synth_audio(80 + i, S) for i in range(nin)])
    irs = {(i, o): oracle.synth_ir(i, o, L) for i in range(nin) for o in range(nout)}
    script = []
    for k in range(1, 300):
        pos = 1000 * k + int(rng.integers(0, 999))
        i, o = int(rng.integers(0, nin)), int(rng.integers(0, nout))
        what = k % 7
        if what == 5:
            script.append((pos, lambda c, i=i, o=o: c.reset(i, o)))
        elif what == 6:
            script.append((pos, lambda c, i=i, o=o: c.clear(i, o, False)))
        else:
            h = oracle.synth_ir(int(rng.integers(0, 50)), o, int(rng.integers(2000, L)))
            script.append((pos, lambda c, i=i, o=o, h=h: c.set(i, o, h, True)))
    ref, gpu = loaded(oracle, nin, nout, 0, irs), loaded(H, nin, nout, 0, irs)
    y_ref = drive(ref, xs, nout, script, 1024)
    torch.cuda.synchronize()
    free = []
    cuts = sorted({0, S} | {pos for pos, _ in script})
    out = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        for pos, fn in script:
            if pos == a:
                fn(gpu)
        out.append(gpu.run(np.ascontiguousarray(xs[:, a:b]), nout, [128, 500, 64]))
        free.append(torch.cuda.mem_get_info()[0])
    y = np.concatenate(out, axis=1)
    peak = float(np.abs(y_ref).max())
    for o in range(nout):
        assert float(np.abs(y[o].astype(np.float64) - y_ref[o]).max()) / peak < TOL_SUM, o
    # the second half of the run allocates nothing new
    assert min(free[len(free) // 2:]) >= min(free[: len(free) // 2]) - (1 << 20), (min(free[: len(free) // 2]), min(free[len(free) // 2:]))


@pytest.mark.parametrize("block", [[8192, 32768, 1000, 333], 256])
def test_ir_swap_on_the_extended_ladder(H, oracle, block):
    """the MI355X far-tail ladder (FFTs of 131072 and 1048576 points through the four-step transforms): ghost spectra and the
    retiring of pending output go through the same big-FFT kernels; checked against float64 ground truth"""
    from scipy.signal import fftconvolve
    L, S, cut = 700_000, 900_000, 300_000 + 4321
    if block == 256:
        S, cut = 420_000, 200_000 + 77
    xs = np.stack([oracle.synth_audio(90 + i, S) for i in range(2)])
    h0, h1, h1n = oracle.synth_ir(0, 0, L), oracle.synth_ir(1, 0, L - 5000), oracle.synth_ir(6, 6, L - 100_000)
    c = H.Convolver(2, 1, custom=(L, True, 256, 1024, 4096, 16384), tailRatio=8, maxBlock=32768)
    assert c.set(0, 0, h0, True) == 0 and c.set(1, 0, h1, True) == 0
    y_a = c.run(np.ascontiguousarray(xs[:, :cut]), 1, block)[0]
    assert c.set(1, 0, h1n, True) == 0
    y_b = c.run(np.ascontiguousarray(xs[:, cut:]), 1, block)[0]
    y = np.concatenate([y_a, y_b])
    f = lambda x, h: fftconvolve(x.astype(np.float64), h.astype(np.float64))[:S]      # noqa: E731
    x1_after = xs[1].copy()
    x1_after[:cut] = 0
    before = f(xs[1], h1)
    before[cut:] = 0
    truth = f(xs[0], h0) + before + f(x1_after, h1n)
    assert float(np.abs(y - truth).max()) / float(np.abs(truth).max()) < TOL_SUM


def test_active_channel_counts_change_mid_stream(H, oracle):
    """process(ins, outs, numIns, numOuts) with counts that change between calls (Convolver.cpp:148-153): a pair that drops
    out is muted at the sample, a pair that comes back restarts from silence — never stale history or stale pending output.
    (The reference freezes an inactive pair's private state and resumes it later; the engine's semantics are the restart.)
    Checked against float64 ground truth built per activation interval."""
    from scipy.signal import fftconvolve
    nin, nout, L, B = 3, 3, 14_000, 1000
    plan = [(3, 3)] * 20 + [(2, 2)] * 9 + [(3, 3)] * 14 + [(1, 3)] * 7 + [(3, 1)] * 6 + [(3, 3)] * 20      # (numIns, numOuts) per call
    S = B * len(plan)
    xs = np.stack([oracle.synth_audio(130 + i, S) for i in range(nin)])
    irs = {(i, o): oracle.synth_ir(i, o, L) for i in range(nin) for o in range(nout)}
    c = loaded(H, nin, nout, 0, irs)
    y = np.zeros((nout, S), np.float32)
    for k, (ni, no) in enumerate(plan):
        blk = np.full((nout, B), 7.0, np.float32)
        c.process(np.ascontiguousarray(xs[:, k * B:(k + 1) * B]), blk, numIns=ni, numOuts=no)
        y[:no, k * B:(k + 1) * B] = blk[:no]
        assert np.all(blk[no:] == 7.0)                              # rows beyond numOuts are not written
    truth = np.zeros((nout, S))
    for (i, o), h in irs.items():
        active = [i < ni and o < no for ni, no in plan]
        k = 0
        while k < len(plan):
            if not active[k]:
                k += 1
                continue
            k1 = k
            while k1 < len(plan) and active[k1]:
                k1 += 1
            a, b = k * B, k1 * B
            x = xs[i].astype(np.float64).copy()
            x[:a] = 0
            truth[o, a:b] += fftconvolve(x, h.astype(np.float64))[a:b]
            k = k1
    peak = np.abs(truth).max()
    for o in range(nout):
        assert np.abs(y[o] - truth[o]).max() / peak < TOL_SUM, (o, int(np.abs(y[o] - truth[o]).argmax()))
