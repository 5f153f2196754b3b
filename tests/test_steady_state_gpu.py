"""Full-depth, steady-state parity of the launches the headline roofline is quoted on — the tail stage of a matrix with >= 8 outputs once
EVERY partition of every pair is live (PartitionedConvolve.cpp:321-348 scheduling all P partitions, :387-426 the multiply-accumulate):
  * device-pointer one-hop blocks (what bench.py times; from round 6 on streamed engines too): the n x m fused block, mac_meet_kernel
    (hcv_fused_nxm.hip) — the same unchecked, nontemporal sum with its k-slices meeting in LDS;
  * the stage's own stream (host-pointer calls, a stage that stood down, HCV_SERIAL=0): spectral_mac_kernel<OT = 8, TT = 1, CHECK = false,
    NT = true> with split-K (hcv_mac.hip).
The ramp-up after a reset runs the CHECK = true, non-NT variant, so each test here streams past the whole IR length, puts energy into every
region of the IR (every partition index, every k-slice, the ring wrap at R = Pcap + 2 Tmax), and asserts through the stage statistics that
the steady-state launch actually ran — and which one.

Tolerance: max|y - y_ref| <= 1e-5 * max|y_ref| per channel for the long-IR / many-input shapes.
"""
import os
import subprocess
import sys

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


@pytest.fixture(scope="module")
def torch():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    return torch


def _sparse_device_case(H, torch, nin, nout, L, hops, taps_per_pair, seed, B=8192, spread=None, tail_ratio=0, call_hops=1):
    """Impulse IRs built in HBM (a few scaled taps per pair, spread over the WHOLE IR), audio resident in HBM, streamed in
    whole tail hops.  The exact answer is a gain-weighted sum of delayed inputs, computed in float64 (torch on the GPU: plain
    shifted adds, nothing of this library).  Returns the tail stage's statistics."""
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(seed)
    S = hops * B
    # every partition index of the tail gets a tap from some pair: the first tap of a pair walks the partitions in order
    # (pair k -> partition k mod P), the others are uniform over the IR
    P = -(-(L - B) // B)
    pairs = nin * nout
    delays = rng.randint(0, L, size=(nout, nin, taps_per_pair))
    walk = (np.arange(pairs) % P) * B + B + rng.randint(0, B, size=pairs)
    delays[:, :, 0] = np.minimum(walk, L - 1).reshape(nout, nin)
    if spread is not None:
        delays[:, :, -1] = rng.randint(spread[0], spread[1], size=(nout, nin))
    gains = rng.uniform(-1, 1, size=(nout, nin, taps_per_pair))
    c = H.Convolver(nin, nout, 0, custom=(L, True, 256, 1024, 4096, 16384), maxBlock=B * call_hops, tailRatio=tail_ratio)
    h = torch.zeros(L, dtype=torch.float32, device=dev)
    for o in range(nout):
        for i in range(nin):
            idx = torch.from_numpy(delays[o, i]).to(dev)
            # (two taps of a pair may coincide: accumulate, as the float64 sum below does)
            h.index_put_((idx,), torch.from_numpy(gains[o, i].astype(np.float32)).to(dev), accumulate=True)
            torch.cuda.synchronize()
            assert c.set_dev(i, o, h.data_ptr(), L, True) == 0
            h.index_fill_(0, idx, 0.0)
    xs = torch.from_numpy(rng.uniform(-1, 1, size=(nin, S)).astype(np.float32)).to(dev)
    ys = torch.zeros((nout, S), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    c.clear_stats()
    for pos in range(0, S, B * call_hops):
        c.process_dev(xs.data_ptr() + 4 * pos, S, ys.data_ptr() + 4 * pos, S, nin, nout, min(B * call_hops, S - pos))
    c.synchronize()
    x64 = xs.to(torch.float64)
    worst = 0.0
    for o in range(nout):
        t = torch.zeros(S, dtype=torch.float64, device=dev)
        for i in range(nin):
            for k in range(taps_per_pair):
                d = int(delays[o, i, k])
                if d < S:
                    t[d:] += float(np.float32(gains[o, i, k])) * x64[i, : S - d]
        peak = float(t.abs().max())
        err = float((ys[o].to(torch.float64) - t).abs().max()) / peak
        # the END of the stream (every partition live, steady-state kernel) on its own
        err_end = float((ys[o, -8 * B:].to(torch.float64) - t[-8 * B:]).abs().max()) / peak
        worst = max(worst, err, err_end)
        assert err < TOL_SUM and err_end < TOL_SUM, (o, err, err_end)
    st = c.stage_stats()
    if tail_ratio:
        return st, worst
    return st[-1], worst


def test_config5_full_depth_steady_state(H, torch):
    """c5 (16x16, 60 s @ 96 kHz, P = 703, ksplit 24): 712 hops, three taps per pair over the whole 5.76 M-sample IR."""
    L, hops = 5760000, 712
    tail, worst = _sparse_device_case(H, torch, 16, 16, L, hops, 3, seed=55, spread=(L - 3 * 8192, L))
    assert tail["fft_size"] == 16384 and tail["partitions"] == 703
    assert tail["out_tile"] == 8 and tail["hop_tile"] == 1 and (tail["ksplit"] > 1 or tail["fused_launches"] > 0), tail     # (HCV_SERIAL=1: the n x m block)
    assert tail["mac_launches"] == hops
    # every hop from the 703rd on runs the unchecked instantiation with nontemporal loads
    assert tail["mac_steady_launches"] >= hops - 704, tail
    # ... and which launch that is: the fused block by default, the split-K kernel on the stage's stream where a knob says so
    if os.environ.get("HCV_SERIAL") == "0" or os.environ.get("HCV_COOP") == "0" or os.environ.get("HCV_TAIL_HEAD") == "0":
        assert tail["fused_launches"] == 0 and tail["ksplit"] > 1, tail
    elif os.environ.get("HCV_SERIAL") is None:
        assert tail["fused_launches"] >= hops - 704 - 64 * tail["fused_stood_down"] and tail["fused_launches"] > 0, tail


def test_config5_full_depth_on_the_stage_stream():
    """The same case with the whole-hop blocks kept on the stage's own stream (HCV_SERIAL=0, read once per process: a child process) —
    spectral_mac_kernel<8, 1, false, true> with its 24 k-slices through memory, reduce / inverse behind it: what host-pointer calls, a
    stage that stood down and every round before the sixth run at this depth."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HCV_SERIAL="0")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_steady_state_gpu.py") + "::test_config5_full_depth_steady_state",
                          "-q", "-m", "gpu", "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=1200, cwd=root, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "1 passed" in out.stdout, out.stdout[-1000:]


def test_config5_extended_ladder_full_depth(H, torch):
    """The extended far-tail ladder (MI355X extension, tailRatio = 8: 16384 -> 131072 -> 2^20-point rungs past the reference's
    largest FFT) at config-5 scale — the shape bench.py's `extended_layout` leg times: 16x16, 5.76 M-sample IRs, three taps per
    pair over the WHOLE IR (every rung, every partition index of the 16384-point rung), 712 hops, so that the far rungs — their
    four-step transforms, their deferred slices, their first emission 2^19 samples in — all contribute; <= 1e-5 of the float64
    answer.  It must be the same convolution as the reference partitioning gives (same test above)."""
    L, hops = 5760000, 712
    stats, worst = _sparse_device_case(H, torch, 16, 16, L, hops, 3, seed=58, spread=(L - 3 * 8192, L), tail_ratio=8)
    assert [s["fft_size"] for s in stats] == [256, 1024, 4096, 16384, 131072, 1 << 20]
    assert stats[-1]["partitions"] == 10 and stats[-2]["partitions"] == 7 and stats[-3]["partitions"] == 7
    assert worst < TOL_SUM


@pytest.mark.parametrize("nin,nout,L,hops", [(16, 16, 8192 * 201 - 77, 232), (8, 24, 8192 * 130, 160), (16, 9, 8192 * 140 + 5, 168)])
def test_batched_calls_long_rows_full_depth(H, torch, nin, nout, L, hops):
    """65536-sample calls (8 tail hops per launch: the software-pipelined 4 x 8 tile, hcv_mac_tiled.hip) on LONG rows — 130 to 200
    partitions, the batched leg of bench.py runs 703 — against float64 truth: taps over the whole IR, streamed past its length so
    that the unchecked instantiation runs with every partition live, every k-slice and the ring wrap included; output counts that
    are not a multiple of the tile (9 = two full tiles + one clamped tile).  <= 1e-5 of the peak."""
    tail, worst = _sparse_device_case(H, torch, nin, nout, L, hops, 3, seed=91 + nout, spread=(L - 2 * 8192, L), call_hops=8)
    assert tail["fft_size"] == 16384 and tail["hop_tile"] == 8 and tail["out_tile"] == 4, tail
    assert tail["mac_steady_launches"] >= 2, tail
    assert worst < TOL_SUM


def test_config4_full_depth_steady_state(H, torch):
    """c4 (64x64, 2 s @ 48 kHz, P = 11, ksplit 6): 16 hops, two taps per pair over all 96000 samples."""
    tail, worst = _sparse_device_case(H, torch, 64, 64, 96000, 16, 2, seed=44)
    assert tail["fft_size"] == 16384 and tail["partitions"] == 11
    # (HCV_SERIAL=1 — tools/knob_matrix.sh — makes this engine a serial one, whose steady-state hop is the n x m block: one k-slice group
    # for 64 x 64, counted in fused_launches)
    assert tail["out_tile"] == 8 and tail["hop_tile"] == 1 and (tail["ksplit"] > 1 or tail["fused_launches"] > 0), tail
    assert tail["mac_steady_launches"] >= 4, tail


def test_north_star_shape_64x64_10s_steady_state(H, torch):
    """The north-star target shape (64x64, 10 s @ 48 kHz: P = 58, 15.7 GB of spectra): 64 hops, two taps per pair."""
    tail, worst = _sparse_device_case(H, torch, 64, 64, 480000, 64, 2, seed=64)
    assert tail["fft_size"] == 16384 and tail["partitions"] == 58
    assert tail["out_tile"] == 8 and tail["hop_tile"] == 1 and (tail["ksplit"] > 1 or tail["fused_launches"] > 0), tail     # (HCV_SERIAL=1: the n x m block)
    assert tail["mac_steady_launches"] >= 4, tail


def test_dense_irs_16x16_steady_state_vs_oracle(H, oracle, torch):
    """Dense (decaying-noise) IRs on a 16x16 matrix with a 40-partition tail: OT = 8 with split-K against the reference
    ARITHMETIC (the CPU oracle, bit-identical to the unmodified reference), not only against impulses.  Rows 0, 7, 8, 15 —
    first and last row of both output tiles — are compared over the whole stream, ramp-up and steady state."""
    nin = nout = 16
    B = 8192
    L = B + 40 * B - 1234
    hops = 56
    S = hops * B
    rows = [0, 7, 8, 15]
    xs = np.stack([oracle.synth_audio(i, S) for i in range(nin)])
    c = H.Convolver(nin, nout, 0, custom=(L, True, 256, 1024, 4096, 16384), maxBlock=B)
    ref = oracle.Convolver(nin, len(rows), 0)
    ref.setResetOffset(0)
    for o in range(nout):
        for i in range(nin):
            h = oracle.synth_ir(i, o, L - 1000 * (i % 3))
            assert c.set(i, o, h, True) == 0
            if o in rows:
                assert ref.set(i, rows.index(o), h, True) == 0
    c.clear_stats()
    y = c.run(xs, nout, B)
    y_ref, _ = ref.stream_timed(xs, len(rows), 2048)
    for k, o in enumerate(rows):
        assert rel_err(y[o], y_ref[k]) < TOL_SUM, (o, rel_err(y[o], y_ref[k]))
        assert rel_err(y[o][-8 * B:], y_ref[k][-8 * B:]) < TOL_SUM
    tail = c.stage_stats()[-1]
    assert tail["partitions"] == 40 and tail["out_tile"] == 8 and (tail["ksplit"] > 1 or tail["fused_launches"] > 0) and tail["hop_tile"] == 1
    assert tail["mac_steady_launches"] >= hops - 41, tail


@pytest.mark.parametrize("rows,cols", [([0, 7, 8, 15], [0, 1, 2, 3]), ([13], list(range(16)))], ids=["4rows_x_4inputs", "1row_x_all16inputs"])
def test_config5_depth_dense_irs_vs_oracle(H, oracle, torch, rows, cols):
    """Dense-IR parity with the reference ARITHMETIC at config 5's own depth (what bench.py's self-check does after its timed region, here
    in the suite): 16x16, L = 5 760 000 (P = 703 tail partitions), dense decaying-noise IRs on EVERY pair, inputs 0..3 carrying audio and
    the others silent, streamed in 8192-sample hops PAST the whole IR (712 hops: ramp-up with partition bounds, then the unchecked
    split-K instantiation with all 703 partitions live).  Rows 0, 7, 8, 15 — first and last row of both output tiles — against
    oracle.Convolver(4, 4) holding the same sixteen IRs; and row 13 fed by all sixteen inputs against oracle.Convolver(16, 1) (PartitionedConvolve.cpp:321-348 scheduling, :387-426 the multiply-accumulate)."""
    dev = torch.device("cuda:0")
    nin = nout = 16
    L, B, hops = 5760000, 8192, 712
    S = hops * B
    # (second case, round 6: ONE output row fed by ALL sixteen inputs carrying audio — every input's ring and forward transform in the sum, where
    # the first case leaves twelve inputs silent; the same sixteen pairs of oracle work)
    c = H.Convolver(nin, nout, 0, custom=(L, True, 256, 1024, 4096, 16384), maxBlock=B)
    ref = oracle.Convolver(len(cols), len(rows), 0)
    ref.setResetOffset(0)
    spare = []
    for o in rows:
        for i in cols:
            h = oracle.synth_ir(i, o, L)
            assert c.set(i, o, h, True) == 0 and ref.set(i, rows.index(o), h, True) == 0
            if len(spare) < 4:
                spare.append(torch.from_numpy(h).to(dev))
    # (every other pair dense too — their spectra are part of every launch's traffic: four of the IRs above, from HBM)
    k = 0
    for o in range(nout):
        for i in range(nin):
            if not (o in rows and i in cols):
                torch.cuda.synchronize()
                assert c.set_dev(i, o, spare[k % len(spare)].data_ptr(), L, True) == 0
                k += 1
    xs = np.zeros((nin, S), np.float32)
    for i in cols:
        xs[i] = oracle.synth_audio(i, S)
    xd, yd = torch.from_numpy(xs).to(dev), torch.zeros((nout, S), device=dev)
    torch.cuda.synchronize()
    c.clear_stats()
    for pos in range(0, S, B):
        c.process_dev(xd.data_ptr() + 4 * pos, S, yd.data_ptr() + 4 * pos, S, nin, nout, B)
    c.synchronize()
    y = yd[rows].cpu().numpy()
    y_ref, _ = ref.stream_timed(xs[cols], len(rows), 2048)
    for k, o in enumerate(rows):
        assert rel_err(y[k], y_ref[k]) < TOL_SUM, (o, rel_err(y[k], y_ref[k]))
        assert rel_err(y[k][-8 * B:], y_ref[k][-8 * B:]) < TOL_SUM, (o, rel_err(y[k][-8 * B:], y_ref[k][-8 * B:]))
    tail = c.stage_stats()[-1]
    assert tail["fft_size"] == 16384 and tail["partitions"] == 703 and tail["out_tile"] == 8 and (tail["ksplit"] > 1 or tail["fused_launches"] > 0) and tail["hop_tile"] == 1
    assert tail["mac_launches"] == hops and tail["mac_steady_launches"] >= hops - 704, tail


def test_batched_calls_steady_state_vs_hop_calls(H, torch):
    """Offline-style calls (65536 samples = 8 tail hops per call: the hop-tiled instantiations) must give the stream the
    hop-sized calls give, in the steady state too — 16x16 with a 40-partition tail (>= 32: hop tile 8), dense random IRs."""
    dev = torch.device("cuda:0")
    nin = nout = 16
    B, BB = 8192, 65536
    L = B + 40 * B
    S = 7 * BB
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    decay = torch.pow(torch.tensor(10.0, device=dev), -3.0 * torch.arange(L, device=dev, dtype=torch.float32) / L)
    convs = [H.Convolver(nin, nout, 0, custom=(L, True, 256, 1024, 4096, 16384), maxBlock=BB) for _ in range(2)]
    for o in range(nout):
        for i in range(nin):
            h = (torch.rand(L, generator=g, device=dev) * 2 - 1) * decay
            h = h / torch.linalg.vector_norm(h)
            torch.cuda.synchronize()
            for c in convs:
                assert c.set_dev(i, o, h.data_ptr(), L, True) == 0
    xs = torch.rand((nin, S), generator=g, device=dev) * 2 - 1
    ys = [torch.zeros((nout, S), device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    for c, y, blk in zip(convs, ys, (B, BB)):
        for pos in range(0, S, blk):
            c.process_dev(xs.data_ptr() + 4 * pos, S, y.data_ptr() + 4 * pos, S, nin, nout, blk)
        c.synchronize()
    a, b = ys[0].cpu().numpy(), ys[1].cpu().numpy()
    for o in range(nout):
        assert rel_err(b[o], a[o]) < TOL_SUM, (o, rel_err(b[o], a[o]))
    st = convs[1].stage_stats()[-1]
    assert st["hop_tile"] == 8, st
