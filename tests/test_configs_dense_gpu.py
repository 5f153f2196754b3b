"""BASELINE configs 3 and 4 at FULL size with DENSE (decaying-noise) impulse responses against the reference arithmetic
(the CPU oracle, bit-identical to the unmodified reference on the golden vectors) — not only impulse IRs and float64 truth.

config 3: NToMonoConvolve 8 -> 1, 5 s IRs (NToMonoConvolve.cpp:35-43: out = sum_i Mono_i(ins[i]); MonoConvolve.cpp:179-201).
config 4: Convolver 64x64, 2 s IRs (Convolver.cpp:138-154); rows {0, 7, 8, 63} — first / last row of the first output tile of
          eight, first row of the second, last row of the matrix — so that the 64-input sum order of the OT = 8 / split-K launch is
          compared with the reference's sequential per-pair sums at config-4 size.

Tolerance: max|y - y_ref| <= 1e-5 * max|y_ref| per output channel for many-input sums.
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


def test_config3_dense_full_size_vs_oracle(H, oracle):
    """8 -> 1, L = 240 000 dense IRs, 300 000 samples: the stream runs past the IR length, so every one of the tail's 29
    partitions per input is live at the end; hop-sized calls (whole-hop mode, the bench's steps) and ragged calls."""
    nin, L, S = 8, 240000, 300000
    xs = np.stack([oracle.synth_audio(i, S) for i in range(nin)])
    irs = [oracle.synth_ir(i, 0, L) for i in range(nin)]
    ref = oracle.NToMonoConvolve(nin, L, 0)
    ref.setResetOffset(0)
    gpu = H.NToMonoConvolve(nin, L, 0)
    cnv = H.Convolver(nin, 1, 0, custom=(L, True, 256, 1024, 4096, 16384), maxBlock=8192)
    for i in range(nin):
        assert ref.set(i, irs[i], True) == 0 and gpu.set(i, irs[i], True) == 0 and cnv.set(i, 0, irs[i], True) == 0
    y_ref = ref.run(xs, 2048)
    y = gpu.run(xs, [4096, 333, 8192, 128, 20000])
    assert rel_err(y, y_ref) < TOL_SUM, rel_err(y, y_ref)
    assert rel_err(y[-40000:], y_ref[-40000:]) < TOL_SUM
    # the bench's shape of call: whole 8192-sample hops through the Convolver-level object (one uniform convolution per block)
    yc = cnv.run(xs[:, : 36 * 8192], 1, 8192)[0]
    assert rel_err(yc, y_ref[: 36 * 8192]) < TOL_SUM, rel_err(yc, y_ref[: 36 * 8192])
    tail = cnv.stage_stats()[-1]
    assert tail["fft_size"] == 16384 and tail["partitions"] == 29


def test_config4_dense_rows_vs_oracle(H, oracle):
    """64x64, L = 96 000 dense IRs on every pair; rows 0, 7, 8, 63 against oracle.Convolver(64, 4) holding the same IRs;
    16 hops of 8192 (past the IR length: P = 11 partitions all live, the unchecked split-K instantiation)."""
    nin = nout = 64
    L, B, hops = 96000, 8192, 16
    S = hops * B
    rows = [0, 7, 8, 63]
    xs = np.stack([oracle.synth_audio(i, S) for i in range(nin)])
    c = H.Convolver(nin, nout, 0, custom=(L, True, 256, 1024, 4096, 16384), maxBlock=B)
    ref = oracle.Convolver(nin, len(rows), 0)
    ref.setResetOffset(0)
    for o in range(nout):
        for i in range(nin):
            # (rows that are not compared still carry full-length dense IRs: their traffic is part of the launch)
            h = oracle.synth_ir(i, o, L - 37 * ((i + o) % 5))
            assert c.set(i, o, h, True) == 0
            if o in rows:
                assert ref.set(i, rows.index(o), h, True) == 0
    c.clear_stats()
    y = c.run(xs, nout, B)
    y_ref, _ = ref.stream_timed(xs, len(rows), 2048)
    for k, o in enumerate(rows):
        assert rel_err(y[o], y_ref[k]) < TOL_SUM, (o, rel_err(y[o], y_ref[k]))
        assert rel_err(y[o][-3 * B:], y_ref[k][-3 * B:]) < TOL_SUM
    tail = c.stage_stats()[-1]
    assert tail["partitions"] == 11 and tail["out_tile"] == 8 and (tail["ksplit"] > 1 or tail["fused_launches"] > 0) and tail["hop_tile"] == 1
    assert tail["mac_steady_launches"] >= 3, tail


def test_hop_sized_host_pointer_calls_on_a_streamed_engine_vs_oracle(H, oracle):
    """hcv_convolver_process_f32 (host pointers: what HISSTools::Convolver::process calls) in 8192-sample calls on an engine with more than a
    GB of live tail spectra (16 x 16, 70 partitions): from the steady state on, the part of each hop's multiply-accumulate that needs nothing of
    the new block — partitions >= 1 — is enqueued ahead of the staging copy and the upload (Engine::host_pre_mac; Convolver.cpp:138-154 over
    PartitionedConvolve.cpp:352-377), the lead slot's terms behind them.  Dense decaying-noise IRs on every pair (inputs 0..3 carry audio), rows
    0, 7, 8, 15 against oracle.Convolver(4, 4) over the whole stream: ramp-up (whole launches) and steady state (split launches) alike."""
    import torch
    dev = torch.device("cuda:0")
    nin = nout = 16
    B = 8192
    L = B + 70 * B - 999
    hops = 96
    S = hops * B
    rows, cols = [0, 7, 8, 15], [0, 1, 2, 3]
    c = H.Convolver(nin, nout, 0, custom=(L, True, 256, 1024, 4096, 16384), maxBlock=B)
    ref = oracle.Convolver(len(cols), len(rows), 0)
    ref.setResetOffset(0)
    spare = []
    for o in rows:
        for i in cols:
            h = oracle.synth_ir(i, o, L - 555 * (i % 3))
            assert c.set(i, o, h, True) == 0 and ref.set(i, rows.index(o), h, True) == 0
            if len(spare) < 4:
                spare.append(torch.from_numpy(np.ascontiguousarray(h)).to(dev))
    k = 0
    for o in range(nout):
        for i in range(nin):
            if not (o in rows and i in cols):
                torch.cuda.synchronize()
                assert c.set_dev(i, o, spare[k % len(spare)].data_ptr(), spare[k % len(spare)].numel(), True) == 0
                k += 1
    xs = np.zeros((nin, S), np.float32)
    for i in cols:
        xs[i] = oracle.synth_audio(i, S)
    c.clear_stats()
    y = c.run(xs, nout, B)                                  # host pointers, 8192 samples per call
    y_ref, _ = ref.stream_timed(xs[cols], len(rows), 2048)
    for k, o in enumerate(rows):
        assert rel_err(y[o], y_ref[k]) < 1e-5, (o, rel_err(y[o], y_ref[k]))
        assert rel_err(y[o][-8 * B:], y_ref[k][-8 * B:]) < 1e-5, (o, "steady span")
    tail = c.stage_stats()[-1]
    assert tail["partitions"] == 70, tail
    import os
    off = os.environ.get("HCV_HOST_PRE_MAC") == "0" or os.environ.get("HCV_SERIAL") == "1" or os.environ.get("HCV_TAIL_HEAD") == "0"
    if not off:                                              # (tools/knob_matrix.sh: the result above holds under every knob, the schedule not)
        assert tail["mac_launches"] == hops, tail
        assert tail["host_pre_launches"] >= hops - 75, tail  # every steady-state hop took the split form


def test_live_swap_on_a_streamed_engine_between_fused_blocks_vs_oracle(H, oracle):
    """A streamed engine (16 x 16, 71 partitions: 1.2 GB of live tail spectra) driven through DEVICE pointers one hop per call: its
    steady-state blocks are the n x m fused block on the main stream (round 6), its ramp-up blocks the checked kernels on the stage's stream.
    One pair's impulse response is replaced in mid-stream (Convolver::set beside process, MonoConvolve.cpp:118-140): the engine leaves the
    fused block for the pair's ramp-up (partition bounds: the checked kernels, back on the stage's stream) and returns to it — forward
    stream joined and lined up again both ways.  Rows 0, 7, 8, 15 against the oracle driven through the same calls, over the whole stream."""
    import os
    import torch
    dev = torch.device("cuda:0")
    nin = nout = 16
    B = 8192
    L = B + 70 * B - 999
    hops_a, hops_b = 80, 82
    S = (hops_a + hops_b) * B
    rows, cols = [0, 7, 8, 15], [0, 1, 2, 3]
    c = H.Convolver(nin, nout, 0, custom=(L, True, 256, 1024, 4096, 16384), maxBlock=B)
    ref = oracle.Convolver(len(cols), len(rows), 0)
    ref.setResetOffset(0)
    spare = []
    for o in rows:
        for i in cols:
            h = oracle.synth_ir(i, o, L - 555 * (i % 3))
            assert c.set(i, o, h, True) == 0 and ref.set(i, rows.index(o), h, True) == 0
            if len(spare) < 4:
                spare.append(torch.from_numpy(np.ascontiguousarray(h)).to(dev))
    k = 0
    for o in range(nout):
        for i in range(nin):
            if not (o in rows and i in cols):
                torch.cuda.synchronize()
                assert c.set_dev(i, o, spare[k % len(spare)].data_ptr(), spare[k % len(spare)].numel(), True) == 0
                k += 1
    xs = np.zeros((nin, S), np.float32)
    for i in cols:
        xs[i] = oracle.synth_audio(i, S)
    xd = torch.from_numpy(xs).to(dev)
    yd = torch.zeros((nout, S), device=dev)
    torch.cuda.synchronize()
    c.clear_stats()
    new_ir = oracle.synth_ir(41, 42, L - 123)
    y_ref = []
    for seg, (h0, h1) in enumerate(((0, hops_a), (hops_a, hops_a + hops_b))):
        if seg == 1:
            assert c.set(1, 7, new_ir, True) == 0 and ref.set(1, rows.index(7), new_ir, True) == 0
        for hop in range(h0, h1):
            c.process_dev(xd.data_ptr() + 4 * hop * B, S, yd.data_ptr() + 4 * hop * B, S, nin, nout, B, sync=False)
        c.synchronize()
        y_ref.append(ref.run(xs[cols][:, h0 * B:h1 * B], len(rows), 2048))
    y = yd.cpu().numpy()
    y_ref = np.concatenate(y_ref, axis=1)
    for k, o in enumerate(rows):
        assert rel_err(y[o], y_ref[k]) < TOL_SUM, (o, rel_err(y[o], y_ref[k]))
        assert rel_err(y[o][hops_a * B:], y_ref[k][hops_a * B:]) < TOL_SUM, (o, "after the swap")
    tail = c.stage_stats()[-1]
    assert tail["partitions"] == 70 and tail["mac_launches"] == hops_a + hops_b, tail
    if all(os.environ.get(v) is None for v in ("HCV_SERIAL", "HCV_COOP", "HCV_TAIL_HEAD", "HCV_EXACT_RESTART")):
        # both steady stretches took the fused block (9 + 11 blocks), the two ramp-ups did not
        assert 12 <= tail["fused_launches"] <= 24, tail
