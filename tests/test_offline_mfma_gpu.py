"""Offline calls (B = S): one process() call spans 32 hops or more of the tail stage (PartitionedConvolve.cpp:298-299, 321-348 loops the
hops of any numSamples; SURVEY section 7 "streaming vs batched", 8d "report at B = S").  Such a call's steady-state multiply-accumulate is
the dense per-bin contraction on the f32 matrix cores (hcv_mac_mfma.hip: spectral_mac_mfma_kernel, tiles of 32 / 64 hops x 16 outputs x
16 bins) instead of the register tiles.  Checked here

  * against the reference ARITHMETIC (the CPU oracle, bit-identical to the unmodified reference) with dense IRs on a tail of 130
    partitions in 32-hop calls,
  * against float64 truth on long rows with output counts and call lengths that are not multiples of the tile,
  * against the same engine layout fed hop-sized calls (the register-tiled kernels), <= 1e-5 of the peak,

each asserting through the stage statistics that the matrix-core instantiation ran (hop_tile 32 / 64, out_tile 16).

Tolerance: max|y - y_ref| <= 1e-5 * max|y_ref| per channel for the long-IR / many-input shapes.
"""
import numpy as np
import pytest

from conftest import rel_err
from test_steady_state_gpu import _sparse_device_case

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


def test_offline_32hop_calls_dense_irs_vs_oracle(H, oracle, torch):
    """16x16, tail of 130 partitions, dense decaying-noise IRs on EVERY pair (inputs 0..3 carry audio), 32-hop calls streamed past the
    IR: the last two calls run the matrix-core kernel with every partition live.  Rows 0, 7, 8, 15 against oracle.Convolver(4, 4)."""
    dev = torch.device("cuda:0")
    nin = nout = 16
    B, call_hops = 8192, 32
    L = B + 130 * B - 4321
    hops = 6 * call_hops
    S = hops * B
    rows, cols = [0, 7, 8, 15], [0, 1, 2, 3]
    c = H.Convolver(nin, nout, 0, custom=(L, True, 256, 1024, 4096, 16384), maxBlock=B * call_hops)
    ref = oracle.Convolver(len(cols), len(rows), 0)
    ref.setResetOffset(0)
    spare = []
    for o in rows:
        for i in cols:
            h = oracle.synth_ir(i, o, L - 777 * (i % 3))
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
    xd, yd = torch.from_numpy(xs).to(dev), torch.zeros((nout, S), device=dev)
    torch.cuda.synchronize()
    c.clear_stats()
    for pos in range(0, S, B * call_hops):
        c.process_dev(xd.data_ptr() + 4 * pos, S, yd.data_ptr() + 4 * pos, S, nin, nout, B * call_hops)
    c.synchronize()
    y = yd[rows].cpu().numpy()
    y_ref, _ = ref.stream_timed(xs[cols], len(rows), 2048)
    for k, o in enumerate(rows):
        assert rel_err(y[k], y_ref[k]) < TOL_SUM, (o, rel_err(y[k], y_ref[k]))
        assert rel_err(y[k][-call_hops * B:], y_ref[k][-call_hops * B:]) < TOL_SUM, (o, "steady span")
    tail = c.stage_stats()[-1]
    assert tail["fft_size"] == 16384 and tail["partitions"] == 130, tail
    assert tail["hop_tile"] == 32 and tail["out_tile"] == 16, tail        # the last call's launch: the matrix-core kernel
    assert tail["mac_steady_launches"] >= 1, tail


@pytest.mark.parametrize("nin,nout,L,hops,call_hops", [(16, 16, 8192 * 150 - 77, 256, 64), (8, 24, 8192 * 131, 264, 64), (16, 9, 8192 * 140 + 5, 208, 48)])
def test_offline_calls_long_rows_vs_float64(H, torch, nin, nout, L, hops, call_hops):
    """64-hop calls (and 48-hop calls: a ragged two-tile launch) on rows of 130 - 150 partitions against float64 truth: taps over the whole
    IR, streamed past its length; 24 outputs = one full and one half-empty output tile, 9 = one clamped tile; 264 hops in 64-hop
    calls end in a ragged call of 8 hops (the register tiles) after three matrix-core calls."""
    tail, worst = _sparse_device_case(H, torch, nin, nout, L, hops, 3, seed=17 + nout, spread=(L - 2 * 8192, L), call_hops=call_hops)
    assert tail["fft_size"] == 16384, tail
    assert tail["mac_steady_launches"] >= 1, tail
    if hops % call_hops == 0:
        assert tail["hop_tile"] == 64 and tail["out_tile"] == 16, tail
    assert worst < TOL_SUM


@pytest.mark.parametrize("nin,nout,call_hops", [(16, 16, 64), (64, 24, 32)])
def test_offline_calls_equal_hop_calls(H, torch, nin, nout, call_hops):
    """The stream of offline calls must be the stream of hop-sized calls (<= 1e-5 of the peak), steady state included: dense random
    IRs, a 40-partition tail, 5 calls of 64 (32) hops against 320 (160) hop-sized calls on a second engine holding the same IRs."""
    dev = torch.device("cuda:0")
    B = 8192
    BB = B * call_hops
    L = B + 40 * B
    S = 5 * BB
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    decay = torch.pow(torch.tensor(10.0, device=dev), -3.0 * torch.arange(L, device=dev, dtype=torch.float32) / L)
    convs = [H.Convolver(nin, nout, 0, custom=(L, True, 256, 1024, 4096, 16384), maxBlock=mb) for mb in (B, BB)]
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
        c.clear_stats()
        for pos in range(0, S, blk):
            c.process_dev(xs.data_ptr() + 4 * pos, S, y.data_ptr() + 4 * pos, S, nin, nout, blk)
        c.synchronize()
    a, b = ys[0].cpu().numpy(), ys[1].cpu().numpy()
    for o in range(nout):
        assert rel_err(b[o], a[o]) < TOL_SUM, (o, rel_err(b[o], a[o]))
        assert rel_err(b[o][-BB:], a[o][-BB:]) < TOL_SUM, (o, "steady span")
    st = convs[1].stage_stats()[-1]
    assert st["hop_tile"] == call_hops and st["out_tile"] == 16, st
    assert st["mac_steady_launches"] >= 2, st


def test_matrix_core_kernel_against_the_register_tiles_over_random_shapes():
    """tools/micro/mac_mfma_fuzz.cpp, launch level: 300 random shapes the engine can hand the kernel — 16 .. 2048 bins, 1 .. 20 inputs, 2 .. 40
    outputs (ragged output tiles), 1 .. 60 partitions (k-slices ending inside an input, chunks of fewer than 16 partitions), 32 .. 150 hops
    (ragged hop tiles), random ring lengths and first hops — each against the register-tiled kernels of the same library on the same
    operands, one case in three a ramp-up (a uniform first hop inside the reach of the partitions): within 1e-5 of the peak — two f32 evaluation
    orders; a missing term would show at 1e-2."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "tests", "cpp", "build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "mac_mfma_fuzz")
    lib = os.path.join(root, "hisstools_library_amd")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(lib, "csrc"), os.path.join(root, "tools", "micro", "mac_mfma_fuzz.cpp"),
                           "-L", lib, "-lhisstools_amd", f"-Wl,-rpath,{lib}", "-o", exe])
    out = subprocess.run([exe, "300", "11"], capture_output=True, text=True, timeout=600)
    print(out.stdout)
    assert out.returncode == 0 and "0 mismatches" in out.stdout, out.stdout + out.stderr
