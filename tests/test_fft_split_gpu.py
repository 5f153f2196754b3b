"""The residue-split hop transforms (hcv_fft_split.hip: one real transform over R/2 + 1 forward / R/2 inverse workgroups that share
nothing; replaces hisstools_rfft / hisstools_rifft per hop, HISSTools_FFT.cpp:226-248, for blocks of a few transforms).

By default they serve whole-hop blocks of at most 16 transforms of 16384 points (the 1 x 1 and 8 -> 1 engines: BASELINE configs 1
and 3), so the default suites already run them.  Here they are FORCED for every block of both transform sizes
(HCV_FFT_SPLIT=1 — read once per process, hence the child process) and the parity suites that drive whole-hop blocks are run
under it: oracle, golden vectors and float64 truth, tolerance as stated in those files (2e-6 / 1e-5 of the output peak).
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SUITE = ["tests/test_small_engine_pipeline_gpu.py", "tests/test_configs_dense_gpu.py::test_config3_dense_full_size_vs_oracle",
         "tests/test_gpu_parity.py::test_config2_shape_partitioned_10s", "tests/test_gpu_parity.py::test_config3_shape_8to1_5s_impulse_irs",
         "tests/test_pair_restart_gpu.py", "tests/test_steady_state_gpu.py::test_dense_irs_16x16_steady_state_vs_oracle"]


def test_parity_suites_with_split_transforms_forced():
    e = dict(os.environ, HCV_FFT_SPLIT="1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-k", "mono or config1 or config2 or config3 or pipelined or restart or dense or swap"]
                         + SUITE + ["tests/test_gpu_parity.py"], capture_output=True, text=True, timeout=1500, cwd=ROOT, env=e)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


def test_split_off_matches_split_on_bitwise_in_neither_direction_but_to_rounding():
    """the same 8 -> 1 stream with the split transforms (default) and without (HCV_FFT_SPLIT=0): two different factorizations of
    the same transforms, equal to rounding (<= 2e-6 of the peak), and both within tolerance of the float64 truth"""
    code = ("import numpy as np, hisstools_library_amd as H\n"
            "from oracle import oracle as O\n"
            "nin, L, S = 8, 100000, 20 * 8192\n"
            "xs = np.stack([O.synth_audio(i, S) for i in range(nin)])\n"
            "c = H.Convolver(nin, 1, 0, custom=(L, True, 256, 1024, 4096, 16384), maxBlock=8192)\n"
            "for i in range(nin): assert c.set(i, 0, O.synth_ir(i, 0, L), True) == 0\n"
            "np.save(__import__('sys').argv[1], c.run(xs, 1, 8192))\n")
    import numpy as np
    import tempfile
    ys = []
    with tempfile.TemporaryDirectory() as d:
        for k, mode in enumerate(("0", "1")):
            path = os.path.join(d, f"y{k}.npy")
            out = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, HCV_FFT_SPLIT=mode))
            assert out.returncode == 0, out.stderr[-2000:]
            ys.append(np.load(path))
    peak = np.abs(ys[0]).max()
    assert np.abs(ys[0] - ys[1]).max() <= 2e-6 * peak
    assert np.abs(ys[0] - ys[1]).max() > 0           # (they ARE different code paths)


@pytest.mark.parametrize("block", [8192, 6144, 4096])
def test_fused_multi_hop_block_of_a_4096_point_stage_vs_oracle(block):
    """BASELINE config 2's shape at reduced length — PartitionedConvolve(4096) called with 4, 3 and 2 hops per block — through the
    fused multi-hop block (hcv_fft_split.hip: fused_block_hops_kernel; forward transforms, multiply-accumulate over all partitions
    and inverse of the block's hops in ONE launch) against the CPU oracle (bit-identical to the unmodified reference,
    PartitionedConvolve.cpp:243-426) with a dense decaying-noise IR, streamed past the IR length so that every partition is live,
    <= 2e-6 of the peak; the stage statistics must show the fused launch (one slice, hop tile = hops per block), and the same
    stream with the fused block switched off (HCV_COOP=0: transforms, hop-tiled MAC, reduction, inverse as four launches)
    must agree to rounding."""
    code = ("import sys, json, numpy as np, hisstools_library_amd as H\n"
            "from oracle import oracle as O\n"
            "B = int(sys.argv[2]); L, N = 60000 - 77, 4096; S = 40 * B\n"
            "h = O.synth_ir(0, 0, L); x = O.synth_audio(3, S)\n"
            "p = H.Convolver(1, 1, 0, custom=(L, False, N, 0, 0, 0), maxBlock=B); assert p.set(0, 0, h, True) == 0\n"
            "y = p.run(x[None, :], 1, B)[0]\n"
            "r = O.PartitionedConvolve(N, L, 0, 0); r.setResetOffset(0); assert r.set(h) == 0\n"
            "yr = r.run(x, 2048)\n"
            "st = p.stage_stats()[-1]\n"
            "np.save(sys.argv[1], np.stack([y, yr]))\n"
            "print(json.dumps({k: int(st[k]) for k in ('hop_tile', 'ksplit', 'partitions') if k in st}))\n")
    import json
    import numpy as np
    import tempfile
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for mode in ("1", "0"):
            path = os.path.join(d, f"y{mode}.npy")
            out = subprocess.run([sys.executable, "-c", code, path, str(block)], capture_output=True, text=True, timeout=600, cwd=ROOT,
                                 env=dict(os.environ, HCV_COOP=mode))
            assert out.returncode == 0, out.stderr[-2000:]
            res[mode] = (np.load(path), json.loads(out.stdout.strip().splitlines()[-1]))
    (y1, st1), (y0, st0) = res["1"], res["0"]
    peak = np.abs(y1[1]).max()
    assert np.abs(y1[0] - y1[1]).max() <= 2e-6 * peak, np.abs(y1[0] - y1[1]).max() / peak
    assert np.abs(y0[0] - y0[1]).max() <= 2e-6 * peak
    assert np.abs(y1[0] - y0[0]).max() <= 2e-6 * peak
    if st1:
        assert st1["hop_tile"] == block // 2048 and st1["ksplit"] == 1, st1         # the fused launch ran the steady state ...
        assert st0["ksplit"] > 1, st0                                                # ... and the separate launches did without it


def test_fused_multi_hop_block_with_a_lead_slot_vs_float64_truth():
    """The same fused multi-hop block on a stage WITH a lead slot: a zero-latency 1 x 1 ladder that ends in a 4096-point stage
    (MonoConvolve(L, true, 256, 1024, 4096): MonoConvolve.cpp:203-258), called with 8192 samples — whole-hop mode, four hops of the last
    stage per block, partition 0 = the lead slot reading the block's own spectra.  Checked against a float64 FFT convolution (the
    reference has a defect for zero-latency ladders of fewer than four sizes, DESIGN section 4 deviation 6), <= 2e-6 of the peak, fused
    on and off."""
    code = ("import sys, json, numpy as np, hisstools_library_amd as H\n"
            "from oracle import oracle as O\n"
            "from scipy.signal import fftconvolve\n"
            "B = 8192; L = 50000 + 13; S = 36 * B\n"
            "h = O.synth_ir(1, 0, L); x = O.synth_audio(5, S)\n"
            "p = H.Convolver(1, 1, 0, custom=(L, True, 256, 1024, 4096, 0), maxBlock=B); assert p.set(0, 0, h, True) == 0\n"
            "y = p.run(x[None, :], 1, B)[0]\n"
            "t = fftconvolve(x.astype(np.float64), h.astype(np.float64))[:S]\n"
            "st = p.stage_stats()[-1]\n"
            "print(json.dumps({'err': float(np.abs(y - t).max() / np.abs(t).max()), 'hop_tile': int(st['hop_tile']), 'ksplit': int(st['ksplit']), 'fft': int(st['fft_size'])}))\n")
    import json
    for mode in ("1", "0"):
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, HCV_COOP=mode))
        assert out.returncode == 0, out.stderr[-2000:]
        r = json.loads(out.stdout.strip().splitlines()[-1])
        assert r["fft"] == 4096 and r["err"] <= 2e-6, r
        if mode == "1":
            assert r["hop_tile"] == 4 and r["ksplit"] == 1, r
