"""The extended far-tail ladder reached by UNCHANGED callers (hcv_api.hip: Matrix::relayout_for, HCV_TAIL_RATIO and the automatic
rule) — the reference-shaped constructors, HISSTools::Convolver(numIns, numOuts, latency) (Convolver.h:23-50), know nothing of
impulse-response lengths; the partition ladder of MonoConvolve::setPartitions (MonoConvolve.cpp:203-258) is continued past its
16384-point stage (131072, 1048576 points: PartitionedConvolve.h:18-19 allows FFTs up to 2^20) when the first impulse response is
loaded into the still empty object.  Same convolution, same latency: every case is checked against the CPU oracle (bit-identical to
the unmodified reference) or a float64 truth, <= 1e-5 of the output peak (SURVEY 8c's bound for long IRs).

Whole-hop blocks of such an engine run on the PIVOT stage (the reference's own 16384-point tail) with the rungs keeping their
deferred schedule beside it (hcv_engine_block.hip); BASELINE's long-tail shapes (config 5, the 64 x 64 / 10 s shape, config 3) are
streamed at full size with the ladder on, in hop-sized, ragged and real-time-sized calls, with live IR swaps.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-5


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


def _child(code, env, timeout=900):
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=dict(os.environ, **env))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


STREAM = ("import json, numpy as np, hisstools_library_amd as H\n"
          "from oracle import oracle as O\n"
          "nin, nout, L, S, cut = 2, 2, 600000, 1300000, 700000 + 4321\n"
          "xs = np.stack([O.synth_audio(40 + i, S) for i in range(nin)])\n"
          "c = H.Convolver(nin, nout, 0, maxBlock=32768)          # the reference's constructor: no length, no layout\n"
          "r = O.Convolver(nin, nout, 0); r.setResetOffset(0)\n"
          "for i in range(nin):\n"
          "    for o in range(nout):\n"
          "        h = O.synth_ir(i, o, L - 30000 * i - 777 * o)\n"
          "        assert c.set(i, o, h, True) == 0 and r.set(i, o, h, True) == 0\n"
          "stages = [s['fft_size'] for s in c.stage_stats()]\n"
          "ya = c.run(np.ascontiguousarray(xs[:, :cut]), nout, BLOCKS)\n"
          "ra = r.run(np.ascontiguousarray(xs[:, :cut]), nout, 2048)\n"
          "hn = O.synth_ir(9, 9, L - 100000)\n"
          "assert c.set(1, 0, hn, True) == 0 and r.set(1, 0, hn, True) == 0     # a live swap: the other pairs keep running\n"
          "yb = c.run(np.ascontiguousarray(xs[:, cut:]), nout, BLOCKS)\n"
          "rb = r.run(np.ascontiguousarray(xs[:, cut:]), nout, 2048)\n"
          "y, ref = np.concatenate([ya, yb], axis=1), np.concatenate([ra, rb], axis=1)\n"
          "err = max(float(np.abs(y[o].astype(np.float64) - ref[o]).max() / np.abs(ref[o]).max()) for o in range(nout))\n"
          "print(json.dumps({'stages': stages, 'err': err, 'stages_after': [s['fft_size'] for s in c.stage_stats()],\n"
          "                  'fused': sum(int(s['fused_launches']) for s in c.stage_stats())}))\n")


@pytest.mark.parametrize("blocks", ["[8192, 32768, 1000, 333, 16384]", "8192", "128"])
def test_env_tail_ratio_for_the_reference_constructor(blocks):
    """HCV_TAIL_RATIO=8 and HISSTools::Convolver(2, 2, zero latency): the first set() of a 600 000-sample IR lays both rungs, the
    stream — whole-hop, ragged and 128-sample calls, with a live IR swap 700 000 samples in — equals the oracle's"""
    r = _child(STREAM.replace("BLOCKS", blocks), {"HCV_TAIL_RATIO": "8"})
    assert r["stages"] == [256, 1024, 4096, 16384, 131072, 1 << 20] and r["stages_after"] == r["stages"], r
    assert r["err"] <= TOL, r
    # without the variable this small matrix keeps the reference's own partitioning (the rule asks for >= 1 GiB of tail spectra)
    r0 = _child(STREAM.replace("BLOCKS", "8192").replace("S, cut = 2, 2, 600000, 1300000, 700000 + 4321", "S, cut = 2, 2, 600000, 200000, 100000"), {})
    assert r0["stages"] == [256, 1024, 4096, 16384] and r0["err"] <= TOL, r0


def test_the_pivot_on_the_n_x_m_block_when_asked_for():
    """HCV_NXM_LADDER=1 (off by default, hcv_engine_block.hip: as a process's second engine it lost): the pivot stage's hop as the
    n x m block writing into the stage's timeline, the rungs beside it — the same stream"""
    r = _child(STREAM.replace("BLOCKS", "8192"), {"HCV_TAIL_RATIO": "8", "HCV_NXM_LADDER": "1"})
    assert r["stages"] == [256, 1024, 4096, 16384, 131072, 1 << 20] and r["err"] <= TOL and r["fused"] >= 8, r


def test_busy_streams_are_placed_on_hardware_queues_at_creation():
    """hcv_queue_probe.hip: the pivot's two lanes and the first rung on hardware queues of their own, the last rung on the main
    stream's (profiles/r05_queue_probe.txt) — by what the experiment has found about the process's pooled streams (each stream is
    classified once in its life, and never beside a running stream); the stream stays the oracle's.  A process that has made other
    streams before (here: eleven of them, held) gets the same placement."""
    import re
    pre = ("import torch\n"
           "held = [torch.cuda.Stream() for _ in range(11)]\n"
           "for s_ in held:\n"
           "    with torch.cuda.stream(s_):\n"
           "        torch.zeros(8, device='cuda').add_(1)\n"
           "torch.cuda.synchronize()\n")
    for code in (STREAM.replace("BLOCKS", "8192"), pre + STREAM.replace("BLOCKS", "8192")):
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=ROOT,
                             env=dict(os.environ, HCV_TAIL_RATIO="8", HCV_VERBOSE="1"))
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
        r = json.loads(out.stdout.strip().splitlines()[-1])
        assert r["err"] <= TOL, r
        probes = [l for l in out.stderr.splitlines() if "queue probe" in l]
        assert probes, out.stderr[-2000:]
        m = re.search(r"(\d+) queues known, .* main stream on queue (\d+), (\d+) of (\d+) busy streams .* classes:((?: -?\d+)+)", probes[-1])
        assert m, probes
        queues, main_q, served, want = int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4))
        classes = [int(c) for c in m.group(5).split()]
        assert want == 4 and served == want, probes                # two lanes, two rungs
        assert classes[-1] == main_q and len(set(classes)) == 4 and queues >= 4, probes


def test_the_automatic_rule_on_the_baseline_shapes(H, oracle):
    """unset HCV_TAIL_RATIO: the ladder where the reference's tail would be HBM-bound (>= 32 partitions and >= 1 GiB of tail
    spectra: config 5 and the 64 x 64 / 10 s shape), the reference's own stage list elsewhere (config 4: 11 partitions; config 3:
    cache resident, one launch per block as it is).  The stage list is laid when the first IR arrives."""
    if os.environ.get("HCV_TAIL_RATIO"):
        pytest.skip("HCV_TAIL_RATIO is set")
    # (32 x 32 with 700 000-sample IRs: the 2^20-point rung is within reach of the IR, but its hop boundary would stream 4.3 GB inside one
    #  process call — the rule stops at the 131072-point rung)
    for (nin, nout, L, want) in ((16, 16, 5760000, [256, 1024, 4096, 16384, 131072, 1 << 20]), (64, 64, 480000, [256, 1024, 4096, 16384, 131072]),
                                 (32, 32, 700000, [256, 1024, 4096, 16384, 131072]),
                                 (64, 64, 96000, [256, 1024, 4096, 16384]), (8, 1, 240000, [256, 1024, 4096, 16384])):
        c = H.Convolver(nin, nout, 0)
        assert [s["fft_size"] for s in c.stage_stats()] == [256, 1024, 4096, 16384]
        assert c.set(0, 0, oracle.synth_ir(0, 0, L), True) == 0
        assert [s["fft_size"] for s in c.stage_stats()] == want, (nin, nout, L)
        # once a pair is loaded the stage list stays: a longer IR extends the last stage, as the reference extends its tail
        assert c.set(nin - 1, 0, oracle.synth_ir(1, 0, L + 300000), True) == 0
        assert [s["fft_size"] for s in c.stage_stats()] == want
        del c


def test_relayout_of_an_empty_object_under_a_running_stream(H, oracle):
    """a host that starts its audio callback BEFORE it loads impulse responses: the empty object streams silence, the first set()
    replaces its engine (ladder laid for the IR) while calls keep coming from another thread — never blocked, never an error —
    and from the set on the output is the oracle's"""
    import threading
    import time
    code_env = os.environ.get("HCV_TAIL_RATIO")
    if code_env not in (None, "8"):
        pytest.skip("needs HCV_TAIL_RATIO unset or 8")
    nin, nout, L, B = 16, 16, 700000, 512
    c = H.Convolver(nin, nout, 0)
    stop, calls, errors = threading.Event(), [0], []
    x = np.zeros((nin, B), np.float32)
    y = np.full((nout, B), 3.0, np.float32)

    def audio():
        try:
            while not stop.is_set():
                c.process(x, y)
                calls[0] += 1
                assert not y.any()           # silence: nothing loaded, or loaded and fed zeros
        except Exception as e:               # noqa: BLE001
            errors.append(e)

    th = threading.Thread(target=audio)
    th.start()
    time.sleep(0.2)
    before = calls[0]
    h = oracle.synth_ir(3, 3, L)
    assert c.set(0, 0, h, True) == 0         # 16 x 16 x 86 partitions x 64 KiB = 1.4 GiB: the rule lays the ladder
    time.sleep(0.2)
    stop.set()
    th.join()
    assert not errors, errors
    assert before > 50 and calls[0] > before
    assert [s["fft_size"] for s in c.stage_stats()] == [256, 1024, 4096, 16384, 131072, 1 << 20]
    # the pair restarts at its set(): stream it against the oracle
    S = 900000
    xs = np.zeros((nin, S), np.float32)
    xs[0] = oracle.synth_audio(5, S)
    c.reset()
    yg = c.run(xs, nout, [8192, 700, 16384])[0]
    r = oracle.Convolver(1, 1, 0)
    r.setResetOffset(0)
    assert r.set(0, 0, h, True) == 0
    yr = r.run(xs[:1], 1, 2048)[0]
    assert rel_err(yg, yr) <= TOL


@pytest.mark.parametrize("nin,nout,L,hops,taps,stages,ratio", [
    (64, 64, 480000, 72, 2, [256, 1024, 4096, 16384, 131072], 8),              # the 64 x 64 / 10 s @ 48 kHz shape north_star names
    (64, 64, 480000, 72, 2, [256, 1024, 4096, 16384, 32768, 65536, 131072], 2),    # ... on three rungs of ratio 2 (0.616 ms per step against 0.690)
    (64, 64, 480000, 72, 2, [256, 1024, 4096, 16384, 65536, 262144], 4),       # ... and as bench.py's extended leg asks for it (1010 - 1026 Msamples/s)
    (8, 1, 240000, 48, 3, [256, 1024, 4096, 16384, 131072], 8),                # BASELINE config 3
])
def test_baseline_long_tail_shapes_on_the_ladder(H, torch, nin, nout, L, hops, taps, stages, ratio):
    """full-size long-tail shapes with the ladder on (config 5 is test_steady_state_gpu.py::test_config5_extended_ladder_full_depth),
    sparse taps over the WHOLE impulse response against the exact float64 answer, hop-sized calls: the pivot stage's whole-hop
    convolution plus the rungs' deferred schedule"""
    from test_steady_state_gpu import _sparse_device_case
    stats, worst = _sparse_device_case(H, torch, nin, nout, L, hops, taps, seed=77 + nin, spread=(L - 2 * 8192, L), tail_ratio=ratio)
    assert [s["fft_size"] for s in stats] == stages
    assert worst < TOL


@pytest.mark.parametrize("block", [32, 128])
def test_real_time_calls_on_the_ladder_vs_oracle(H, oracle, block):
    """32- and 128-sample calls (deferred slices of the 2^20-point rung in flight between its boundaries) of a 3 x 2 ladder engine,
    streamed past the first emission of the far rung, against the oracle"""
    nin, nout, L = 3, 2, 560000
    S = 1 << 20 if block == 128 else 600000
    S += 20000
    xs = np.stack([oracle.synth_audio(60 + i, S) for i in range(nin)])
    c = H.Convolver(nin, nout, 0, custom=(L, True, 256, 1024, 4096, 16384), tailRatio=8, maxBlock=8192)
    r = oracle.Convolver(nin, nout, 0)
    r.setResetOffset(0)
    for i in range(nin):
        for o in range(nout):
            h = oracle.synth_ir(i, o, L - 1000 * i)
            assert c.set(i, o, h, True) == 0 and r.set(i, o, h, True) == 0
    y, yr = c.run(xs, nout, block), r.run(xs, nout, 2048)
    for o in range(nout):
        assert rel_err(y[o], yr[o]) <= TOL
