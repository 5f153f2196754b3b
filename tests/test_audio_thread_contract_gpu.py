"""The audio-thread contract of the reference (MemorySwap::attempt, MemorySwap.h:182-185; MonoConvolve.cpp:118-140,181-183;
ThreadLocks.hpp:51-87): `process` never waits for a control thread's `set` / `resize` — not for the IR upload, not for an allocation,
not for the device, and not for a lock: the engine's host state has an OWNER (hcv_engine.h: Engine::mOwner), the audio thread makes
one compare-exchange per call, and while a stream is running control calls POST their swap section — retiring kernels, the copies of
the staged spectra, the restart's fence and ghost spectra — for the audio thread to run between two of its blocks (`mailbox_runs`,
`mailbox_ns_max`).  The reference mutes the pair being replaced for the blocks processed meanwhile; here the pair keeps playing its
previous IR until the swap.  A control thread that loses its core holds nothing the audio thread needs (tests/cpp/audio_contract.cpp
is the same scenario from C++, SCHED_FIFO audio thread, with the control thread stalled on purpose).

A control thread loops set(resize=True) with GROWING 10 s-class IRs on a 16x16 zero-latency engine — every growth re-strides the
tail stage's whole spectrum store — while the audio thread issues paced 128-sample calls.  Asserted AT ANY LOAD, engine-exact: no
call ever found the engine owned by a control thread (`start_collisions == 0`), every set()'s section ran on the audio thread, no
control thread owned the engine while the stream ran; the outputs whose pairs are NOT being replaced equal the CPU oracle's sample
for sample (tolerance 1e-5) right through the swaps and regrows; the replaced pairs stay finite; and after the control thread has
finished, a known IR set + reset gives the oracle's stream again.  The wall-clock side (Python threads: the GIL and the host's
scheduler are in it) is asserted on a quiet host only (load average up to 2: a time-sharing thread has no claim on a CPU inside a
millisecond on a host that runs anything else) — p99 below three quarters of the budget, at most one call in two hundred over it, none
near a stall — and printed otherwise; the C++ programme carries the strict form (SCHED_FIFO: no call over budget).
"""
import os
import threading
import time

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import hisstools_library_amd as H
    assert H.load().hcv_device_count() > 0, "no GPU visible: the HIP path cannot run (and there is no fallback)"
    return H


def _paced(call, ncalls, period):
    ts = np.zeros(ncalls)
    t_start = time.perf_counter()
    for k in range(ncalls):
        while time.perf_counter() < t_start + k * period:
            pass
        t0 = time.perf_counter()
        call(k)
        ts[k] = time.perf_counter() - t0
    return ts * 1e3


@pytest.mark.parametrize("entry", ["host_pointers", "device_pointers", "sharded_host_pointers"])
def test_process_never_waits_for_set_or_regrow(H, oracle, entry):
    _scenario(H, oracle, entry)


def test_process_never_waits_for_set_or_regrow_at_32_samples_per_call(H, oracle):
    """The same contract at the smallest block size hosts use: 32-sample calls (0.67 ms budget), 4200 of them (2.8 s of audio), beside
    ~700 set(resize) calls.  The same criteria."""
    _scenario(H, oracle, "device_pointers", RB=32, ncalls=4200)


def _engine_criteria(rt, sets):
    """Exact counters of the engine: the same at any load"""
    assert rt["start_collisions"] == 0, rt
    assert rt["ctl_sections"] == 0, rt                                  # no control thread owned the engine while the stream ran
    assert rt["arena_misses"] == 0, rt                                  # every regrow came out of the arena reserved for it
    assert rt["mailbox_runs"] >= sets["n"] - 2, (rt, sets["n"])         # every set()'s section (a regrow's two more) ran on the audio thread


def _wall_clock_criteria(ts, budget):
    """p99 inside three quarters of the budget, at most one call in two hundred over it, none that looks like a stall behind an upload
    or a regrow (19 ms and more: what a regrow's memory mapping cost before the control arena)"""
    over = int((ts > budget).sum())
    assert over <= len(ts) // 200 and ts.max() < 19.0, f"{over} process calls over the {budget:.2f} ms budget beside set(), worst {ts.max():.3f} ms"
    assert np.percentile(ts, 99) < 0.75 * budget, f"p99 {np.percentile(ts, 99):.3f} ms"


def _scenario(H, oracle, entry, RB=128, ncalls=1400):
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    nin = nout = 16
    fs = 48000
    steady = [0, 1]                                    # output rows checked against the oracle (their pairs are never replaced)
    L_fix = 60000                                      # (ncalls = 1400 calls of 128 samples: 3.7 s of audio)
    S = ncalls * RB
    xs = np.stack([oracle.synth_audio(i, S) for i in range(nin)])
    # The object is created for 1.25 s impulse responses and the control thread grows them eightfold beside the running stream: a host
    # that does that reserves the control path's memory first (include/hisstools_amd.h: hcv_ctl_reserve), or the regrows have the driver
    # map a gigabyte under the audio thread — 19 - 49 ms in which every HIP call of the process stalls.
    H.ctl_reserve(0, 4 << 30)
    try:
        _scenario_body(H, oracle, entry, RB, ncalls, torch, dev, nin, nout, fs, steady, L_fix, S, xs)
    finally:
        H.ctl_reserve(0, 0)


def _scenario_body(H, oracle, entry, RB, ncalls, torch, dev, nin, nout, fs, steady, L_fix, S, xs):
    # ("sharded": the same through ONE object driving two engines on the GPU — rows 0..7 on the first, 8..15, the ones being
    #  replaced, on the second: control calls and process calls meet per shard)
    c = H.Convolver(nin, nout, 0, maxBlock=8192, devices=[0, 0]) if entry.startswith("sharded") else H.Convolver(nin, nout, 0, maxBlock=8192)
    ref = oracle.Convolver(nin, len(steady), 0)
    ref.setResetOffset(0)
    for o in range(nout):
        for i in range(nin):
            h = oracle.synth_ir(i, o, L_fix)
            assert c.set(i, o, h, True) == 0
            if o in steady:
                assert ref.set(i, steady.index(o), h, True) == 0
    # growing IRs for the control thread: 1.25 s .. 10 s, host memory (pageable: the upload is part of what must not block)
    grow = [oracle.synth_ir(3, 9, n) for n in range(60000, 480001, 60000)]

    ys = np.zeros((nout, S), np.float32)
    if entry == "device_pointers":
        xd, yd = torch.from_numpy(xs).to(dev), torch.zeros((nout, S), device=dev)
        torch.cuda.synchronize()

        def call(k):
            c.process_dev(xd.data_ptr() + 4 * k * RB, S, yd.data_ptr() + 4 * k * RB, S, nin, nout, RB, sync=True)
    else:
        def call(k):
            c.process(xs[:, k * RB:(k + 1) * RB], ys[:, k * RB:(k + 1) * RB])

    for k in range(40):                                # settle clocks and allocator pools before the control thread starts
        call(k)
    c.reset()
    c.clear_stats()
    stop = threading.Event()
    sets = {"n": 0, "worst_ms": 0.0, "errors": []}

    def control():
        k = 0
        while not stop.is_set():
            h = grow[k % len(grow)]
            o = 8 + (k % 8)                            # rows 8..15 only: the checked rows keep their IRs
            t0 = time.perf_counter()
            rc = c.set((5 * k) % nin, o, h, True)
            sets["worst_ms"] = max(sets["worst_ms"], 1e3 * (time.perf_counter() - t0))
            if rc != 0:
                sets["errors"].append(rc)
            sets["n"] += 1
            k += 1
            time.sleep(0.002)

    th = threading.Thread(target=control)
    th.start()
    try:
        ts = _paced(call, ncalls, RB / fs)
        # (the counters as the stream ends: a set() still in flight then waits the streaming window out and serves itself, rightly)
        rt = c.rt_stats()
    finally:
        stop.set()
        th.join()
    if entry == "device_pointers":
        ys = yd.cpu().numpy()
    budget = 1e3 * RB / fs
    print(f"[{entry}] over budget: {int((ts > 1e3 * RB / fs).sum())} of {ncalls} calls, slowest five {np.sort(ts)[-5:].round(3).tolist()}")
    print(f"[{entry}] {sets['n']} set(resize) calls (worst {sets['worst_ms']:.1f} ms each) beside {ncalls} paced calls: p50 {np.percentile(ts, 50):.3f} "
          f"p99 {np.percentile(ts, 99):.3f} max {ts.max():.3f} ms (budget {budget:.2f}); start collisions {rt['start_collisions']}, sections run by the "
          f"audio thread {rt['mailbox_runs']} (longest {rt['mailbox_ns_max'] / 1e3:.1f} us, mean "
          f"{rt['mailbox_ns_total'] / 1e3 / max(1, rt['mailbox_runs']):.1f} us), by control threads {rt['ctl_sections']}; arena misses {rt['arena_misses']}")
    assert not sets["errors"] and sets["n"] >= 8                         # every length was loaded at least once: the stage regrew
    _engine_criteria(rt, sets)
    load = os.getloadavg()[0]
    if os.environ.get("SAN_RUN") or load > 2.0:
        # (an instrumented library — tools/sanitize/run.sh — or a shared host under load: a pre-empted Python audio thread is one slow
        # call with every counter of the engine clean; the wall clock is printed above and not asserted)
        print(f"[{entry}] wall-clock criteria not asserted (load average {load:.1f}{', sanitizer run' if os.environ.get('SAN_RUN') else ''})")
    else:
        _wall_clock_criteria(ts, budget)
    assert np.isfinite(ys).all()
    y_ref = ref.run(xs, len(steady), 2048)
    for k, o in enumerate(steady):
        assert rel_err(ys[o], y_ref[k]) < 1e-5, (o, rel_err(ys[o], y_ref[k]))
    # afterwards: known IRs everywhere + reset -> the oracle's stream again (nothing stale survived the swaps and regrows)
    rows = [0, 9, 15]
    ref2 = oracle.Convolver(nin, len(rows), 0)
    ref2.setResetOffset(0)
    for o in range(8, nout):
        for i in range(nin):
            assert c.set(i, o, oracle.synth_ir(i, o, 30000), True) == 0
    for k, o in enumerate(rows):
        for i in range(nin):
            assert ref2.set(i, k, oracle.synth_ir(i, o, L_fix if o < 8 else 30000), True) == 0
    c.reset()
    y2 = c.run(xs[:, :40000], nout, [128, 4096, 1000])
    y2_ref = ref2.run(xs[:, :40000], len(rows), 2048)
    for k, o in enumerate(rows):
        assert rel_err(y2[o], y2_ref[k]) < 1e-5


_COLLISION_SCRIPT = r"""
import sys, threading, time, json
import numpy as np
sys.path.insert(0, ".")
import hisstools_library_amd as H
from oracle import oracle as O
c = H.Convolver(2, 2, 0)
ha, hb = O.synth_ir(0, 0, 9000), O.synth_ir(1, 1, 9000)
for o in range(2):
    for i in range(2):
        assert c.set(i, o, ha, True) == 0
x = np.stack([O.synth_audio(i, 8192) for i in range(2)])
res = {}
# no stream is running: a control thread serves itself, and is held inside its section for 150 ms (HCV_TEST_CTL_STALL_US).  The FIRST call of a
# stream arrives meanwhile: 128 samples — an audio callback: silence at once, no wait; then (another pause, another section) 4096 samples — an
# offline loop's chunk: it waits the section out and delivers
for name, n in (("callback", 128), ("chunk", 4096)):
    time.sleep(0.55)
    c.clear_stats()
    th = threading.Thread(target=lambda: c.set(0, 0, hb if name == "callback" else ha, True))
    th.start()
    time.sleep(0.04)
    y = np.full((2, n), 7.0, np.float32)
    t0 = time.perf_counter()
    c.process(x[:, :n], y)
    dt = 1e3 * (time.perf_counter() - t0)
    th.join()
    rt = c.rt_stats()
    res[name] = {"ms": dt, "peak": float(np.abs(y).max()), "start_collisions": rt["start_collisions"], "start_waits": rt["start_waits"], "ctl_sections": rt["ctl_sections"]}
# and the call after a collision proceeds
y = np.zeros((2, 128), np.float32)
c.process(x[:, :128], y)
res["after"] = {"peak": float(np.abs(y).max())}
print(json.dumps(res))
"""


def test_the_first_call_of_a_stream_meeting_a_control_section():
    """The one case in which a process call cannot have the engine (hcv_engine.h: start_collisions): no stream running, a control thread inside a
    section — held there 150 ms by HCV_TEST_CTL_STALL_US — and the stream's first call arrives.  A 128-sample call (an audio callback) returns at
    once with a silent block, waits for nothing, and the next call plays; a 4096-sample call (an offline loop's chunk, not a callback) waits the
    section out instead of delivering 4096 samples of silence (start_waits)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", _COLLISION_SCRIPT], capture_output=True, text=True, timeout=300, cwd=root,
                         env=dict(os.environ, HCV_TEST_CTL_STALL_US="150000"))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    print(r)
    assert r["callback"]["start_collisions"] == 1 and r["callback"]["start_waits"] == 0 and r["callback"]["peak"] == 0.0, r
    assert r["callback"]["ms"] < 20.0, r                           # it did not wait for the 150 ms section
    assert r["chunk"]["start_collisions"] == 0 and r["chunk"]["start_waits"] == 1 and r["chunk"]["peak"] > 0.0, r
    assert 50.0 < r["chunk"]["ms"] < 400.0, r                      # it waited the section out
    assert r["after"]["peak"] > 0.0, r
