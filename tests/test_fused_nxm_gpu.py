"""The n x m fused block (hcv_fused_nxm.hip): one hop of the last stage of an engine with SEVERAL outputs — Convolver::process
(Convolver.cpp:138-154) over NToMonoConvolve::process (NToMonoConvolve.cpp:35-43) over PartitionedConvolve::process
(PartitionedConvolve.cpp:243-385) — as a forward launch on one stream and ONE multiply-accumulate + inverse launch on another that
meet through agent-scope arrival counters instead of events.  The shape it exists for is one rank's share of BASELINE config 4
strong-scaled over 8 GPUs: 64 inputs x 8 output rows, 2 s impulse responses; the 4 x 2 grid layout's share is
32 inputs x 16 rows.

Every case runs in a child process under a hard timeout (the launches spin-wait on each other: a hang must fail the test, not the
box) with DENSE decaying-noise impulse responses on every pair, against the CPU oracle (bit-identical to the unmodified reference
on the golden vectors).  Tolerance: 1e-5 of the channel's peak (many-input sums).  HCV_COOP_SPIN=0 makes every
in-launch wait run out at once, so the helping path — the multiply-accumulate workgroups doing the forward transforms themselves —
computes the block: the same bits.  HCV_COOP=0 takes the separate kernels: the same stream to rounding.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-5


def _run(nin, nout, L, hops, mode, engines=1, extra_env=None):
    env = dict(os.environ)
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_fused_nxm_worker.py"), str(nin), str(nout), str(L), str(hops), mode, str(engines)],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    print(r)
    return r


def test_64x8_dense_vs_oracle_host_calls():
    """one rank's share of config 4 over 8 GPUs at full size: 64 x 8, L = 96 000, 16 hops (past the IR length: all 12 partitions live)"""
    r = _run(64, 8, 96000, 16, "host")
    assert r["max_err"] < TOL and r["tail_err"] < TOL, r
    assert r["fused_launches"] >= 3, r                      # the steady-state blocks took the fused launch, not the four-launch chain
    off = _run(64, 8, 96000, 16, "host", extra_env={"HCV_COOP": "0"})
    assert off["fused_launches"] == 0 and off["max_err"] < TOL, off


def test_64x8_async_calls_repeat_bit_for_bit_and_helping_path_agrees():
    r = _run(64, 8, 96000, 20, "dev")
    assert r["max_err"] < TOL and r["all_same"] and r["fused_launches"] >= 5, r
    h = _run(64, 8, 96000, 20, "dev", extra_env={"HCV_COOP_SPIN": "0"})
    assert h["max_err"] < TOL and h["all_same"] and h["fused_launches"] >= 5, h
    assert h["sha"] == r["sha"], (h, r)                     # whoever runs a task writes the same values


def test_32x16_grid_share_and_ragged_outputs():
    """the 4 x 2 grid layout's share (two output tiles, two k-slice groups) and a matrix whose last output tile is ragged (8 + 3)"""
    r = _run(32, 16, 96000, 16, "dev")
    assert r["max_err"] < TOL and r["all_same"] and r["fused_launches"] >= 3, r
    g = _run(24, 11, 60000, 14, "dev")
    assert g["max_err"] < TOL and g["all_same"] and g["fused_launches"] >= 3, g


def test_mode_changes_and_a_live_swap():
    """hop-sized calls, a ragged stretch (the block scheduler leaves whole-hop mode and comes back: the forward stream is joined and
    lined up again), then a live IR swap of one pair between two fused blocks (control work on the main stream)"""
    r = _run(16, 8, 60000, 24, "mixed")
    assert r["max_err"] < TOL and r["fused_launches"] >= 4, r
    h = _run(16, 8, 60000, 24, "mixed", extra_env={"HCV_COOP_SPIN": "0"})
    assert h["max_err"] < TOL and h["fused_launches"] >= 4, h


def test_four_engines_at_once():
    """Four engines, a host thread each, eight busy streams over the device's four hardware queues: forward launches DO arrive late here
    now and then (behind another engine's packets).  A late launch costs time, not bits (the helping path) — but three of them within 64
    blocks and the stage takes the separate kernels for a while (fused_stood_down), whose sums agree to rounding only.  So: every stream
    within tolerance of the oracle always; the same bits from every engine and repetition wherever no stage stood down, and always
    with every wait forced out (no stand-down then: nothing is reported as late) — the bits of one undisturbed engine."""
    one = _run(16, 8, 48000, 16, "dev")
    assert one["max_err"] < TOL and one["all_same"] and one["stood_down"] == 0, one
    r = _run(16, 8, 48000, 16, "many", engines=4)
    assert r["max_err"] < TOL and r["fused_launches"] >= 3, r
    if r["stood_down"] == 0:
        assert r["all_same"] and r["sha"] == one["sha"], (r, one)
    h = _run(16, 8, 48000, 16, "many", engines=4, extra_env={"HCV_COOP_SPIN": "0"})
    assert h["max_err"] < TOL and h["all_same"] and h["stood_down"] == 0 and h["sha"] == one["sha"], (h, one)


def test_a_forward_stream_that_falls_blocks_behind_keeps_off_the_rings():
    """Nothing makes the main stream wait for the forward stream: a multiply-accumulate launch whose wait runs out does the transforms
    itself and goes on.  With the forward stream held back 0.3 ms in front of every launch (HCV_NXM_TEST_DELAY_US, a test aid: the
    stream stuck behind other engines' packets) and no wait granted at all, the main stream is a dozen blocks ahead when the forward
    launches arrive — late launches that would file an old hop over a newer one in the history ring and an old spectrum into a ring
    slot that has come round.  They must notice and keep off (fwd_publish_kernel: `progress`): the same bits as the undisturbed run."""
    r = _run(16, 8, 96000, 40, "dev")
    assert r["max_err"] < TOL and r["all_same"], r
    d = _run(16, 8, 96000, 40, "dev", extra_env={"HCV_COOP_SPIN": "0", "HCV_NXM_TEST_DELAY_US": "300"})
    assert d["max_err"] < TOL and d["all_same"] and d["sha"] == r["sha"] and d["fused_launches"] >= 20, (d, r)
    # (and with the ordinary bounded waits: three launches whose wait ran out and the stage stands the block down for the separate kernels —
    # hcv_stage_stats.fused_stood_down — so the repetitions agree to rounding, not bit for bit)
    s = _run(16, 8, 96000, 40, "dev", extra_env={"HCV_NXM_TEST_DELAY_US": "300"})
    assert s["max_err"] < TOL, (s, r)


def test_a_late_forward_launch_never_transforms_the_next_calls_upload():
    """Host-pointer calls stage every block in ONE buffer, rewritten by the next call's upload.  With every wait forced out (the helping path
    computes each block, the call returns) and the forward stream held back 0.3 ms, the forward launch of block k arrives while call k + 1 .. k + 3
    is uploading: it must neither redo a task a helper has done (fwd_publish_kernel: the task's mark, `progress`) nor find its input rewritten
    (Engine::input_behind_forward: the upload goes behind the pipe stream) — the oracle's stream, and the same bits as the undisturbed run
    (ADVICE r5, high)."""
    r = _run(16, 8, 96000, 28, "host")
    assert r["max_err"] < TOL and r["tail_err"] < TOL and r["fused_launches"] >= 10, r
    d = _run(16, 8, 96000, 28, "host", extra_env={"HCV_COOP_SPIN": "0", "HCV_NXM_TEST_DELAY_US": "300"})
    assert d["max_err"] < TOL and d["tail_err"] < TOL and d["fused_launches"] >= 10, d
    assert d["sha"] == r["sha"], (d, r)
