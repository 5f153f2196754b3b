"""Forward progress of the fused one-launch blocks under contention (hcv_fft_split.hip: fused_block_1x1_kernel,
fused_block_nx1_kernel, fused_block_hops_kernel — one hop of PartitionedConvolve::process, PartitionedConvolve.cpp:243-385, or of
NToMonoConvolve::process, NToMonoConvolve.cpp:35-43, as ONE launch whose workgroups hand data over inside it).

A workgroup that spins on a hand-over counter holds its CU, so such a launch is only safe if nobody can ever wait without bound for
a workgroup that is not yet resident.  The kernels guarantee that by construction — producers have the low block indices, every
wait is bounded, and a workgroup whose wait runs out does the missing (idempotent) work itself — and these tests put that under
the loads that could hang an unbounded wait: eight engines at once, each with its own host thread issuing back-to-back
asynchronous 8192-sample calls for two seconds (8 x 64 spinning consumers against 256 CUs), and the same under a CU mask that
leaves the process 32 CUs (one engine's 64 consumers alone outnumber them).  HCV_COOP_SPIN=0 makes every wait run out at once, so
the helping path itself is what computes: its results must be the same.  Every run is a child process under a hard timeout (a
hang must fail the test, not the box), every engine's output is compared with the oracle (<= 2e-6 of the peak, the float32 bound
of SURVEY 8c for short IRs) and every repetition must reproduce the first bit for bit.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 2e-6
CU_MASK_32 = "0:0-31"      # HSA_CU_MASK: GPU 0, compute units 0 .. 31


def _run(kind, engines, seconds, mask=None, extra_env=None):
    env = dict(os.environ)
    if mask:
        env["HSA_CU_MASK"] = mask
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_fused_contention_worker.py"), kind, str(engines), str(seconds)],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    print(r)
    return r


def _check(r):
    assert r["max_err"] <= TOL, r
    assert r["all_same"], r
    assert r["fused_launches"] > 0, r           # the fused launch is what ran, not the four-launch fallback
    assert r["reps"] >= 2, r


@pytest.mark.parametrize("kind", ["nx1", "1x1", "hops"])
def test_eight_concurrent_engines(kind):
    _check(_run(kind, 8, 2.0))


@pytest.mark.parametrize("kind", ["nx1", "1x1", "hops"])
def test_helping_path_computes_the_same(kind):
    """HCV_COOP_SPIN=0: no wait is ever granted, every consumer runs the producers' tasks itself (all of them redundantly, beside
    the producers): one engine, and eight at once"""
    _check(_run(kind, 1, 1.0, extra_env={"HCV_COOP_SPIN": "0"}))
    _check(_run(kind, 8, 1.0, extra_env={"HCV_COOP_SPIN": "0"}))


@pytest.mark.parametrize("kind,engines", [("nx1", 1), ("1x1", 1), ("hops", 1), ("nx1", 8), ("1x1", 8)])
def test_under_a_32_cu_mask(kind, engines):
    free = _run(kind, 1, 0.3)
    r = _run(kind, engines, 2.0, mask=CU_MASK_32)
    _check(r)
    # the mask was in force: the bandwidth probe of the masked child ran at least twice as long
    assert r["probe_ms"] >= 2.0 * free["probe_ms"], (r, free)
