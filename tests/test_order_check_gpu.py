"""HCV_ORDER_CHECK=1 (hcv_order_check.h): the stream / event order DESIGN.md section 2 documents, asserted while the engine enqueues —
vector clocks over the engine's streams and events, every launch's buffer accesses (input-spectrum ring slots, partial spectra, stage
timelines, the history ring) declared next to it, a read required to come after every earlier write of what it reads and a write after
every earlier access.  The scenarios are the block kinds of that section: the n x m block with asynchronous calls (forward launch on
its own stream, back-pressure every few blocks), an extended ladder on two lanes with rungs, ragged and small calls across the modes
with a live swap, the two-stream pipeline of a small engine.  None may report a violation; and with HCV_ORDER_CHECK_DROP=k the checker
overlooks every k-th wait (the device does not): it must then report some, or it is not looking.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TAIL = ("\nimport json as _j, hisstools_library_amd as _H\n"
        "print(_j.dumps({'violations': int(_H.load().hcv_order_check_violations())}))\n")

WORKER = ("import sys, runpy\n"
          "sys.argv = ['_fused_nxm_worker.py'] + ARGS\n"
          "try:\n"
          "    runpy.run_path('tests/_fused_nxm_worker.py', run_name='__main__')\n"
          "except SystemExit:\n"
          "    pass\n")

PIPE = ("import numpy as np, torch, hisstools_library_amd as H\n"
        "from oracle import oracle as O\n"
        "dev = torch.device('cuda:0')\n"
        "c = H.Convolver(NIN, 1, 0, maxBlock=8192)\n"
        "for i in range(NIN):\n"
        "    assert c.set(i, 0, O.synth_ir(i, 0, 120000), True) == 0\n"
        "x = torch.randn(NIN, 8192 * 40, device=dev)\n"
        "y = torch.zeros(1, 8192 * 40, device=dev)\n"
        "for k in range(40):                                     # asynchronous hop-sized calls: the small engine's two-stream pipeline / fused block\n"
        "    c.process_dev(x.data_ptr() + 4 * 8192 * k, x.shape[1], y.data_ptr() + 4 * 8192 * k, y.shape[1], NIN, 1, 8192)\n"
        "c.synchronize()\n"
        "for k in range(300):                                    # then real-time sizes: deferred slices, late chains\n"
        "    c.process_dev(x.data_ptr() + 4 * 128 * k, x.shape[1], y.data_ptr() + 4 * 128 * k, y.shape[1], NIN, 1, 128)\n"
        "c.synchronize()\n"
        "assert bool(torch.isfinite(y).all())\n")


def _run(code, env):
    out = subprocess.run([sys.executable, "-c", code + TAIL], capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, **env))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    v = json.loads(out.stdout.strip().splitlines()[-1])["violations"]
    return v, out.stderr


def _ladder_code():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_tail_ladder_gpu import STREAM
    return STREAM.replace("BLOCKS", "[8192, 32768, 1000, 333, 16384, 128, 128, 8192]")


SCENARIOS = {
    "nxm_async_64x8": (WORKER.replace("ARGS", "['64', '8', '96000', '24', 'dev', '1']"), {}),
    "nxm_mixed_modes_live_swap": (WORKER.replace("ARGS", "['16', '8', '60000', '24', 'mixed', '1']"), {}),
    "four_engines": (WORKER.replace("ARGS", "['16', '8', '48000', '16', 'many', '4']"), {}),
    "ladder_two_lanes": (None, {"HCV_TAIL_RATIO": "8"}),
    "ladder_nxm_pivot": (None, {"HCV_TAIL_RATIO": "8", "HCV_NXM_LADDER": "1"}),
    "small_engine_8to1": (PIPE.replace("NIN", "8"), {}),
    "small_engine_1x1_pipe2": (PIPE.replace("NIN", "1"), {"HCV_COOP": "0"}),
    "streamed_no_whole_hop_mode": (WORKER.replace("ARGS", "['16', '8', '60000', '24', 'mixed', '1']"), {"HCV_TAIL_HEAD": "0"}),
}


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_documented_order_holds(name):
    code, env = SCENARIOS[name]
    v, err = _run(code or _ladder_code(), dict(env, HCV_ORDER_CHECK="1"))
    assert v == 0, err[-3000:]


def test_the_check_is_off_by_default_and_sees_a_missing_wait():
    code, env = SCENARIOS["ladder_two_lanes"]
    v, _ = _run(_ladder_code(), env)
    assert v == -1                                              # not enabled: nothing is followed
    found = 0
    for k in (2, 3, 5):
        v, err = _run(_ladder_code(), dict(env, HCV_ORDER_CHECK="1", HCV_ORDER_CHECK_DROP=str(k)))
        found += v
        if v:
            assert "order check:" in err
    assert found > 0


def test_roctx_ranges_on_request():
    """HCV_ROCTX=1: `hcv:block` / `hcv:set` / `hcv:resize` marker ranges through the marker library opened at run
    time; the smoke stream — live IR swap included — still equals the oracle's, with and without a profiler attached to read them."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], capture_output=True, text=True, timeout=600, cwd=root,
                         env=dict(os.environ, HCV_ROCTX="1"))
    assert out.returncode == 0 and "smoke ok" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]
