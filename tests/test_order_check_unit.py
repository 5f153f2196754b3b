"""The order checker's logic without a GPU (hisstools_library_amd/csrc/hcv_order_check.h: vector clocks over streams and events, ring
positions, the host's waits, hand-overs inside a launch): tests/cpp/order_check_unit.cpp on made-up handles.  The checker at work on
the engine's real enqueue order: tests/test_order_check_gpu.py."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_order_checker_logic(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "order_check_unit")
    build = subprocess.run([hipcc, "-std=c++17", "-O1", "-x", "hip", "--offload-arch=gfx950", os.path.join(ROOT, "tests", "cpp", "order_check_unit.cpp"),
                            "-I", os.path.join(ROOT, "hisstools_library_amd", "csrc"), "-o", exe], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert run.returncode == 0, run.stdout + run.stderr[-2000:]
    lines = run.stdout.splitlines()
    assert sum(l.startswith("ok ") for l in lines) == 16 and not any(l.startswith("FAILED") for l in lines), run.stdout
    assert "[hcv] order check:" in run.stderr                      # the reports name both accesses and both streams
