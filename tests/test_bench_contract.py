"""The bench.py contract: one JSON line on stdout with the driver's keys plus `roofline` and `cpu_baseline`; no GPU, no number."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"}


def run_bench(*args, timeout=600):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    out = run_bench("--steps", "1", "--warmup", "0", "--no-cpu-baseline", timeout=300)
    assert out.returncode != 0 and "needs a GPU" in (out.stdout + out.stderr)
    assert not any(line.startswith("{") for line in out.stdout.splitlines())          # never a number from a fallback


def test_bench_options_exist():
    out = run_bench("--help", timeout=120)
    assert out.returncode == 0
    for opt in ("--gpus", "--steps", "--warmup", "--workload", "--sharding"):
        assert opt in out.stdout


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields():
    out = run_bench("--workload", "c3", "--steps", "5", "--warmup", "2", "--no-all-cores", "--extended-ratio", "0")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [line for line in out.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1                                                              # ONE JSON line
    d = json.loads(lines[0])
    assert KEYS <= set(d), KEYS - set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "Msamples/s" and d["dtype"] == "f32" and d["scaling"] == "weak" and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 1 * 8192 * 5 / (d["ms_per_step"] * 5e-3) / 1e6) <= 0.02 * d["value"]     # 1 output, 8192-sample steps
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "Msamples/s" and c["sample"]
