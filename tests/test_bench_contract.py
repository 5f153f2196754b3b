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
    for opt in ("--gpus", "--steps", "--warmup", "--workload", "--sharding", "--scaling", "--realtime-block", "--also"):
        assert opt in out.stdout


def test_bench_inputs_are_the_survey_8d_generator(oracle):
    """bench.py draws its audio and IRs from SURVEY 8d's mt19937 generator without touching oracle/: the audio must be
    bit-identical to the oracle's, the IRs equal to within one float32 ulp on a handful of samples (libm pow vs torch's)."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    for ch in (0, 3, 15):
        assert np.array_equal(bench.synth_audio(ch, 50000), oracle.synth_audio(ch, 50000))
    L = 120000
    synth = bench.IrSynth(L, torch.device("cpu"), [(2, 5), (0, 0), (63, 63)], workers=2)
    for i, o in ((2, 5), (0, 0), (63, 63)):
        h, ref = synth.get(i, o).numpy(), oracle.synth_ir(i, o, L)
        assert abs(float(np.dot(h.astype(np.float64), h.astype(np.float64))) - 1.0) < 1e-6
        assert np.abs(h - ref).max() <= 2.0 ** -23 * np.abs(ref).max() and (h != ref).mean() < 1e-4
    synth.close()


def test_shard_plans_cover_the_matrix_exactly_once():
    sys.path.insert(0, ROOT)
    import bench
    for (nin, nout, world) in ((64, 64, 8), (16, 16, 4), (8, 1, 8), (8, 1, 2), (16, 16, 1), (64, 64, 2)):
        seen = set()
        for r in range(world):
            p = bench.shard_plan(nin, nout, world, r, "strong", "rows")
            assert p["nin_total"] == nin and p["nout_total"] == nout and p["go"] * p["gi"] == world
            for i in range(*p["in"]):
                for o in range(*p["out"]):
                    assert (i, o) not in seen
                    seen.add((i, o))
        assert len(seen) == nin * nout
    # weak scaling: every rank brings its own rows
    p = bench.shard_plan(16, 16, 8, 3, "weak", "rows")
    assert p["out"] == (48, 64) and p["in"] == (0, 16) and p["nout_total"] == 128
    p = bench.shard_plan(16, 16, 8, 5, "weak", "grid")
    assert p["out"] == (32, 48) and p["in"] == (8, 16) and p["nout_total"] == 64 and p["gi"] == 2


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields(tmp_path):
    details = str(tmp_path / "details.json")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c3", "--steps", "5", "--warmup", "2", "--no-all-cores", "--extended-ratio", "0",
                          "--also", "c2,m16l"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, BENCH_DETAILS=details))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [line for line in out.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1                                                              # ONE JSON line
    assert len(lines[0]) < 4096                                                         # ... that an 8 KB output tail holds whole
    d = json.loads(lines[0])
    assert KEYS <= set(d), KEYS - set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "Msamples/s" and d["dtype"] == "f32" and d["scaling"] == "weak" and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 1 * 8192 * 5 / (d["ms_per_step"] * 5e-3) / 1e6) <= 0.02 * d["value"]     # 1 output, 8192-sample steps
    # every config entry is a scalar: the record survives a parser that keeps scalars only
    assert all(v is None or isinstance(v, (str, int, float, bool)) for v in d["config"].values()), d["config"]
    r = d["roofline"]
    assert all(v is None or isinstance(v, (str, int, float, bool)) for v in r.values()), r
    # c3's live spectra (a few MB) sit in the Infinity Cache: the line must say so instead of quoting an HBM fraction, name the kernel that
    # ran — the block is ONE fused launch — and never quote a kernel time above the step's
    assert r["bound"] == "launch" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and "traffic" in r
    assert r["kernel"].startswith("fused_block_nx1_kernel") and r["launches"] > 0
    assert r["avg_launch_ms"] is None or 0 < r["avg_launch_ms"] <= d["ms_per_step"]
    assert d["config"]["self_check_ok"] is True and d["config"]["max_rel_err"] <= 1e-5
    assert d["config"]["rt128_host_p99_ms"] > 0 and d["config"]["rt128_dev_p99_ms"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "Msamples/s" and c["sample"]
    assert c["flags"] == "-O3 -msse2" and c["block"] == 512 and c["best_of"] == 3 and c["b2048"] > 0        # BASELINE.md section 3's protocol
    # the further workloads as flat scalars (default run: ns64, c4, c3, c2, c1)
    cfg = d["config"]
    assert "c2_error" not in cfg and "m16l_error" not in cfg, cfg
    assert cfg["c2_msamples_per_s"] > 0 and cfg["c2_ms_per_step"] > 0 and cfg["c2_max_rel_err"] <= 1e-5 and cfg["c2_self_check_ok"] is True
    assert cfg["c2_cpu_1core"] > 0 and cfg["c2_kernel"] == "fused_block_hops_kernel"
    assert 0 < cfg["m16l_mac_frac"] < 1 and cfg["m16l_max_rel_err"] <= 1e-5 and cfg["m16l_mac_ms"] <= cfg["m16l_ms_per_step"]
    # the rich record went to the side file
    assert cfg["details_file"] == details
    rich = json.load(open(details))
    assert len(rich["config"]["also"]) == 2 and rich["config"]["also"][0]["workload"].startswith("c2:")
    assert rich["config"]["self_check"]["ok"] and rich["config"]["realtime"]["finite"]
