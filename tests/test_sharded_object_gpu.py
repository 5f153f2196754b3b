"""Multi-GPU as a property of the C ABI: ONE Convolver object driving several engines
(hcv_convolver_create_sharded; Convolver.h:23-50 stays the interface, the per-output sum of NToMonoConvolve.cpp:39-42 is what
crosses devices).  On the one-GPU test box the device list names the same GPU twice or more: two or four HIP engines behind one
object, the same code path as distinct devices minus the peer mapping.

Bars: a sharded run matches the unsharded HIP output to <= 1e-6 (row split: identical arithmetic per row) and
<= 1e-5 (input split: the sum order changes), and both match the CPU oracle within the stated tolerance.
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


@pytest.fixture(scope="module")
def H():
    import hisstools_library_amd as H
    assert H.load().hcv_device_count() > 0, "no GPU visible: the HIP path cannot run (and there is no fallback)"
    return H


def _devs(H, n):
    """The device list of an n-shard object: the same GPU n times on a one-GPU box (what these tests were written on), and
    DISTINCT devices round-robin wherever more than one is visible — so that peer access, cross-device event waits and peer reads of
    the caller's buffers (hcv_api.hip's sharded paths; the sum of NToMonoConvolve.cpp:39-42 crossing GPUs) run the first time a
    multi-GPU box sees the suite."""
    count = H.load().hcv_device_count()
    return [k % count for k in range(n)] if count > 1 else [0] * n


def _load(c, irs):
    for (i, o), h in irs.items():
        assert c.set(i, o, h, True) == 0


@pytest.mark.parametrize("nin,nout,ndev,tol", [(4, 4, 2, 1e-6), (5, 3, 2, 1e-6), (8, 1, 2, 1e-5), (8, 1, 4, 1e-5), (6, 2, 4, 1e-5)])
def test_sharded_object_matches_unsharded_and_oracle(H, oracle, nin, nout, ndev, tol):
    L, S = 30000, 60000
    irs = {(i, o): oracle.synth_ir(i, o, L - 700 * i - 90 * o) for i in range(nin) for o in range(nout)}
    xs = np.stack([oracle.synth_audio(i, S) for i in range(nin)])
    one = H.Convolver(nin, nout, 0)
    many = H.Convolver(nin, nout, 0, devices=_devs(H, ndev))
    assert many.num_shards() == ndev and one.num_shards() == 1
    ref = oracle.Convolver(nin, nout, 0)
    ref.setResetOffset(0)
    for c in (one, many, ref):
        _load(c, irs)
    blocks = [512, 8192, 100, 3000, 16384]
    y1, yn, yr = one.run(xs, nout, blocks), many.run(xs, nout, blocks), ref.run(xs, nout, 2048)
    for o in range(nout):
        assert rel_err(yn[o], y1[o]) < tol, (o, rel_err(yn[o], y1[o]))
        assert rel_err(yn[o], yr[o]) < 1e-5
    # a live IR swap and a reset of single pairs go to the shard that owns them
    new_ir = oracle.synth_ir(9, 9, L)
    for c in (one, many):
        assert c.set(nin - 1, nout - 1, new_ir, True) == 0
        assert c.reset(0, 0) == 0
    y1b, ynb = one.run(xs[:, :20000], nout, 1000), many.run(xs[:, :20000], nout, 1000)
    for o in range(nout):
        assert rel_err(ynb[o], y1b[o]) < max(tol, 2e-6)
    # channel-range errors are those of the unsharded object (Convolver.cpp:88-134)
    assert many.set(nin, 0, new_ir, True) == one.set(nin, 0, new_ir, True) == 1
    assert many.set(0, nout, new_ir, True) == one.set(0, nout, new_ir, True) == 2
    assert many.resize(0, nout, 100) == one.resize(0, nout, 100) == 1
    assert many.reset(nin + 3, 0) == one.reset(nin + 3, 0) == 1


def test_sharded_object_fewer_active_channels(H, oracle):
    """process with fewer inputs / outputs than constructed (Convolver.cpp:148-153): the active ranges are split per shard."""
    nin, nout, L, S = 6, 4, 9000, 20000
    irs = {(i, o): oracle.synth_ir(i, o, L) for i in range(nin) for o in range(nout)}
    xs = np.stack([oracle.synth_audio(i, S) for i in range(nin)])
    one, many = H.Convolver(nin, nout, 0), H.Convolver(nin, nout, 0, devices=_devs(H, 2))
    for c in (one, many):
        _load(c, irs)
    for (ai, ao) in ((6, 4), (3, 3), (5, 1)):
        for c in (one, many):
            c.reset()
        ya, yb = np.zeros((nout, S), np.float32), np.zeros((nout, S), np.float32)
        one.process(xs, ya, ai, ao)
        many.process(xs, yb, ai, ao)
        for o in range(ao):
            assert rel_err(yb[o], ya[o]) < 2e-6, (ai, ao, o)
        assert not yb[ao:].any()


def test_sharded_object_parallel_mode_and_double_api(H, oracle):
    n, L, S = 5, 7000, 30000
    irs = [oracle.synth_ir(o, o, L - 100 * o) for o in range(n)]
    xs = np.stack([oracle.synth_audio(o, S) for o in range(n)])
    one, many = H.Convolver(n, None, 1), H.Convolver(n, None, 1, devices=_devs(H, 3))
    for c in (one, many):
        for o in range(n):
            assert c.set(o, o, irs[o], True) == 0
        assert c.set(0, 1, irs[0], True) == 1                     # parallel mode: only in == out (Convolver.cpp:92,106,118)
    y1, yn = one.run(xs, n, 4096), many.run(xs, n, 4096)
    for o in range(n):
        assert rel_err(yn[o], y1[o]) < 1e-6
    for c in (one, many):
        c.reset()
    d1, dn = one.run(xs.astype(np.float64), n, 777), many.run(xs.astype(np.float64), n, 777)
    for o in range(n):
        assert rel_err(dn[o], d1[o]) < 1e-6


@pytest.mark.parametrize("nin,nout,ndev", [(4, 4, 2), (8, 1, 4), (8, 2, 4)])
def test_sharded_object_device_pointers(H, oracle, nin, nout, ndev):
    """HBM-resident audio: every shard reads / writes the caller's buffers in place; the input-split layouts sum the row
    group's partial blocks with one kernel on the group's root stream (events across the shards' streams)."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    L, S, B = 40000, 10 * 8192, 8192
    irs = {(i, o): oracle.synth_ir(i, o, L) for i in range(nin) for o in range(nout)}
    xs_h = np.stack([oracle.synth_audio(i, S) for i in range(nin)])
    xs = torch.from_numpy(xs_h).to(dev)
    one = H.Convolver(nin, nout, 0, maxBlock=B)
    many = H.Convolver(nin, nout, 0, maxBlock=B, devices=_devs(H, ndev))
    for c in (one, many):
        _load(c, irs)
    ys = [torch.zeros((nout, S), device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    for c, y in zip((one, many), ys):
        for pos in range(0, S, B):                                     # asynchronous calls back to back, then one wait
            c.process_dev(xs.data_ptr() + 4 * pos, S, y.data_ptr() + 4 * pos, S, nin, nout, B)
        c.synchronize()
    a, b = ys[0].cpu().numpy(), ys[1].cpu().numpy()
    for o in range(nout):
        assert rel_err(b[o], a[o]) < (1e-6 if nout >= ndev else 1e-5), (o, rel_err(b[o], a[o]))
    # ragged and small calls (the deferred tail) through the sharded object, against the oracle
    ref = oracle.Convolver(nin, nout, 0)
    ref.setResetOffset(0)
    _load(ref, irs)
    yr = ref.run(xs_h[:, :30000], nout, 2048)
    many.reset()
    y2 = torch.zeros((nout, 30000), device=dev)
    pos = 0
    for n in [128] * 100 + [1000, 333, 8192, 5000, 2675]:
        many.process_dev(xs.data_ptr() + 4 * pos, S, y2.data_ptr() + 4 * pos, 30000, nin, nout, n, sync=True)
        pos += n
    assert pos == 30000
    for o in range(nout):
        assert rel_err(y2[o].cpu().numpy(), yr[o]) < 1e-5


def test_env_devices_shards_the_reference_shaped_constructor(H):
    """HCV_DEVICES makes hcv_convolver_create (what HISSTools::Convolver calls) build the sharded object."""
    code = ("import hisstools_library_amd as H, numpy as np\n"
            "c = H.Convolver(3, 4, 0)\n"
            "assert c.set(2, 3, np.ones(100, np.float32), True) == 0\n"
            "y = c.run(np.ones((3, 600), np.float32), 4, 256)\n"
            "print(c.num_shards(), float(y[3, -1]), float(y[0, -1]))\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(os.environ, HCV_DEVICES="0,0"))
    assert out.returncode == 0, out.stderr[-2000:]
    shards, last, silent = out.stdout.split()[-3:]
    assert int(shards) == 2 and abs(float(last) - 100.0) < 1e-3 and float(silent) == 0.0


def test_rccl_allreduce_entry_point_world_of_one(H, oracle):
    """hcv_convolver_comm_init + process_f32_dev_allreduce with a one-rank communicator: RCCL is found and bound at run time, the
    collective is enqueued on the engine's stream behind the block, and (one rank) leaves the block as process_dev wrote it."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    nin, nout, L, S, B = 3, 2, 20000, 4 * 8192, 8192
    irs = {(i, o): oracle.synth_ir(i, o, L) for i in range(nin) for o in range(nout)}
    xs = torch.from_numpy(np.stack([oracle.synth_audio(i, S) for i in range(nin)])).to(dev)
    a, b = H.Convolver(nin, nout, 0, maxBlock=B), H.Convolver(nin, nout, 0, maxBlock=B)
    for c in (a, b):
        _load(c, irs)
    b.comm_init(H.rccl_unique_id(), 0, 1)
    ya, yb = torch.zeros((nout, S), device=dev), torch.zeros((nout, S), device=dev)
    torch.cuda.synchronize()
    for pos in range(0, S, B):
        a.process_dev(xs.data_ptr() + 4 * pos, S, ya.data_ptr() + 4 * pos, S, nin, nout, B)
        b.process_dev_allreduce(xs.data_ptr() + 4 * pos, S, yb.data_ptr() + 4 * pos, S, nin, nout, B)      # strided rows: grouped calls
    a.synchronize()
    b.synchronize()
    assert torch.equal(ya, yb)


@pytest.mark.parametrize("workload,flags", [("c3", ["--scaling", "strong"]), ("m16", ["--scaling", "strong"]), ("m16", ["--sharding", "grid"]), ("m16", [])])
def test_bench_two_ranks_share_the_gpu(workload, flags):
    """bench.py --gpus 2 with two ranks on the one GPU (gloo for the rendezvous and, on the reduce path, for the all-reduce):
    strong scaling splits the workload's own matrix — c3 (8 -> 1) by inputs with one all-reduce per step, m16 by output rows."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BENCH_BACKEND="gloo", MASTER_PORT=str(port))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", workload, "--steps", "4", "--warmup", "1", "--batched-block", "0",
           "--extended-ratio", "0", "--realtime-block", "0", "--no-cpu-baseline"] + flags
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["finite_output"]
    if not flags:
        assert d["scaling"] == "weak" and "(16x32 over 2 GPU)" in d["config"]["workload"]          # every rank brings its own 16 rows
    if "strong" in flags:
        assert d["scaling"] == "strong"
        nin, nout = {"c3": (8, 1), "m16": (16, 16)}[workload]
        assert f"({nin}x{nout} over 2 GPU)" in d["config"]["workload"]


def test_bench_default_two_ranks_attach_strong_legs():
    """A bare `bench.py --gpus 2` (what the driver's scaling run calls, here with two ranks on the one GPU over gloo): the headline
    stays the weak-scaled default workload, and the line carries BASELINE config 4 as stated — the 64x64 matrix split over the
    ranks by output rows — and config 3's input split with one all-reduce per step, each strong-scaled on the same ranks with a
    self-check in which every rank streams its share: flat scalars in the line, the digests in the side file."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile
    details = os.path.join(tempfile.mkdtemp(), "details.json")
    env = dict(os.environ, BENCH_BACKEND="gloo", MASTER_PORT=str(port), BENCH_DETAILS=details)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--batched-block", "0",
           "--extended-ratio", "0", "--realtime-block", "0", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "(16x32 over 2 GPU)" in d["config"]["workload"] and d["value"] > 0
    cfg = d["config"]
    for k in ("c4_strong", "c3_strong"):
        assert k + "_error" not in cfg, cfg
        assert cfg[k + "_msamples_per_s"] > 0 and cfg[k + "_ms_per_step"] > 0 and cfg[k + "_self_check_ok"] is True and cfg[k + "_max_rel_err"] <= 1e-5, cfg
    # config 4 in BOTH strong layouts (rows over the ranks; (N/2) x 2 grid with one all-reduce per step), the faster one quoted with its
    # efficiency against the whole matrix on one GPU measured in the same run
    assert cfg["c4_strong_layout"] in ("rows 2 x 1", "grid 1 x 2") and cfg["c4_strong_msamples_per_s"] > 0 and cfg["c4_one_gpu_msamples_per_s"] > 0
    assert 0 < cfg["c4_strong_efficiency"] < 1.5 and cfg["c4_strong_rows_msamples_per_s"] > 0 and cfg["c4_strong_grid_msamples_per_s"] > 0
    assert cfg["c4_strong_grid_max_rel_err"] <= 1e-5 and "c4_strong_grid_error" not in cfg
    legs = json.load(open(details))["config"]["also"]
    assert [a["workload"].split(":")[0] for a in legs] == ["c4", "c4grid", "c3"]
    legs = [legs[0], legs[2]]
    for a, shape, word in zip(legs, ("(64x64 over 2 GPU)", "(8x1 over 2 GPU)"), ("output rows per rank", "all-reduce")):
        assert "error" not in a, a
        assert a["scaling"] == "strong" and a["n_gpus"] == 2 and shape in a["workload"] and word in a["sharding"]
        assert a["value"] > 0 and a["self_check"]["ok"] and a["self_check"]["max_rel_err"] <= 1e-5, a


def test_the_drivers_eight_rank_command_on_one_gpu():
    """`bench.py --gpus 8 --steps 20 --warmup 5` — the exact command shape of the driver's scaling run — with the eight ranks sharing
    the one GPU over gloo and the impulse responses 32 times shorter (BENCH_IR_DIV, a test aid the line owns up to): rc 0, ONE short
    JSON line for N = 8, the weak-scaled headline with both strong-scaled digests (config 4 split by rows over eight ranks, config 3's
    inputs split eight ways with one all-reduce per step) checked against the reference, in bounded wall time."""
    import socket
    import tempfile
    import time
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    details = os.path.join(tempfile.mkdtemp(), "details.json")
    env = dict(os.environ, BENCH_BACKEND="gloo", MASTER_PORT=str(port), BENCH_IR_DIV="32", BENCH_DETAILS=details)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"], capture_output=True, text=True,
                         timeout=1500, cwd=ROOT, env=env)
    wall = time.time() - t0
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096
    d = json.loads(lines[0])
    cfg = d["config"]
    assert d["n_gpus"] == 8 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak" and d["value"] > 0
    assert "(16x128 over 8 GPU)" in cfg["workload"] and cfg["reduced_ir_div"] == 32 and cfg["finite_output"]
    for k in ("c4_strong", "c3_strong"):
        assert k + "_error" not in cfg, cfg
        assert cfg[k + "_msamples_per_s"] > 0 and cfg[k + "_self_check_ok"] is True and cfg[k + "_max_rel_err"] <= 1e-5, cfg
    assert cfg["c4_strong_layout"] in ("rows 8 x 1", "grid 4 x 2") and cfg["c4_strong_msamples_per_s"] > 0 and cfg["c4_strong_efficiency"] > 0
    assert cfg["c4_strong_grid_max_rel_err"] <= 1e-5
    legs = json.load(open(details))["config"]["also"]
    assert [a["n_gpus"] for a in legs] == [8, 8, 8] and "(64x64 over 8 GPU)" in legs[0]["workload"] and "(8x1 over 8 GPU)" in legs[2]["workload"]
    assert "grid 4 x 2" in legs[1]["sharding"]
    assert wall < 1200, wall


def test_bare_multi_rank_run_keeps_its_headline_when_a_strong_leg_hangs():
    """The strong-scaled legs of a bare `bench.py --gpus N` must never cost the run its headline: with the watchdog set to fire at
    once (BENCH_ALSO_TIMEOUT) rank 0 still prints exactly one JSON line — the weak-scaled headline — with the failure noted."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BENCH_BACKEND="gloo", MASTER_PORT=str(port), BENCH_ALSO_TIMEOUT="0.2")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--batched-block", "0",
           "--extended-ratio", "0", "--realtime-block", "0", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert any(k.endswith("_error") for k in d["config"]), d["config"]


def test_random_cases_through_sharded_objects(monkeypatch):
    """The randomised differential test (tests/perf/fuzz_parity.py: random matrices, latencies, call sizes, live IR swaps, clears and
    resets) with HCV_DEVICES set, so that every Convolver it builds is ONE object over two engines."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
    import fuzz_parity
    monkeypatch.setenv("HCV_DEVICES", "0,0")
    ran = 0
    for seed in range(7001, 7061):
        kind, desc, err = fuzz_parity.one_case(seed)
        assert err <= fuzz_parity.TOL, (seed, kind, desc, err)
        ran += kind in ("convolver", "parallel")
    assert ran >= 20


def _sparse_sharded_case(H, torch, nin, nout, L, hops, taps, seed, ndev, tol_vs_one, B=8192):
    """BASELINE config shapes through ONE object over `ndev` engines (all on the one GPU of the test box): sparse taps spread over
    the whole IR, built and streamed in HBM; compared with the unsharded HIP object and with the exact float64 answer (a
    gain-weighted sum of delayed inputs, computed with torch: shifted adds, nothing of this library)."""
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(seed)
    S = hops * B
    P = -(-(L - B) // B)
    pairs = nin * nout
    delays = rng.randint(0, L, size=(nout, nin, taps))
    walk = (np.arange(pairs) % P) * B + B + rng.randint(0, B, size=pairs)       # every tail partition index carries a tap of some pair
    delays[:, :, 0] = np.minimum(walk, L - 1).reshape(nout, nin)
    gains = rng.uniform(-1, 1, size=(nout, nin, taps))
    one = H.Convolver(nin, nout, 0, custom=(L, True, 256, 1024, 4096, 16384), maxBlock=B)
    many = H.Convolver(nin, nout, 0, custom=(L, True, 256, 1024, 4096, 16384), maxBlock=B, devices=_devs(H, ndev))
    assert many.num_shards() == ndev
    h = torch.zeros(L, dtype=torch.float32, device=dev)
    for o in range(nout):
        for i in range(nin):
            idx = torch.from_numpy(delays[o, i]).to(dev)
            h.index_put_((idx,), torch.from_numpy(gains[o, i].astype(np.float32)).to(dev), accumulate=True)
            torch.cuda.synchronize()
            assert one.set_dev(i, o, h.data_ptr(), L, True) == 0 and many.set_dev(i, o, h.data_ptr(), L, True) == 0
            h.index_fill_(0, idx, 0.0)
    xs = torch.from_numpy(rng.uniform(-1, 1, size=(nin, S)).astype(np.float32)).to(dev)
    ys = [torch.zeros((nout, S), dtype=torch.float32, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    for c, y in zip((one, many), ys):
        for pos in range(0, S, B):                                   # asynchronous hop-sized calls back to back, one wait
            c.process_dev(xs.data_ptr() + 4 * pos, S, y.data_ptr() + 4 * pos, S, nin, nout, B)
        c.synchronize()
    x64 = xs.to(torch.float64)
    for o in range(nout):
        t = torch.zeros(S, dtype=torch.float64, device=dev)
        for i in range(nin):
            for k in range(taps):
                d = int(delays[o, i, k])
                if d < S:
                    t[d:] += float(np.float32(gains[o, i, k])) * x64[i, : S - d]
        peak = float(t.abs().max())
        e_truth = float((ys[1][o].to(torch.float64) - t).abs().max()) / peak
        e_one = float((ys[1][o] - ys[0][o]).abs().max()) / peak
        assert e_truth < 1e-5 and e_one < tol_vs_one, (o, e_truth, e_one)
    return many


def test_config4_shape_64x64_2s_eight_shards(H):
    """BASELINE config 4 as stated — Convolver 64x64, 2 s @ 48 kHz IRs, channel pairs sharded eight ways — through ONE object:
    output rows split over 8 engines (8 rows x 64 inputs each, no exchange; Convolver.cpp:138-154 per row).  Row sharding leaves
    each row's arithmetic unchanged: <= 1e-6 of the unsharded HIP output."""
    torch = pytest.importorskip("torch")
    many = _sparse_sharded_case(H, torch, 64, 64, 96000, 16, 2, seed=404, ndev=8, tol_vs_one=1e-6)
    tail = many.stage_stats()[-1]                                    # (statistics of the first shard)
    assert tail["num_ins"] == 64 and tail["num_outs"] == 8 and tail["partitions"] == 11


def test_config3_shape_8to1_5s_eight_shards_input_split(H):
    """BASELINE config 3's matrix (8 -> 1, 5 s IRs) over 8 engines: a pure input split — every shard convolves ONE input and the
    per-output sum of NToMonoConvolve.cpp:39-42 is taken across the shards (sum_parts on the root's stream).  The sum order
    changes: <= 1e-5 of the unsharded HIP output."""
    torch = pytest.importorskip("torch")
    many = _sparse_sharded_case(H, torch, 8, 1, 240000, 40, 3, seed=303, ndev=8, tol_vs_one=1e-5)
    tail = many.stage_stats()[-1]
    assert tail["num_ins"] == 1 and tail["num_outs"] == 1 and tail["partitions"] == 29


@pytest.mark.parametrize("nin,nout,ndev,tol", [(4, 4, 2, 1e-6), (8, 1, 4, 1e-5), (6, 2, 4, 1e-5), (9, 9, 8, 1e-6)])
def test_sharded_object_enqueue_threads(H, oracle, monkeypatch, nin, nout, ndev, tol):
    """One persistent enqueue thread per shard (hcv_shard_pool.h; the default when the shards are on several devices, forced here
    on the one GPU): host-pointer calls (begin halves side by side, row groups' end halves side by side) and device-pointer
    calls (blocks side by side, then the row roots' sums) give the unsharded object's stream; many small calls hammer the
    post / report hand-off, pauses longer than the workers' spin window send them to sleep on their futex."""
    import time
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    monkeypatch.setenv("HCV_SHARD_THREADS", "1")
    L, S, B = 20000, 6 * 8192, 8192
    irs = {(i, o): oracle.synth_ir(i, o, L - 300 * i - 70 * o) for i in range(nin) for o in range(nout)}
    xs_h = np.stack([oracle.synth_audio(i, S) for i in range(nin)])
    one = H.Convolver(nin, nout, 0, maxBlock=B)
    many = H.Convolver(nin, nout, 0, maxBlock=B, devices=_devs(H, ndev))
    for c in (one, many):
        _load(c, irs)
    blocks = [64] * 200 + [8192, 100, 3000, 8192]
    y1, yn = one.run(xs_h, nout, blocks), many.run(xs_h, nout, blocks)
    for o in range(nout):
        assert rel_err(yn[o], y1[o]) < tol, (o, rel_err(yn[o], y1[o]))
    xs = torch.from_numpy(xs_h).to(dev)
    ys = [torch.zeros((nout, S), device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    for c, y in zip((one, many), ys):
        c.reset()
        for k, pos in enumerate(range(0, S, B)):
            if k == 3:
                time.sleep(0.01)                                       # the workers go to sleep; the next call wakes them
            c.process_dev(xs.data_ptr() + 4 * pos, S, y.data_ptr() + 4 * pos, S, nin, nout, B)
        c.synchronize()
    a, b = ys[0].cpu().numpy(), ys[1].cpu().numpy()
    for o in range(nout):
        assert rel_err(b[o], a[o]) < tol, (o, rel_err(b[o], a[o]))
    # control calls from this thread while the workers exist, then a destroy with sleeping workers
    assert many.set(nin - 1, nout - 1, irs[(0, 0)], True) == 0 and one.set(nin - 1, nout - 1, irs[(0, 0)], True) == 0
    y1b, ynb = one.run(xs_h[:, :9000], nout, 1000), many.run(xs_h[:, :9000], nout, 1000)
    for o in range(nout):
        assert rel_err(ynb[o], y1b[o]) < max(tol, 2e-6)
    time.sleep(0.01)
    del many
