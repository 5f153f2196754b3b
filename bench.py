#!/usr/bin/env python3
"""bench.py — throughput + HBM roofline of the partitioned-convolution hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c5|c4|c3|c2|ns64] [--block B]

One "step" = one Convolver::process call of B samples (default 8192 = one hop of the 16384-point tail stage) over
the whole channel matrix, audio and IR spectra resident in HBM.  Metric (BASELINE.json): output-channel
Msamples/s for the node, next to the achieved-vs-peak HBM bandwidth of the dominant kernel (spectral_mac of the
tail stage).

Workloads (BASELINE.json configs; default c5 = the config the metric's HBM clause is quoted on):
    c5    Convolver 16x16, 60 s @ 96 kHz IRs (L = 5,760,000), zero latency      11.8 GB of tail spectra
    c4    Convolver 64x64, 2 s @ 48 kHz IRs  (L = 96,000),    zero latency
    c3    NToMono-shaped 8 -> 1, 5 s IRs     (L = 240,000),   zero latency
    c2    PartitionedConvolve-shaped 1x1, 10 s IR, one 4096-point stage (cache resident, launch bound)
    ns64  64x64, 10 s @ 48 kHz IRs (north-star target shape),  zero latency     15.7 GB of spectra

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank owns its own block of `nout` output rows
of an (N*nout) x nin system and receives the same inputs — output-row sharding, no collective on the data path —
so per-GPU work is fixed: weak scaling.

The CPU baseline leg (rank 0, N = 1 only) times the UNMODIFIED reference (oracle/_ref, when the prebuilt library
travelled with the repo; else the C port) on a bounded sub-matrix of the same workload on one host core.

`--sharding grid` exercises the other half of SURVEY 8e instead: (N/2) x 2 ranks, the two ranks of a row group convolve half of
the inputs each and sum their partial outputs with ONE all-reduce per step (RCCL; BENCH_BACKEND=gloo lets two ranks share a GPU to
check the path on a one-GPU box).  The default stays output-row sharding.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak (≈6.3 TB/s achievable)

WORKLOADS = {
    #        nin nout  L         fs     layout (zeroLatency, A, B, C, D)
    "c5":   (16, 16, 5760000, 96000, (True, 256, 1024, 4096, 16384)),
    "c4":   (64, 64, 96000,   48000, (True, 256, 1024, 4096, 16384)),
    "c3":   (8,  1,  240000,  48000, (True, 256, 1024, 4096, 16384)),
    "c2":   (1,  1,  480000,  48000, (False, 4096, 0, 0, 0)),
    "ns64": (64, 64, 480000,  48000, (True, 256, 1024, 4096, 16384)),
    "m16":  (16, 16, 96000,   48000, (True, 256, 1024, 4096, 16384)),   # mid-size: 16x16, 2 s IRs (not a BASELINE config)
    "m16l": (16, 16, 480000,  48000, (True, 256, 1024, 4096, 16384)),   # mid-size: 16x16, 10 s IRs (not a BASELINE config)
}


def stage_layout(L, layout):
    """(fft_size, partitions) per FFT stage for an IR of L samples — MonoConvolve::setPartitions arithmetic."""
    zero, *sizes = layout
    sizes = [s for s in sizes if s]
    offset = sizes[0] // 2 if zero else 0
    out = []
    fixed = list(zip(sizes[:-1], sizes[1:]))
    for size, nxt in fixed:
        seg = (nxt - size) // 2
        take = max(0, min(seg, L - offset))
        out.append((size, -(-take // (size // 2))))
        offset += seg
    tail = sizes[-1]
    out.append((tail, -(-max(0, L - offset) // (tail // 2))))
    return out


def algorithmic_bytes_per_hop(H, P, nin, nout):
    """bytes one hop of one stage must move for the whole matrix (fp32, hop-streaming)."""
    return 8 * H * P * nin * nout + 8 * H * P * nin + 8 * H * nin + 4 * H * (nin + nout)


def cpu_baseline(workload, hops=64):
    """Reference CPU path on one host core, on a bounded sub-matrix of the same workload (steady state: the
    stream is first run for as many hops as the tail has partitions so every partition is live, then timed)."""
    import numpy as np
    from oracle import oracle as O

    nin, nout, L, fs, layout = WORKLOADS[workload]
    kind = "reference" if O.have_ref() else "port"
    backend = "ref" if kind == "reference" else "port"
    tail, p_tail = stage_layout(L, layout)[-1]
    max_pairs = max(1, min(32, 3000 // max(1, p_tail)))
    sub_in = min(nin, 8)
    while sub_in > 1 and sub_in > max_pairs:
        sub_in //= 2
    sub_out = max(1, min(nout, max_pairs // sub_in))
    hop = tail // 2
    warm, S = p_tail * hop, hops * hop
    block = 512
    xs = np.stack([O.synth_audio(i, warm + S) for i in range(sub_in)])
    t_set = time.perf_counter()
    if workload == "c2":
        block = 2048
        p = O.PartitionedConvolve(4096, L, 0, 0, backend=backend)
        p.setResetOffset(0)
        p.set(O.synth_ir(0, 0, L))
        t_set = time.perf_counter() - t_set
        p.run(xs[0, :warm], block)
        t0 = time.perf_counter()
        p.run(xs[0, warm:], block)
        secs = time.perf_counter() - t0
    else:
        c = O.Convolver(sub_in, sub_out, 0, backend=backend)
        for o in range(sub_out):
            for i in range(sub_in):
                c.set(i, o, O.synth_ir(i, o, L), True)
        t_set = time.perf_counter() - t_set
        c.stream_timed(np.ascontiguousarray(xs[:, :warm]), sub_out, block)
        _, secs = c.stream_timed(np.ascontiguousarray(xs[:, warm:]), sub_out, block)
    pair_rate = sub_in * sub_out * S / secs                       # pair-samples / s on one core
    value = pair_rate / nin / 1e6                                 # == output-channel Msamples/s for the full matrix
    return {
        "value": round(value, 6), "unit": "Msamples/s", "cores": 1, "kind": kind,
        "sample": f"{sub_in}x{sub_out} sub-matrix of the {nin}x{nout} workload, same {L}-sample IRs, {S} samples timed in {block}-sample calls "
                  f"after a {warm}-sample warm-up (all partitions live), 1 thread; value = pair-samples/s / {nin} inputs = the whole-matrix "
                  f"output rate one core would sustain; IR load took {t_set:.1f} s",
        "pair_msamples_per_s": round(pair_rate / 1e6, 4),
        "seconds": round(secs, 3),
    }


def cpu_baseline_all_cores(workload, max_threads=64, hops=16):
    """The only parallel decomposition the reference API admits (it has no threads of its own): one Convolver per host
    thread over disjoint output rows.  Each thread streams its own (sub_in x 1) Convolver with the workload's IR length;
    the IR set is synthesised once and shared.  Returns the summed rate as a whole-matrix-equivalent output rate."""
    import threading
    import numpy as np
    from oracle import oracle as O

    nin, nout, L, fs, layout = WORKLOADS[workload]
    kind = "reference" if O.have_ref() else "port"
    backend = "ref" if kind == "reference" else "port"
    threads = max(1, min(os.cpu_count() or 1, max_threads))
    tail, p_tail = stage_layout(L, layout)[-1]
    sub_in = min(nin, 4 if p_tail > 100 else 8)
    hop = tail // 2
    warm, S = p_tail * hop, hops * hop
    irs = [O.synth_ir(i, 0, L) for i in range(sub_in)]
    xs = np.stack([O.synth_audio(i, warm + S) for i in range(sub_in)])
    xw, xt = np.ascontiguousarray(xs[:, :warm]), np.ascontiguousarray(xs[:, warm:])
    secs = [0.0] * threads
    gate = threading.Barrier(threads)

    def worker(t):
        c = O.Convolver(sub_in, 1, 0, backend=backend)
        for i in range(sub_in):
            c.set(i, 0, irs[i], True)
        c.stream_timed(xw, 1, 512)                       # every partition live
        gate.wait()
        _, secs[t] = c.stream_timed(xt, 1, 512)

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    t0 = time.perf_counter()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    wall = time.perf_counter() - t0
    pair_rate = sum(sub_in * S / s_ for s_ in secs if s_ > 0)
    return {
        "value": round(pair_rate / nin / 1e6, 6), "unit": "Msamples/s", "cores": threads, "kind": kind,
        "sample": f"{threads} host threads, each streaming its own {sub_in}x1 Convolver with the workload's {L}-sample IRs ({S} samples timed in "
                  f"512-sample calls after a {warm}-sample warm-up); value = summed pair-samples/s / {nin} inputs; whole leg took {wall:.1f} s",
        "pair_msamples_per_s": round(pair_rate / 1e6, 3),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c5", choices=sorted(WORKLOADS))
    ap.add_argument("--block", type=int, default=8192)
    ap.add_argument("--batched-block", type=int, default=65536, help="also time offline-style calls of this many samples (0 = skip)")
    ap.add_argument("--tail-ratio", type=int, default=0, help="run the HEADLINE on the extended far-tail ladder (0 = reference partitioning)")
    ap.add_argument("--extended-ratio", type=int, default=8, help="also measure the extended far-tail ladder with this ratio (0 = skip)")
    ap.add_argument("--ir-file", default="", help="WAVE / AIFF / AIFC file with real impulse responses instead of the synthetic ones")
    ap.add_argument("--sharding", default="rows", choices=["rows", "grid"],
                    help="rows: every rank owns output rows and all inputs, no data-path collective (default).  grid: (N/2) x 2 ranks — the two "
                         "ranks of a row group each convolve half of the inputs and sum their partial outputs with one RCCL all-reduce per step "
                         "(SURVEY 8e: the reduce path; needs an even N >= 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-all-cores", action="store_true", help="skip the all-host-cores CPU leg")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # convenience: re-launch ourselves one rank per GPU
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", os.environ.get("MASTER_PORT", "29517"), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the convolution engine has no CPU fallback")
    local = local % torch.cuda.device_count()       # (several ranks may share a GPU when the reduce path is exercised with gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    grid = args.sharding == "grid"
    if grid and (world < 2 or world % 2):
        raise SystemExit("--sharding grid needs an even number of ranks (>= 2)")

    import hisstools_library_amd as H

    nin, nout, L, fs, layout = WORKLOADS[args.workload]
    B = args.block
    # grid sharding: rank = row * 2 + col; a row group of two ranks owns `nout` output rows, each rank half of the inputs
    nin_total, row, col, row_group = nin, rank, 0, None
    if grid:
        if nin % 2:
            raise SystemExit("--sharding grid needs an even number of inputs")
        row, col = divmod(rank, 2)
        nin = nin_total // 2
        for r in range(world // 2):                     # every rank creates every group
            grp = dist.new_group(ranks=[2 * r, 2 * r + 1])
            if r == row:
                row_group = grp
    stages = stage_layout(L, layout)

    BB = max(args.batched_block, 0)
    g = torch.Generator(device=dev)
    decay = torch.pow(torch.tensor(10.0, device=dev), -3.0 * torch.arange(L, device=dev, dtype=torch.float32) / L)
    nring = max(8, -(-BB // B))
    g.manual_seed(777)
    xs = torch.rand((nin_total, nring * B), generator=g, device=dev, dtype=torch.float32) * 2.0 - 1.0     # same audio on every rank
    xs = xs[col * nin:(col + 1) * nin].contiguous()                                                    # (grid: this rank's inputs)
    yb = torch.zeros((nout, B), device=dev, dtype=torch.float32) if grid else None                     # contiguous block for the all-reduce
    ys = torch.zeros((nout, nring * B), device=dev, dtype=torch.float32)

    file_irs = None
    if args.ir_file:
        from hisstools_library_amd.audiofile import load_impulse_responses
        data, file_rate = load_impulse_responses(args.ir_file)
        buf = torch.zeros((data.shape[0], L), device=dev, dtype=torch.float32)
        take = min(L, data.shape[1])
        buf[:, :take] = torch.from_numpy(data[:, :take]).to(dev)
        file_irs = buf

    def run(tail_ratio, steps, warmup, batched_block):
        """Build an engine (reference partitioning, or the extended far-tail ladder), load the synthetic IRs straight
        into HBM (decaying noise, unit L2 norm), reach steady state, then time `steps` process calls of B samples."""
        conv = H.Convolver(nin, nout, 0, device=local, maxBlock=max(B, batched_block), tailRatio=tail_ratio,
                           custom=(L, layout[0], layout[1], layout[2], layout[3], layout[4]))
        t_load = time.perf_counter()
        for o in range(nout):
            for i in range(nin):
                if file_irs is not None:
                    # real impulse responses (--ir-file): pair (i, o) takes channel (i * nout + o) mod channels, cut or
                    # zero-padded to the workload's IR length
                    h = file_irs[((col * nin + i) * nout + row * nout + o) % file_irs.shape[0]]
                else:
                    g.manual_seed(1000 * (col * nin + i) + (row * nout + o) + 1)
                    h = (torch.rand(L, generator=g, device=dev, dtype=torch.float32) * 2.0 - 1.0) * decay
                    h = h / torch.linalg.vector_norm(h)
                torch.cuda.synchronize()
                rc = conv.set_dev(i, o, h.data_ptr(), L, True)
                if rc != 0:
                    raise SystemExit(f"set_dev failed with ConvolveError {rc}")
        t_load = time.perf_counter() - t_load
        torch.cuda.synchronize()

        def step(k):
            off = 4 * (k % nring) * B
            if not grid:
                conv.process_dev(xs.data_ptr() + off, nring * B, ys.data_ptr() + off, nring * B, nin, nout, B)
                return
            # reduce path: this rank's partial block, then ONE all-reduce over the row group (the only exchange step)
            conv.process_dev(xs.data_ptr() + off, nring * B, yb.data_ptr(), B, nin, nout, B)
            conv.synchronize()
            dist.all_reduce(yb, op=dist.ReduceOp.SUM, group=row_group)

        # reach steady state first (every partition of every stage live, so the unpredicated kernel variant runs)
        for k in range(L // B + 2):
            step(k)
        for k in range(warmup):
            step(k)
        conv.synchronize()
        # a fresh box can run several times slower for its first seconds (clock / memory power states): keep stepping,
        # untimed, until two consecutive 20-step probes agree within 5 %
        prev = None
        for _ in range(12):
            tp = time.perf_counter()
            for k in range(20):
                step(k)
            conv.synchronize()
            tp = time.perf_counter() - tp
            settled = prev is not None and abs(tp - prev) <= 0.05 * prev
            if grid:
                # every step holds a collective: all ranks must leave the probe loop together
                flag = torch.tensor([1.0 if settled else 0.0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                settled = bool(flag.item() > 0.5)
            if settled:
                break
            prev = tp
        conv.clear_stats()
        conv.set_profiling(True)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            step(warmup + k)
        t_enq = time.perf_counter() - t0
        conv.synchronize()
        t_syn = time.perf_counter() - t0
        torch.cuda.synchronize()
        if os.environ.get("BENCH_DEBUG"):
            print(f"[bench debug] enqueue {1e3 * t_enq:.2f} ms, engine sync at {1e3 * t_syn:.2f} ms, torch sync at {1e3 * (time.perf_counter() - t0):.2f} ms",
                  file=sys.stderr)
        if world > 1:
            dist.barrier()
        tmax = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        stats = conv.stage_stats()
        conv.set_profiling(False)
        finite = bool(torch.isfinite(ys).all().item())

        # offline-style calls: one process() of `batched_block` samples spans several tail hops, so spectral_mac re-uses
        # every IR spectrum across the hops of the call (hop tiling) instead of re-reading it per hop
        batched = None
        if batched_block > B and not grid:
            ksteps = max(2, min(steps, 8))
            for _ in range(2):
                conv.process_dev(xs.data_ptr(), nring * B, ys.data_ptr(), nring * B, nin, nout, batched_block)
            conv.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for _ in range(ksteps):
                conv.process_dev(xs.data_ptr(), nring * B, ys.data_ptr(), nring * B, nin, nout, batched_block)
            conv.synchronize()
            torch.cuda.synchronize()
            tb = torch.tensor([time.perf_counter() - tb], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(tb, op=dist.ReduceOp.MAX)
            batched = {"block": batched_block, "steps": ksteps, "msamples_per_s": round(nout * world * batched_block * ksteps / float(tb.item()) / 1e6, 2)}
        del conv
        return float(tmax.item()), stats, finite, batched, t_load

    elapsed, stats, finite, batched, t_load = run(args.tail_ratio, args.steps, args.warmup, BB)

    # the same workload on the extended far-tail ladder (MI355X extension, not the reference's partitioning): reported
    # beside the headline, never as it
    extended = None
    if args.extended_ratio and not args.tail_ratio and not grid:
        try:
            e_el, e_stats, e_fin, _, _ = run(args.extended_ratio, args.steps, args.warmup, 0)
            extended = {"tail_ratio": args.extended_ratio, "stages": [(s_["fft_size"], s_["partitions"]) for s_ in e_stats],
                        "msamples_per_s": round(nout * world * B * args.steps / e_el / 1e6, 2), "ms_per_step": round(1e3 * e_el / args.steps, 4),
                        "finite_output": e_fin}
        except Exception as e:      # never let the side measurement take the headline down
            extended = {"error": str(e)}

    if rank == 0:
        total_out = nout * (world // 2 if grid else world)
        value = total_out * B * args.steps / elapsed / 1e6
        tail = stats[-1]
        Hh = tail["fft_size"] // 2
        launches = max(1, tail["mac_launches"])
        hops_per_launch = tail["mac_hops"] / launches
        alg_bytes = algorithmic_bytes_per_hop(Hh, tail["partitions"], nin, nout) * hops_per_launch
        avg_ms = tail["mac_ms"] / launches
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "Msamples/sec/node partitioned conv + achieved HBM GB/s vs peak",
            "value": round(value, 4),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not args.ir_file else "synthetic audio, impulse responses from " + os.path.basename(args.ir_file),
            "config": {
                "workload": f"{args.workload}: Convolver {nin}x{nout} per GPU ({nin_total}x{total_out} over {world} GPU), IR {L} samples @ {fs} Hz, "
                            f"stages {stages}, process block {B} samples, audio + spectra resident in HBM",
                "sharding": ("output rows per rank, no data-path collective" if not grid else
                             f"grid {world // 2} x 2: a row group's two ranks take half of the inputs each and sum their partial outputs with one "
                             f"all-reduce per step ({os.environ.get('BENCH_BACKEND', 'nccl')})"),
                "realtime_factor": round(B * args.steps / elapsed / fs, 3),
                "pair_msamples_per_s": round(value * nin_total, 2),
                "ir_load_s": round(t_load, 2),
                "finite_output": finite,
                "batched": batched,
                "extended_layout": extended,
                "tail_ratio": args.tail_ratio,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": f"spectral_mac (tail stage, FFT {tail['fft_size']}, P={tail['partitions']}, ksplit={tail['ksplit']}, out_tile={tail['out_tile']})",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "alg_bytes_per_launch": int(alg_bytes),
                "avg_launch_ms": round(avg_ms, 5),
                "launches": int(tail["mac_launches"]),
                "all_stage_mac_ms": {str(s["fft_size"]): round(s["mac_ms"], 3) for s in stats},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args.workload)
            except Exception as e:      # the baseline must never take the GPU number down with it
                line["cpu_baseline"] = {"value": None, "unit": "Msamples/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
            if not args.no_all_cores:
                try:
                    line["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args.workload)
                except Exception as e:
                    line["cpu_baseline_all_cores"] = {"value": None, "unit": "Msamples/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
