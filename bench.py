#!/usr/bin/env python3
"""bench.py — throughput + HBM roofline of the partitioned-convolution hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c5|c4|c3|c2|ns64] [--block B] [--scaling weak|strong]

One "step" = one Convolver::process call of B samples (default 8192 = one hop of the 16384-point tail stage) over
the whole channel matrix, audio and IR spectra resident in HBM.  Metric (BASELINE.json): output-channel
Msamples/s for the node, next to the achieved-vs-peak HBM bandwidth of the dominant kernel (the tail stage's spectral
multiply-accumulate launch: mac_meet_kernel of the n x m block, or spectral_mac_kernel where that block does not apply —
`roofline.kernel` names it).

Workloads (BASELINE.json configs; default c5 = the config the metric's HBM clause is quoted on):
    c5    Convolver 16x16, 60 s @ 96 kHz IRs (L = 5,760,000), zero latency      11.8 GB of tail spectra
    c4    Convolver 64x64, 2 s @ 48 kHz IRs  (L = 96,000),    zero latency
    c3    NToMono-shaped 8 -> 1, 5 s IRs     (L = 240,000),   zero latency
    c2    PartitionedConvolve-shaped 1x1, 10 s IR, one 4096-point stage (cache resident, launch bound)
    c1    MonoConvolve-shaped 1x1, 1 s IR, one 16384-point stage (the reference's own CPU-runnable case; launch bound)
    ns64  64x64, 10 s @ 48 kHz IRs (north-star target shape),  zero latency     15.7 GB of spectra

Inputs are SURVEY.md §8(d)'s generator: raw mt19937 draws, u = (r >> 8) * 2^-24; IR(in, out): seed 1000*in + out + 1,
h[k] = (2u - 1) * 10^(-3k/L) scaled to unit L2 norm; audio(ch): seed 777 + ch, x[n] = 2u - 1.  The draws are made on the
host (numpy's MT19937 seeded like std::mt19937), the float arithmetic of the IRs in float64 on the GPU; the audio is
bit-identical to the CPU leg's, the IRs to within one float32 ulp on a few samples in ten million (libm pow vs the GPU's).
After the timed region the run checks itself: the engine is reset and streams the CPU leg's sub-matrix inputs (the other
inputs silent) through the same engine, same spectra, same 8192-sample steps, and the output rows the CPU leg computed
with the unmodified reference are compared sample by sample (`config.self_check.max_rel_err`).

N > 1 (launched by torch.distributed.run, one rank per GPU):
    --scaling weak   (default) every rank owns its own block of `nout` output rows of an (N*nout) x nin system and receives
                     the same inputs — output-row sharding, no collective on the data path, per-GPU work fixed.
    --scaling strong the workload's matrix is FIXED (c4 = 64x64 as BASELINE config 4 states it) and split over the ranks:
                     output rows first (per GPU nin x nout/N, no collective); when there are fewer output rows than ranks
                     (c3: 8 -> 1) the inputs are split as well and the partial output blocks of a row group are summed
                     with one all-reduce per step (RCCL; the only exchange step the path has).
`--sharding grid` forces the input split on weak scaling too ((N/2) x 2 ranks).  BENCH_BACKEND=gloo lets several ranks
share one GPU to check the paths on a one-GPU box.

`config.also` — every BASELINE config under the same clock.  A run with no --workload / --scaling (what the driver calls) carries,
besides the headline, the digests (value, ms per step, roofline object, self-check error, one-core CPU figure) of further workloads:
    one GPU   child benches of ns64 (the 64x64 / 10 s @ 48 kHz shape north_star states its HBM target on; with paced real-time legs
              at 128 / 64 / 32 samples per call), c4 (the same legs), c3, c2 and c1;
    N > 1     in process, on the same ranks, STRONG-scaled: c4 — BASELINE config 4 as stated, the 64x64 matrix split over the ranks by
              output rows — and c3 — 8 -> 1 split by inputs, one all-reduce per step; each with a self-check in which every rank
              streams its block (collective included) and rank 0 compares with the reference CPU leg.
`--also ""` skips them, `--also c4,c2` picks others.  The headline itself is unchanged by them (N = 1: same engine, same steps).

The CPU baseline leg (rank 0, N = 1 only) times the UNMODIFIED reference (oracle/_ref, when the prebuilt library
travelled with the repo; else the C port) on a bounded sub-matrix of the same workload on one host core.
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
    "c1":   (1,  1,  48000,   48000, (False, 16384, 0, 0, 0)),           # BASELINE config 1: MonoConvolve(48000, false, 16384), 1 s IR (P = 6)
    "c2":   (1,  1,  480000,  48000, (False, 4096, 0, 0, 0)),
    "ns64": (64, 64, 480000,  48000, (True, 256, 1024, 4096, 16384)),
    "m16":  (16, 16, 96000,   48000, (True, 256, 1024, 4096, 16384)),   # mid-size: 16x16, 2 s IRs (not a BASELINE config)
    "m16l": (16, 16, 480000,  48000, (True, 256, 1024, 4096, 16384)),   # mid-size: 16x16, 10 s IRs (not a BASELINE config)
    "c4s8": (64, 8,  96000,   48000, (True, 256, 1024, 4096, 16384)),   # one rank's share of c4 strong-scaled over 8 GPUs (rows split): DESIGN section 6
    "c4g":  (32, 16, 96000,   48000, (True, 256, 1024, 4096, 16384)),   # ... the same in the 4 x 2 grid layout (4 row groups x 2 input groups): DESIGN section 6
    "c4s8l": (64, 8, 192000,  48000, (True, 256, 1024, 4096, 16384)),   # c4s8 with 4 s impulse responses (measurement aid: how the block's time scales)
}


# BENCH_IR_DIV=n (tests only): every workload's impulse responses n times shorter — the driver's exact multi-rank command on a one-GPU
# test box in bounded time.  The line says so (config.reduced_ir_div) and names the IR length it ran.
IR_DIV = max(1, int(os.environ.get("BENCH_IR_DIV", "1") or 1))
if IR_DIV > 1:
    WORKLOADS = {k: (v[0], v[1], max(20000, v[2] // IR_DIV), v[3], v[4]) for k, v in WORKLOADS.items()}


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


# ------------------------------------------------------------------------------------------- SURVEY §8(d) synthetic inputs

def mt_raw(seed, n):
    """n raw 32-bit draws of std::mt19937(seed) (numpy's MT19937 under the legacy init_genrand seeding), as uint32."""
    import numpy as np
    st = np.random.RandomState(seed).get_state()
    bg = np.random.MT19937()
    bg.state = {"bit_generator": "MT19937", "state": {"key": st[1], "pos": st[2]}}
    return bg.random_raw(n).astype(np.uint32)


def synth_audio(ch, n):
    """audio(ch): seed 777 + ch, x[n] = 2u - 1 — float32, bit-identical to the oracle's generator"""
    import numpy as np
    r = mt_raw(777 + ch, n)
    return (2.0 * ((r >> 8).astype(np.float64) * (1.0 / 16777216.0)) - 1.0).astype(np.float32)


class IrSynth:
    """IR(in, out): host threads draw the mt19937 words ahead of their use, the GPU finishes them in float64
    (2u - 1, times the 60 dB decay, unit L2 norm) and rounds once to float32."""

    def __init__(self, L, dev, pairs, workers=None):
        import concurrent.futures as cf
        import torch
        self.torch, self.L, self.dev = torch, L, dev
        self.decay = torch.pow(torch.tensor(10.0, device=dev, dtype=torch.float64), -3.0 * torch.arange(L, device=dev, dtype=torch.float64) / L)
        self.pairs = list(pairs)                                    # (in, out) in the order they will be asked for
        workers = workers or max(1, min(32, (os.cpu_count() or 2) - 1))
        self.pool = cf.ThreadPoolExecutor(max_workers=workers)
        self.ahead = max(2, min(2 * workers, (1 << 30) // max(1, 4 * L)))       # bounded look-ahead: at most ~1 GiB of words
        self.futures = {}
        self.next_submit = 0
        self._fill()

    def _fill(self):
        while self.next_submit < len(self.pairs) and len(self.futures) < self.ahead:
            i, o = self.pairs[self.next_submit]
            self.futures[(i, o)] = self.pool.submit(mt_raw, 1000 * i + o + 1, self.L)
            self.next_submit += 1

    def get(self, i, o):
        torch = self.torch
        fut = self.futures.pop((i, o), None)
        words = fut.result() if fut is not None else mt_raw(1000 * i + o + 1, self.L)
        self._fill()
        r = torch.from_numpy(words.view("int32")).to(self.dev).to(torch.int64) & 0xFFFFFFFF
        t = (2.0 * ((r >> 8).to(torch.float64) * (1.0 / 16777216.0)) - 1.0) * self.decay
        return (t * torch.rsqrt(torch.sum(t * t))).to(torch.float32)

    def close(self):
        self.pool.shutdown(wait=False, cancel_futures=True)


# ------------------------------------------------------------------------------------------- CPU legs (the only users of oracle/)

def cpu_sub_matrix(workload):
    """The bounded sub-matrix of a workload the one-core CPU leg (and therefore the self-check) runs on."""
    nin, nout, L, fs, layout = WORKLOADS[workload]
    tail, p_tail = stage_layout(L, layout)[-1]
    max_pairs = max(1, min(32, 3000 // max(1, p_tail)))
    sub_in = min(nin, 8)
    while sub_in > 1 and sub_in > max_pairs:
        sub_in //= 2
    sub_out = max(1, min(nout, max_pairs // sub_in))
    return sub_in, sub_out


def cpu_baseline(workload, hops=64, seg_hops=16, protocol=True):
    """Reference CPU path on one host core, on a bounded sub-matrix of the same workload (steady state: the stream is first run
    for as many hops as the tail has partitions so every partition is live, then timed) — BASELINE.md section 3's protocol:
    best of 3 timed segments, process blocks of 512 and 2048 samples, the -O3 -msse2 build (the reference's fixed 4-wide SIMD;
    `value` = its 512-sample figure) and, where it travelled, the -O3 -mavx2 -mfma build standing in for -march=native (the
    sources cannot be compiled on the GPU box).  `_outs` (popped by the caller before printing) is what the first build computed
    over the warm-up and the first timed segment: [sub_out][warm + S] — the self-check's reference."""
    import numpy as np
    from oracle import oracle as O

    nin, nout, L, fs, layout = WORKLOADS[workload]
    kind = "reference" if O.have_ref() else "port"
    tail, p_tail = stage_layout(L, layout)[-1]
    sub_in, sub_out = cpu_sub_matrix(workload)
    hop = tail // 2
    warm, S, S2 = p_tail * hop, hops * hop, seg_hops * hop
    n_total = warm + S + 5 * S2
    xs = np.stack([O.synth_audio(i, n_total) for i in range(sub_in)])
    mono = workload in ("c2", "c1")
    irs = {(i, o): O.synth_ir(i, o, L) for o in range(1 if mono else sub_out) for i in range(1 if mono else sub_in)}

    def stream(engine, x, block):
        """(outs, seconds) of one segment"""
        if mono:
            t0 = time.perf_counter()
            y = engine.run(x[0], block)                 # (a Python loop over the calls: ~3 us of ctypes per call beside >= 60 us of work)
            return y[None, :], time.perf_counter() - t0
        return engine.stream_timed(np.ascontiguousarray(x), sub_out, block)

    def one_build(backend, keep_outs):
        t_set = time.perf_counter()
        if mono:
            e = O.PartitionedConvolve(layout[1], L, 0, 0, backend=backend)
            e.setResetOffset(0)
            e.set(irs[(0, 0)])
        else:
            e = O.Convolver(sub_in, sub_out, 0, backend=backend)
            for (i, o), h in irs.items():
                e.set(i, o, h, True)
        t_set = time.perf_counter() - t_set
        pos = 0
        y0, _ = stream(e, xs[:, :warm], 512)
        pos = warm
        rates = {512: [], 2048: []}
        outs, secs0 = None, None
        plan = [(512, S)] + ([(512, S2), (512, S2), (2048, S2), (2048, S2), (2048, S2)] if protocol else [])
        for block, n in plan:
            y, secs = stream(e, xs[:, pos:pos + n], block)
            if outs is None:
                outs, secs0 = (np.concatenate([y0, y], axis=1) if keep_outs else None), secs
            pos += n
            rates[block].append((1 if mono else sub_in * sub_out) * n / secs)
        return {b_: max(v) for b_, v in rates.items() if v}, outs, secs0, t_set

    best, outs, secs, t_set = one_build("ref" if kind == "reference" else "port", True)
    wide = None
    if protocol and kind == "reference" and O.have_ref_wide():
        try:
            wide = one_build("ref_wide", False)[0]
        except Exception:
            wide = None
    pair_rate = best[512]                                          # pair-samples / s on one core
    to_value = lambda r: None if r is None else round(r / nin / 1e6, 6)     # == output-channel Msamples/s for the full matrix
    return {
        "value": to_value(pair_rate), "unit": "Msamples/s", "cores": 1, "kind": kind,
        "flags": "-O3 -msse2", "block": 512, "best_of": 3 if protocol else 1,
        "b2048": to_value(best.get(2048)),
        "wide_flags": None if wide is None else "-O3 -mavx2 -mfma (stands in for -march=native: the sources cannot be compiled on this host)",
        "wide_b512": None if wide is None else to_value(wide.get(512)), "wide_b2048": None if wide is None else to_value(wide.get(2048)),
        "sample": f"{sub_in}x{sub_out} sub-matrix of the {nin}x{nout} workload, same {L}-sample IRs, after a {warm}-sample warm-up (all partitions live): "
                  f"best of 3 timed segments ({S}, {S2}, {S2} samples) in 512-sample calls, then 3 x {S2} samples in 2048-sample calls, 1 thread; value = "
                  f"pair-samples/s / {nin} inputs = the whole-matrix output rate one core would sustain; IR load took {t_set:.1f} s",
        "pair_msamples_per_s": round(pair_rate / 1e6, 4),
        "seconds": round(secs, 3),
        "_outs": outs, "_sub": (sub_in, sub_out, warm + S),
    }


def cpu_baseline_all_cores(workload, max_threads=64, hops=8):
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
    # the leg is bounded to about ten seconds: what it costs is the warm-up (P hops at a growing partition count — the reference
    # only multiplies partitions that have seen input, PartitionedConvolve.cpp:285,322), so long tails stream ONE pair per thread
    # and time eight hops; the per-pair work (IR length, partitioning, call size) is the workload's
    sub_in = min(nin, 1 if p_tail > 100 else 8)
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


# ------------------------------------------------------------------------------------------- sharding plan

def split_range(n, parts, index):
    """Contiguous balanced split of range(n) into `parts`; [lo, hi) of block `index`."""
    base, rem = divmod(n, parts)
    lo = index * base + min(index, rem)
    return lo, lo + base + (1 if index < rem else 0)


def shard_plan(nin, nout, world, rank, scaling, sharding):
    """Which block of the (global) matrix this rank convolves.  Returns a dict: the global matrix size, this rank's
    [in_lo, in_hi) x [out_lo, out_hi), the grid (row groups x column groups) and its position in it."""
    if scaling == "strong":
        if sharding == "grid":
            # (N/2) row groups x 2 input groups: half the forward transforms per rank, one all-reduce of the row group's block per step
            if world < 2 or world % 2 or world // 2 > nout or nin < 2:
                raise SystemExit(f"cannot split a {nin}x{nout} matrix over {world} ranks as a (N/2) x 2 grid")
            go = world // 2
        else:
            go = min(world, nout)
            while world % go:
                go -= 1
        gi = world // go
        if gi > nin:
            raise SystemExit(f"cannot split a {nin}x{nout} matrix over {world} ranks")
        row, col = divmod(rank, gi)
        o_lo, o_hi = split_range(nout, go, row)
        i_lo, i_hi = split_range(nin, gi, col)
        return {"nin_total": nin, "nout_total": nout, "go": go, "gi": gi, "row": row, "col": col, "in": (i_lo, i_hi), "out": (o_lo, o_hi)}
    # weak: every row group brings its own `nout` rows
    gi = 2 if sharding == "grid" else 1
    go = world // gi
    row, col = divmod(rank, gi)
    i_lo, i_hi = split_range(nin, gi, col)
    return {"nin_total": nin, "nout_total": nout * go, "go": go, "gi": gi, "row": row, "col": col, "in": (i_lo, i_hi), "out": (row * nout, (row + 1) * nout)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS), help="default c5 (the config the metric's HBM clause is quoted on)")
    ap.add_argument("--block", type=int, default=8192)
    ap.add_argument("--batched-block", type=int, default=65536, help="also time offline-style calls of this many samples (0 = skip)")
    ap.add_argument("--offline-hops", type=int, default=64,
                    help="also time OFFLINE calls (B = S): an engine made for calls of this many tail hops, one process() call per step, its output "
                         "checked against hop-sized calls of the timed engine on the same input (0 = skip; matrices with two outputs or more)")
    ap.add_argument("--tail-ratio", type=int, default=0, help="run the HEADLINE on the extended far-tail ladder (0 = reference partitioning)")
    ap.add_argument("--extended-ratio", type=int, default=8, help="also measure the extended far-tail ladder with this ratio (0 = skip)")
    ap.add_argument("--ir-file", default="", help="WAVE / AIFF / AIFC file with real impulse responses instead of the synthetic ones")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="weak (default): every rank brings its own nout output rows (per-GPU work fixed).  strong: the workload's matrix is fixed and "
                         "split over the ranks — output rows first, inputs too when there are fewer rows than ranks (then one all-reduce per step)")
    ap.add_argument("--sharding", default="rows", choices=["rows", "grid"],
                    help="weak scaling only.  rows: every rank owns output rows and all inputs, no data-path collective (default).  grid: (N/2) x 2 "
                         "ranks — the two ranks of a row group each convolve half of the inputs and sum their partial outputs with one RCCL "
                         "all-reduce per step (SURVEY 8e: the reduce path; needs an even N >= 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-all-cores", action="store_true", help="skip the all-host-cores CPU leg")
    ap.add_argument("--no-self-check", action="store_true", help="skip the comparison with the CPU leg's output after the timed region")
    ap.add_argument("--realtime-block", type=int, default=128,
                    help="also measure paced real-time calls of this many samples through the host-pointer and device-pointer entry points (0 = skip)")
    ap.add_argument("--realtime-extra", default="", help="further paced call sizes, comma separated (e.g. 64,32): p50 / p99 / max against their budgets")
    ap.add_argument("--also", default=None,
                    help="further workloads after the headline, comma separated, each with its own engine, timed region, roofline object and "
                         "self-check against the CPU reference; their digests go to config.also.  One GPU: child processes; default (headline "
                         "= the default workload) ns64,c4,c3,c2,c1 — the north-star shape and every other BASELINE config.  N > 1 ranks: run "
                         "in process by all ranks with STRONG scaling; default c4,c3 — BASELINE config 4 as stated (64x64 split over the "
                         "ranks by output rows) and config 3's input split with one all-reduce per step.  '' = none")
    ap.add_argument("--leg", action="store_true", help=argparse.SUPPRESS)       # a child bench of an `also` leg: a digest-sized run
    args = ap.parse_args(argv)
    args.default_run = args.workload is None and args.scaling is None and not args.ir_file and not args.tail_ratio and args.sharding == "rows"
    if args.workload is None:
        args.workload = "c5"
    if args.scaling is None:
        args.scaling = "weak"
    return args


def main():
    args = parse_args()
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
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    if args.sharding == "grid" and (world < 2 or world % 2):
        raise SystemExit("--sharding grid needs an even number of ranks (>= 2)")
    ctx = {"world": world, "rank": rank, "local": local, "dev": dev, "backend": backend}

    also = args.also
    if also is None:
        also = ("ns64,c4,c3,c2,c1" if world == 1 else "c4,c3") if args.default_run else ""
    also = [w for w in also.split(",") if w and (w != args.workload or world > 1)]

    line = bench_line(args, ctx)

    if also and not args.ir_file:
        digests = []
        if world == 1:
            for w in also:
                digests.append(also_leg(w, local))
        else:
            # N > 1: the further workloads run IN PROCESS on the same ranks and process group, STRONG-scaled — the workload's own
            # matrix split over the ranks (c4: 64x64 by output rows, no exchange; c3: 8 -> 1 by inputs, one all-reduce per step).
            # They must never cost the run its headline: should a rank fail or a collective hang, a watchdog on every rank ends the
            # process after BENCH_ALSO_TIMEOUT seconds (default 300), rank 0 printing the headline line with the error noted.
            import threading

            def give_up():
                if rank == 0 and line is not None:
                    line["config"]["also"] = digests + [{"workload": "legs", "scaling": "strong", "error": "strong-scaled legs did not finish in time; headline unaffected"}]
                    print(json.dumps(emit(line)), flush=True)
                os._exit(0)

            dog = threading.Timer(float(os.environ.get("BENCH_ALSO_TIMEOUT", "300")), give_up)
            dog.daemon = True
            dog.start()
            for w in also:
                digests.append(strong_leg(w, args, ctx))
                if w == "c4" and world % 2 == 0 and not os.environ.get("BENCH_NO_C4_GRID"):
                    # config 4 strong-scaled in BOTH layouts SURVEY 8e names: output rows over the N ranks (every rank transforms all 64 inputs,
                    # no exchange) and the (N/2) x 2 grid (half the transforms per rank, one all-reduce of a row group's block per step); the
                    # line quotes the faster one, and its efficiency against the whole matrix on ONE of these GPUs, measured in this run
                    grid = strong_leg(w, args, ctx, sharding="grid")
                    one = None
                    if rank == 0:
                        one = also_leg("c4", local, steps=min(args.steps, 40), warmup=min(args.warmup, 5), rt=False, timeout=240)
                    dist.barrier()
                    if rank == 0:
                        rows = digests[-1]
                        best, name = rows, f"rows {world} x 1"
                        if grid and not grid.get("error") and (rows.get("error") or (grid.get("value") or 0) > (rows.get("value") or 0)):
                            best, name = grid, f"grid {world // 2} x 2"
                        strong_c4 = {"layout": name, "msamples_per_s": best.get("value"), "ms_per_step": best.get("ms_per_step"),
                                     "rows_msamples_per_s": rows.get("value"), "grid_msamples_per_s": (grid or {}).get("value"),
                                     "grid_error": (grid or {}).get("error"), "grid_max_rel_err": ((grid or {}).get("self_check") or {}).get("max_rel_err"),
                                     "one_gpu_msamples_per_s": (one or {}).get("value"), "one_gpu_error": (one or {}).get("error")}
                        if strong_c4["msamples_per_s"] and strong_c4["one_gpu_msamples_per_s"]:
                            strong_c4["efficiency"] = round(strong_c4["msamples_per_s"] / (world * strong_c4["one_gpu_msamples_per_s"]), 4)
                        line["config"]["c4_strong"] = strong_c4
                        if grid:
                            grid["workload"] = "c4grid" + str(grid.get("workload", ""))[2:]       # (its own key among the digests)
                            digests.append(grid)
            dog.cancel()
        if rank == 0 and line is not None:
            line["config"]["also"] = digests
    if rank == 0 and line is not None:
        # (a child bench of an `also` leg hands its parent the rich line; the driver's line is the short one)
        print(json.dumps(line if args.leg else emit(line)), flush=True)
    if world > 1:
        dist.destroy_process_group()


def _leg_key(d):
    """'ns64' of a digest whose workload text starts with 'ns64: ...' (+ '_strong' for the strong-scaled legs of an N > 1 run)"""
    w = str(d.get("workload", "")).split(":")[0].strip() or "leg"
    return w + ("_strong" if d.get("scaling") == "strong" else "")


def flat_scalars(line):
    """Everything a reader needs to check every shape, as flat scalars (the driver's record keeps scalar config keys only)"""
    cfg, out = line["config"], {}
    rt = cfg.get("realtime") or {}
    if "host_pointers" in rt:
        out.update({"rt128_host_p50_ms": rt["host_pointers"].get("p50_ms"), "rt128_host_p99_ms": rt["host_pointers"].get("p99_ms"),
                    "rt128_over_budget": rt["host_pointers"].get("over_budget"), "rt128_dev_p99_ms": (rt.get("device_pointers") or {}).get("p99_ms")})
    if cfg.get("batched"):
        out["batched_block"] = cfg["batched"].get("block")
        out["batched_msamples_per_s"] = cfg["batched"].get("msamples_per_s")
        out["batched_mac_hbm_frac"] = (line.get("roofline_batched") or {}).get("hbm_frac")
    of = cfg.get("offline") or {}
    if of:
        out.update({"offline_block": of.get("block"), "offline_msamples_per_s": of.get("msamples_per_s"), "offline_bound": of.get("bound"),
                    "offline_frac": of.get("frac"), "offline_max_rel_err": of.get("max_rel_err_vs_hop_calls"), "offline_error": of.get("error")})
    ex = cfg.get("extended_layout") or {}
    if ex:
        out.update({"extended_tail_ratio": ex.get("tail_ratio"), "extended_msamples_per_s": ex.get("msamples_per_s"), "extended_ms_per_step": ex.get("ms_per_step"),
                    "extended_step_frac": (ex.get("roofline_step") or {}).get("frac"), "extended_step_traffic": (ex.get("roofline_step") or {}).get("traffic"),
                    "extended_max_rel_err": (ex.get("self_check") or {}).get("max_rel_err"),
                    "extended_error": ex.get("error")})
    sc4 = cfg.get("c4_strong")
    for d in cfg.get("also") or []:
        if not d:
            continue
        k = _leg_key(d)
        if k == "c4grid_strong":
            continue                    # (the grid layout's digest is in the side file; its scalars are above)
        if d.get("error"):
            out[k + "_error"] = str(d["error"])[:160]
            continue
        rf, sc, cb = d.get("roofline") or {}, d.get("self_check") or {}, d.get("cpu_baseline") or {}
        # (the line must stay under 4 KB: the b2048 / wide-build CPU figures, the 64-sample leg and the realtime factors of the further
        # workloads are in the side file)
        out.update({k + "_msamples_per_s": d.get("value"), k + "_ms_per_step": d.get("ms_per_step"),
                    k + "_bound": rf.get("bound"), k + "_mac_frac": rf.get("frac") if rf.get("bound") == "hbm" else None,
                    k + "_mac_ms": rf.get("avg_launch_ms"), k + "_max_rel_err": sc.get("max_rel_err"), k + "_self_check_ok": sc.get("ok"),
                    k + "_cpu_1core": None if cb.get("value") is None else round(cb["value"], 4)})
        if rf.get("bound") == "hbm":
            out[k + "_whole_step_frac"] = rf.get("whole_step_frac")
        else:
            out[k + "_kernel"] = str(rf.get("kernel", "")).split(" ")[0]
            out[k + "_kernel_share_of_step"] = rf.get("kernel_share_of_step")
        if rf.get("note_ceiling"):
            out[k + "_note"] = f"above SURVEY 8d's {rf.get('survey_8d_ceiling_msamples_per_s')} ceiling: whole-hop calls fold the stages' sum P into lead + tail partitions"
        ex = d.get("extended_layout") or {}
        if ex and not ex.get("error"):
            out[k + "_extended_msamples_per_s"] = ex.get("msamples_per_s")
            out[k + "_extended_max_rel_err"] = (ex.get("self_check") or {}).get("max_rel_err")
        of = d.get("offline") or {}
        if of and not of.get("error"):
            out[k + "_offline_msamples_per_s"] = of.get("msamples_per_s")
            out[k + "_offline_bound_frac"] = f"{of.get('bound')} {of.get('frac')}"
        elif of:
            out[k + "_offline_error"] = str(of.get("error"))[:100]
        sb = (d.get("realtime") or {}).get("small_blocks") or {}
        hp = (d.get("realtime") or {}).get("host_pointers") or {}
        if hp:
            out[k + "_rt128_p99_ms"] = hp.get("p99_ms")
        if "32" in sb:
            out[f"{k}_rt32_p50_ms"] = sb["32"].get("p50_ms")
            out[f"{k}_rt32_p99_ms"] = sb["32"].get("p99_ms")
            out[f"{k}_rt32_over_budget"] = sb["32"].get("over_budget")
    if isinstance(sc4, dict):
        # config 4 strong-scaled over the run's GPUs: the faster of the two layouts, its efficiency against ONE of these GPUs running the whole matrix
        out.update({"c4_strong_layout": sc4.get("layout"), "c4_strong_msamples_per_s": sc4.get("msamples_per_s"), "c4_strong_efficiency": sc4.get("efficiency"),
                    "c4_strong_rows_msamples_per_s": sc4.get("rows_msamples_per_s"), "c4_strong_grid_msamples_per_s": sc4.get("grid_msamples_per_s"),
                    "c4_strong_grid_max_rel_err": sc4.get("grid_max_rel_err"), "c4_one_gpu_msamples_per_s": sc4.get("one_gpu_msamples_per_s"),
                    "c4_strong_grid_error": None if not sc4.get("grid_error") else str(sc4.get("grid_error"))[:120]})
    return out


def emit(line):
    """The ONE line printed: short (< 4 KB) and flat, so that it survives a scalars-only parser and an 8 KB output tail.  The rich
    record (digests of every further workload, the real-time, batched and extended-ladder objects, the samples' descriptions) goes to
    a side file beside it: BENCH_DETAILS (default gpurun_out/bench_details.json)."""
    import copy
    path = os.environ.get("BENCH_DETAILS", os.path.join(ROOT, "gpurun_out", "bench_details.json"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(line, f, indent=1)
    except Exception as e:
        path = f"not written: {e}"
    short = copy.deepcopy(line)
    cfg = short["config"]
    flat = flat_scalars(line)
    sc = cfg.get("self_check") or {}
    keep = {k: cfg.get(k) for k in ("workload", "sharding", "realtime_factor", "pair_msamples_per_s", "ir_load_s", "finite_output", "max_rel_err", "tail_ratio",
                                    "reduced_ir_div")}
    keep["inputs"] = "SURVEY 8d generator (mt19937)"
    keep["self_check_ok"] = sc.get("ok")
    keep["self_check_against"] = None if not sc else str(sc.get("against", "")).split(":")[0]
    keep.update({k: v for k, v in flat.items() if v is not None})         # (absent = not applicable: a launch-bound leg has no mac_frac)
    if isinstance(cfg.get("c4_strong"), dict) and keep.get("c4_strong_layout", "").startswith("grid"):
        keep["c4_strong_ms_per_step"] = cfg["c4_strong"].get("ms_per_step")      # (the keys above quote the faster layout)
    keep["workload"] = str(keep.get("workload", "")).replace(", audio + spectra resident in HBM", "").replace("process block", "block")
    keep["details_file"] = path
    for k in [k for k, v in keep.items() if v is None or (k.endswith("_bound") and k != "offline_bound")]:      # (a leg's bound shows in what it carries: mac_frac or kernel)
        if k not in ("max_rel_err",):
            keep.pop(k)
    short["config"] = keep
    rf = short.get("roofline") or {}
    short["roofline"] = {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_commit", "alg_bytes_per_launch", "avg_launch_ms",
                                                 "launches", "steady_launches", "box_read_GBps", "achieved_over_box_read", "whole_step_frac",
                                                 "survey_8d_ceiling_msamples_per_s", "kernel_share_of_step", "avg_launch_source") if k in rf}
    if rf.get("traffic") is not None:
        short["roofline"]["traffic_source"] = "static: profiles/traffic_<workload>.json (rocprofv3 --pmc), not measured in this run"
    for key in ("cpu_baseline", "cpu_baseline_all_cores"):
        cb = short.get(key)
        if cb:
            short[key] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "flags", "block", "best_of", "b2048", "wide_b512", "wide_b2048") if k in cb}
            short[key]["sample"] = str(cb.get("sample", ""))[:110]
    short.pop("roofline_batched", None)
    return short


def strong_leg(workload, args, ctx, steps=40, warmup=5, sharding="rows"):
    """N > 1: one more workload on the ranks the headline ran on, strong-scaled, with a self-check in which EVERY rank streams its
    share (the collective of an input-split layout included) and rank 0 compares with the reference CPU leg.  Returns the digest
    (rank 0) or None."""
    import copy
    a = copy.copy(args)
    a.workload, a.scaling, a.sharding = workload, "strong", sharding
    a.steps, a.warmup = min(args.steps, steps), min(args.warmup, warmup)
    a.batched_block, a.extended_ratio, a.realtime_block, a.realtime_extra, a.tail_ratio, a.offline_hops = 0, 0, 0, "", 0, 0
    a.no_all_cores, a.no_cpu_baseline, a.leg = True, False, True         # (the bounded CPU leg is this leg's checker; --no-self-check skips both)
    if args.no_self_check:
        a.no_cpu_baseline = True
    t0 = time.perf_counter()
    try:
        d = bench_line(a, ctx)
    except SystemExit as e:         # e.g. a matrix that cannot be split over this many ranks
        return {"workload": workload, "scaling": "strong", "error": str(e)}
    except Exception as e:          # (the other ranks may now wait for this one in a collective: the watchdog in main() ends that)
        return {"workload": workload, "scaling": "strong", "error": f"{type(e).__name__}: {e}"}
    if d is None:
        return None
    out = digest_of(d)
    out.pop("cpu_baseline", None)       # (every rank computed the bounded reference at once: a checker here, not a clean one-core timing)
    out["scaling"] = "strong"
    out["sharding"] = d["config"].get("sharding")
    out["seconds"] = round(time.perf_counter() - t0, 1)
    return out


def digest_of(d):
    """What config.also keeps of a bench line"""
    rf, sc = d.get("roofline", {}), d.get("config", {}).get("self_check") or {}
    out = {
        "workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"], "n_gpus": d["n_gpus"], "steps": d["steps"], "warmup": d["warmup"],
        "ms_per_step": d["ms_per_step"], "realtime_factor": d["config"].get("realtime_factor"),
        "roofline": {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "alg_bytes_per_launch", "avg_launch_ms",
                                            "launches", "steady_launches", "traffic", "traffic_commit", "traffic_kernel", "traffic_source", "profiled_ms_per_step", "box_read_GBps",
                                            "achieved_over_box_read", "box_copy_GBps", "avg_launch_source", "kernel_share_of_step", "whole_step_frac",
                                            "note_ceiling", "survey_8d_ceiling_msamples_per_s") if k in rf},
        "self_check": {k: sc.get(k) for k in ("max_rel_err", "tolerance", "ok", "against", "mac_steady_launches", "error") if k in sc},
        "cpu_baseline": {k: (d.get("cpu_baseline") or {}).get(k) for k in ("value", "unit", "cores", "kind", "flags", "block", "best_of", "b2048",
                                                                           "wide_b512", "wide_b2048")},
    }
    rt = d.get("config", {}).get("realtime")
    if rt:
        out["realtime"] = rt
    ex = d.get("config", {}).get("extended_layout")
    if ex:
        out["extended_layout"] = ex
    of = d.get("config", {}).get("offline")
    if of:
        out["offline"] = of
    return out


def bench_line(args, ctx):
    """One workload on the ranks of `ctx`: engine, inputs, timed region, roofline object, CPU legs, self-check.  Returns the JSON
    line as a dict on rank 0, None elsewhere."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import hisstools_library_amd as H

    world, rank, local, dev, backend = ctx["world"], ctx["rank"], ctx["local"], ctx["dev"], ctx["backend"]
    nin_w, nout_w, L, fs, layout = WORKLOADS[args.workload]
    B = args.block
    plan = shard_plan(nin_w, nout_w, world, rank, args.scaling, args.sharding)
    (in_lo, in_hi), (out_lo, out_hi) = plan["in"], plan["out"]
    nin, nout = in_hi - in_lo, out_hi - out_lo                     # this rank's block
    nin_total, nout_total = plan["nin_total"], plan["nout_total"]
    reduce_path = plan["gi"] > 1
    row_group = None
    if reduce_path:
        for r in range(plan["go"]):                                # every rank creates every group
            grp = dist.new_group(ranks=[r * plan["gi"] + c for c in range(plan["gi"])])
            if r == plan["row"]:
                row_group = grp
    stages = stage_layout(L, layout)

    BB = max(args.batched_block, 0)
    nring = max(8, -(-BB // B))
    xs = torch.from_numpy(np.stack([synth_audio(i, nring * B) for i in range(in_lo, in_hi)])).to(dev)   # (every rank: the same audio)
    yb = torch.zeros((nout, B), device=dev, dtype=torch.float32) if reduce_path else None           # contiguous block for the all-reduce
    ys = torch.zeros((nout, nring * B), device=dev, dtype=torch.float32)

    file_irs = None
    if args.ir_file:
        from hisstools_library_amd.audiofile import load_impulse_responses
        data, file_rate = load_impulse_responses(args.ir_file)
        buf = torch.zeros((data.shape[0], L), device=dev, dtype=torch.float32)
        take = min(L, data.shape[1])
        buf[:, :take] = torch.from_numpy(data[:, :take]).to(dev)
        file_irs = buf

    rccl_direct = reduce_path and backend == "nccl" and not os.environ.get("BENCH_TORCH_ALLREDUCE")

    def build(max_block, tail_ratio):
        """An engine for calls of up to `max_block` samples with the workload's synthetic IRs (decaying noise, unit L2 norm) in HBM"""
        conv = H.Convolver(nin, nout, 0, device=local, maxBlock=max_block, tailRatio=tail_ratio,
                           custom=(L, layout[0], layout[1], layout[2], layout[3], layout[4]))
        t_load = time.perf_counter()
        synth = None if file_irs is not None else IrSynth(L, dev, [(in_lo + i, out_lo + o) for o in range(nout) for i in range(nin)])
        for o in range(nout):
            for i in range(nin):
                if file_irs is not None:
                    # real impulse responses (--ir-file): pair (i, o) takes channel (i * nout + o) mod channels, cut or
                    # zero-padded to the workload's IR length
                    h = file_irs[((in_lo + i) * nout_total + out_lo + o) % file_irs.shape[0]]
                else:
                    h = synth.get(in_lo + i, out_lo + o)
                torch.cuda.synchronize()
                rc = conv.set_dev(i, o, h.data_ptr(), L, True)
                if rc != 0:
                    raise SystemExit(f"set_dev failed with ConvolveError {rc}")
        if synth is not None:
            synth.close()
        return conv, time.perf_counter() - t_load

    def run(tail_ratio, steps, warmup, batched_block, keep=False):
        """Build an engine (reference partitioning, or the extended far-tail ladder), load the synthetic IRs (decaying
        noise, unit L2 norm) into HBM, reach steady state, then time `steps` process calls of B samples."""
        conv, t_load = build(max(B, batched_block), tail_ratio)
        torch.cuda.synchronize()
        # reduce path over RCCL: the library enqueues ncclAllReduce on the engine's own stream behind the block — no host
        # synchronisation between convolution and collective, so the exchange overlaps the next block's FFTs and MAC.  The
        # communicator of a row group is made from a unique id its first rank draws (distributed through torch's store).
        if rccl_direct:
            ids = [None] * world
            dist.all_gather_object(ids, H.rccl_unique_id() if plan["col"] == 0 else None)
            conv.comm_init(ids[plan["row"] * plan["gi"]], plan["col"], plan["gi"])

        def step(k):
            off = 4 * (k % nring) * B
            if not reduce_path:
                conv.process_dev(xs.data_ptr() + off, nring * B, ys.data_ptr() + off, nring * B, nin, nout, B)
                return
            # reduce path: this rank's partial block, then ONE all-reduce over the row group (the only exchange step)
            if rccl_direct:
                conv.process_dev_allreduce(xs.data_ptr() + off, nring * B, yb.data_ptr(), B, nin, nout, B)
                return
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
            if world > 1:
                # (the reduce path holds a collective per step: all ranks must leave the probe loop together)
                flag = torch.tensor([1.0 if settled else 0.0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                settled = bool(flag.item() > 0.5)
            if settled:
                break
            prev = tp

        def timed(profiled):
            """`steps` steps between barriers and device syncs; max over ranks of the wall time"""
            conv.clear_stats()
            conv.set_profiling(profiled)
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
            t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return t

        # HBM-bound workloads (the headline): the HIP events around every spectral_mac launch are recorded IN the timed region — two
        # event records against milliseconds of kernel.  A launch-bound block (c1, c2, c3: four kernels of 5-13 us) is slowed by
        # its own instrumentation (two more packets per block in a chain that is nothing but packet latency), so there the timed
        # region runs bare and the same `steps` steps are repeated with the events on for the roofline object.
        tail_fft, tail_p = stages[-1]
        # (... and the engines the library itself runs as serial chains — last-stage spectra below 1 GiB, hcv_engine_block.hip kSerialMB —
        # whose one-hop block may be the two meeting launches of the n x m fused block: one rank's share of strong-scaled config 4)
        launch_bound = 8 * (tail_fft // 2) * tail_p * nin * nout < (1024 << 20) and tail_ratio == args.tail_ratio
        if launch_bound:
            tmax = timed(False)
            stats = conv.stage_stats()
            if not (stats[-1].get("fused_launches") and stats[-1].get("out_tile", 1) == 1):
                # (a block of several launches: the multiply-accumulate's own time from a second pass with its HIP events on — the n x m
                # block's multiply-accumulate launch included.  A one-output block that IS one launch has no such figure: with the events on
                # the engine takes the separate kernels)
                timing_note["profiled_ms_per_step"] = round(1e3 * float(timed(True).item()) / steps, 4)
                bare = stats
                stats = conv.stage_stats()
                for s_new, s_old in zip(stats, bare):
                    s_new["fused_launches"] = s_old.get("fused_launches", 0)
        else:
            tmax = timed(True)
            stats = conv.stage_stats()
        conv.set_profiling(False)
        finite = bool(torch.isfinite(yb if reduce_path else ys).all().item())

        # offline-style calls: one process() of `batched_block` samples spans several tail hops, so spectral_mac re-uses
        # every IR spectrum across the hops of the call (hop tiling) instead of re-reading it per hop
        batched = None
        if batched_block > B and not reduce_path:
            ksteps = max(2, min(steps, 8))
            for _ in range(2):
                conv.process_dev(xs.data_ptr(), nring * B, ys.data_ptr(), nring * B, nin, nout, batched_block)
            conv.synchronize()
            conv.clear_stats()
            conv.set_profiling(True)
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
            bstats = conv.stage_stats()
            conv.set_profiling(False)
            batched = {"block": batched_block, "steps": ksteps,
                       "msamples_per_s": round(nout_total * batched_block * ksteps / float(tb.item()) / 1e6, 2), "_stats": bstats}
        if not keep:
            del conv
            conv = None
        return float(tmax.item()), stats, finite, batched, t_load, conv

    timing_note = {}
    elapsed, stats, finite, batched, t_load, conv = run(args.tail_ratio, args.steps, args.warmup, BB, keep=True)

    # ---- CPU legs (rank 0, one GPU): the unmodified reference on the box's host cores; the one-core leg's output is the
    # reference of the self-check below.  (A strong-scaled leg of an N > 1 run — `--leg` — checks itself too: there EVERY rank
    # computes the bounded reference, a second or two, instead of idling in a collective while rank 0 does.)
    cpu, cpu_all = None, None
    check_all_ranks = world > 1 and args.leg and args.scaling == "strong" and not args.no_self_check
    if ((rank == 0 and world == 1) or check_all_ranks) and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(args.workload, hops=16 if check_all_ranks else 64, protocol=not check_all_ranks)
        except Exception as e:      # the baseline must never take the GPU number down with it
            cpu = {"value": None, "unit": "Msamples/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}

    # ---- self-check: the same engine, reset, streams the CPU leg's inputs (the other inputs silent) in the same B-sample
    # steps as the timed region — through the ramp-up (partition bounds checked) into the steady state (the timed kernel
    # instantiation) — and the CPU leg's output rows are compared sample by sample.  On N > 1 ranks every rank streams its own
    # block of the matrix (its share of the CPU leg's inputs; on the reduce path the all-reduce of every step included) and
    # rank 0 compares the rows it holds.
    have_ref = cpu is not None and cpu.get("_outs") is not None
    if world > 1 and check_all_ranks:
        flag = torch.tensor([1.0 if have_ref else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        have_ref = bool(flag.item() > 0.5)

    def check_against_cpu_leg(conv):
        """reset `conv`, stream the CPU leg's inputs in B-sample steps, compare with the CPU leg's rows"""
        try:
            sub_in, sub_out, S_chk = cpu["_sub"]
            ref = cpu["_outs"]
            S_run = -(-S_chk // B) * B
            xc = torch.zeros((max(nin, 1), S_run), device=dev, dtype=torch.float32)
            for i in range(in_lo, min(in_hi, sub_in)):
                xc[i - in_lo, :S_chk] = torch.from_numpy(synth_audio(i, S_chk)).to(dev)
            yc = torch.zeros((nout, S_run), device=dev, dtype=torch.float32)
            conv.reset()
            conv.clear_stats()
            for pos in range(0, S_run, B):
                if not reduce_path:
                    conv.process_dev(xc.data_ptr() + 4 * pos, S_run, yc.data_ptr() + 4 * pos, S_run, nin, nout, B)
                elif rccl_direct:
                    conv.process_dev_allreduce(xc.data_ptr() + 4 * pos, S_run, yc.data_ptr() + 4 * pos, S_run, nin, nout, B)
                else:
                    conv.process_dev(xc.data_ptr() + 4 * pos, S_run, yb.data_ptr(), B, nin, nout, B)
                    conv.synchronize()
                    dist.all_reduce(yb, op=dist.ReduceOp.SUM, group=row_group)
                    yc[:, pos:pos + B] = yb
                    torch.cuda.synchronize()
            conv.synchronize()
            rows = [o for o in range(sub_out) if out_lo <= o < out_hi]        # (rank 0 holds the first rows of the matrix)
            got = yc[[o - out_lo for o in rows], :S_chk].cpu().numpy().astype(np.float64) if rows else np.zeros((0, S_chk))
            st_chk = conv.stage_stats()[-1]
            errs = [float(np.abs(got[k] - ref[o]).max() / np.abs(ref[o]).max()) for k, o in enumerate(rows)] or [float("nan")]
            n_hops_ref = 16 if check_all_ranks else 64
            tail_span = slice(S_chk - n_hops_ref * (stages[-1][0] // 2), S_chk)       # the span the CPU leg timed: every partition live
            err_tail = max([float(np.abs(got[k][tail_span] - ref[o][tail_span]).max() / np.abs(ref[o]).max()) for k, o in enumerate(rows)] or [float("nan")])
            return {"max_rel_err": float(f"{max(errs):.3e}"), "max_rel_err_steady_span": float(f"{err_tail:.3e}"),
                    "against": f"{cpu['kind']} CPU leg: output rows {rows[0] if rows else '-'}..{rows[-1] if rows else '-'} from inputs 0..{sub_in - 1} (the other "
                               f"inputs silent), {S_chk} samples in {B}-sample steps after a reset, on the timed engine(s) and spectra"
                               + (f"; every one of the {world} ranks streamed its block" + (", all-reduce per step included" if reduce_path else "")
                                  if world > 1 else ""),
                    "tolerance": 1e-5, "ok": bool(max(errs) <= 1e-5),
                    "mac_launches": int(st_chk["mac_launches"]), "mac_steady_launches": int(st_chk["mac_steady_launches"])}
        except Exception as e:
            return {"max_rel_err": None, "error": str(e)}

    self_check = None
    if have_ref and not args.no_self_check and not args.tail_ratio:
        self_check = check_against_cpu_leg(conv)

    # ---- paced real-time calls (what a plug-in host does): `realtime-block` samples per call at the workload's sample rate,
    # through the host-pointer entry point (hcv_convolver_process_f32: what the C++ drop-in calls) and the device-pointer one
    realtime = None
    if args.realtime_block and world == 1 and not args.tail_ratio:
        try:
            extra = [int(v) for v in args.realtime_extra.split(",") if v.strip()]
            realtime = realtime_leg(conv, np, torch, dev, nin, nout, fs, args.realtime_block, stages, extra_blocks=extra)
        except Exception as e:
            realtime = {"error": str(e)}

    # ---- offline calls (B = S, SURVEY section 7 "streaming vs batched", 8d "report at B = S"; PartitionedConvolve.cpp:298-299, 321-348 loops
    # the hops of any numSamples): an engine made for calls of `offline_hops` tail hops.  One such call re-uses every IR spectrum over all
    # its hops, and the multiply-accumulate becomes a dense contraction per bin on the f32 matrix cores (hcv_mac_mfma.hip) — its bound is
    # the f32 MFMA / VALU peak (157.3 TFLOP/s), or HBM where the IRs are short.  Checked against the timed engine fed the same samples in
    # hop-sized calls.
    offline = None
    if args.offline_hops and world == 1 and not reduce_path and not args.tail_ratio and nout >= 2:
        try:
            offline = offline_leg(conv, build, np, torch, dev, nin, nout, in_lo, stages, args.offline_hops, B, args.steps)
        except Exception as e:
            offline = {"error": f"{type(e).__name__}: {e}"}
    del conv

    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_all_cores:
        try:
            cpu_all = cpu_baseline_all_cores(args.workload)
        except Exception as e:
            cpu_all = {"value": None, "unit": "Msamples/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}

    # the same workload on the extended far-tail ladder (MI355X extension, not the reference's partitioning): reported
    # beside the headline, never as it
    extended = None
    if args.extended_ratio and not args.tail_ratio and not reduce_path:
        try:
            # (timed over at least 128 steps: the ladder's last rung turns over once in 64 steps of config 5 and its streams run well ahead of
            # one another, so K = 20 steps between two device syncs — 2.4 ms — is mostly fill and drain: 0.117 - 0.158 ms per step from run
            # to run on one box where 128 steps and 512 steps both give 0.119 - 0.120, profiles/r05_queue_probe.txt)
            e_steps = max(args.steps, 128)
            e_el, e_stats, e_fin, _, _, e_conv = run(args.extended_ratio, e_steps, args.warmup, 0, keep=True)
            extended = {"tail_ratio": args.extended_ratio, "stages": [(s_["fft_size"], s_["partitions"]) for s_ in e_stats], "steps": e_steps,
                        "msamples_per_s": round(nout_total * B * e_steps / e_el / 1e6, 2), "ms_per_step": round(1e3 * e_el / e_steps, 4),
                        "finite_output": e_fin}
            # the same self-check the headline has: the ladder's engine, reset, against the reference CPU leg's rows (the SAME
            # convolution on a different partitioning).  Its steps are not an HBM test — its spectra are read 8 x less often per rung
            # — so no roofline fraction is quoted for it: the per-stage multiply-accumulate times say where its step goes
            if have_ref and not args.no_self_check and world == 1:
                extended["self_check"] = check_against_cpu_leg(e_conv)
            extended["all_stage_mac_ms_per_step"] = {str(s_["fft_size"]): round(s_["mac_ms"] / e_steps, 4) for s_ in e_stats}
            # SURVEY 8d's per-sample-time figure summed over the ladder's stages (8 Nin (Nout + 1) sum P + 12 Nin stages + 4 Nout stages)
            # against the WHOLE step: no single kernel dominates this leg (profiles/r03_c5_extended_kernel_summary.txt)
            e_sum_p, e_ns = sum(s_["partitions"] for s_ in e_stats), len(e_stats)
            e_bytes = (8.0 * nin * (nout + 1) * e_sum_p + 12.0 * nin * e_ns + 4.0 * nout * e_ns) * B
            e_gbs = e_bytes / (e_el / e_steps) / 1e9
            extended["roofline_step"] = {"bound": "hbm", "achieved": round(e_gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(e_gbs / 8000.0, 4),
                                         "alg_bytes_per_step": int(e_bytes), "sum_partitions": int(e_sum_p),
                                         "note": "whole step (every stage's kernels, transforms included) against SURVEY 8d's bytes for this ladder"}
            try:
                tl = json.load(open(os.path.join(ROOT, "profiles", f"traffic_{args.workload}_ladder.json")))
                extended["roofline_step"]["traffic"] = tl.get("hbm_bytes_per_step")
                extended["roofline_step"]["traffic_source"] = (f"static: profiles/traffic_{args.workload}_ladder.json (rocprofv3 --pmc passes of bench.py --tail-ratio "
                                                                f"{args.extended_ratio} on one stream, tools/pmc_ladder.sh), not measured in this run")
            except Exception:
                extended["roofline_step"]["traffic"] = None
            extended["note"] = ("MI355X extension, not the reference's partitioning: reported beside the headline, never as it; parity at this scale: "
                                "tests/test_steady_state_gpu.py::test_config5_extended_ladder_full_depth")
            del e_conv
        except Exception as e:      # never let the side measurement take the headline down
            extended = {"error": str(e)}

    if rank == 0:
        value = nout_total * B * args.steps / elapsed / 1e6
        tail = stats[-1]
        Hh = tail["fft_size"] // 2
        launches = max(1, tail["mac_launches"])
        hops_per_launch = tail["mac_hops"] / launches
        # (a whole-hop launch reduces over the stage's own partitions plus the one holding the IR in front of its segment —
        #  the work of every shorter stage and the head, SURVEY 8d's sum over stages — : launch_partitions)
        parts = max(tail["partitions"], tail.get("launch_partitions", 0))
        alg_bytes = algorithmic_bytes_per_hop(Hh, parts, nin, nout) * hops_per_launch
        avg_ms = tail["mac_ms"] / launches
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # a working set that fits the 256 MiB Infinity Cache is not an HBM test (SURVEY §8d on c2): the step is bound by its
        # launches, and the "achieved" rate is cache bandwidth
        live_bytes = 8 * Hh * parts * nin * nout
        bound = "hbm" if live_bytes > (256 << 20) else "launch"
        traffic, traffic_commit, traffic_kernel = None, None, None
        tpath = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
        if os.path.exists(tpath) and world == 1:
            try:
                trec = json.load(open(tpath))
                traffic, traffic_commit, traffic_kernel = trec.get("hbm_bytes_per_launch"), trec.get("commit"), trec.get("kernel")
            except Exception:
                traffic = None
        roofline_batched = None
        if batched is not None:
            bst = batched.pop("_stats")[-1]
            bl = max(1, bst["mac_launches"])
            bh = bst["mac_hops"] / bl
            ot = max(1, bst["out_tile"])
            # one launch covers `bh` hops: H is read once per hop tile (ceil(bh / hop_tile) times), X once per output tile
            tiles = -(-int(round(bh)) // max(1, bst["hop_tile"]))
            bp = max(bst["partitions"], bst.get("launch_partitions", 0))
            b_bytes = (8 * Hh * bp * nin * nout * tiles + 8 * Hh * (bp + bh) * nin * (-(-nout // ot))
                       + 8 * Hh * nout * bh * max(1, bst["ksplit"]))
            b_flops = 8.0 * Hh * bp * nin * nout * bh
            b_ms = bst["mac_ms"] / bl
            roofline_batched = {
                "kernel": f"spectral_mac (tail stage, hop tile {bst['hop_tile']}, out tile {ot}, ksplit {bst['ksplit']}), {batched['block']}-sample calls",
                "bytes_per_launch": int(b_bytes), "flops_per_launch": int(b_flops), "avg_launch_ms": round(b_ms, 5),
                "hbm_gbs": round(b_bytes / (b_ms * 1e-3) / 1e9, 1) if b_ms > 0 else None,
                "hbm_frac": round(b_bytes / (b_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if b_ms > 0 else None,
                "tflops": round(b_flops / (b_ms * 1e-3) / 1e12, 2) if b_ms > 0 else None,
                "f32_valu_frac": round(b_flops / (b_ms * 1e-3) / 1e12 / 157.3, 4) if b_ms > 0 else None,
            }
        for d in (batched,):
            if d is not None:
                d.pop("_stats", None)
        sharding_txt = ("output rows per rank, no data-path collective" if not reduce_path else
                        f"grid {plan['go']} x {plan['gi']}: the {plan['gi']} ranks of a row group take a share of the inputs each and sum their partial "
                        f"outputs with one all-reduce per step ({'RCCL ncclAllReduce on the engine stream' if backend == 'nccl' and not os.environ.get('BENCH_TORCH_ALLREDUCE') else backend + ' through torch.distributed after a host sync'})")
        line = {
            "metric": "Msamples/sec/node partitioned conv + achieved HBM GB/s vs peak",
            "value": round(value, 4),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not args.ir_file else "synthetic audio, impulse responses from " + os.path.basename(args.ir_file),
            "config": {
                "workload": f"{args.workload}: Convolver {nin}x{nout} per GPU ({nin_total}x{nout_total} over {world} GPU), IR {L} samples @ {fs} Hz, "
                            f"stages {stages}, process block {B} samples, audio + spectra resident in HBM",
                "inputs": "SURVEY 8d generator (mt19937; IR seed 1000*in+out+1, 60 dB decaying noise of unit norm; audio seed 777+ch)"
                          if not args.ir_file else "audio: SURVEY 8d generator",
                "sharding": sharding_txt,
                "realtime_factor": round(B * args.steps / elapsed / fs, 3),
                "pair_msamples_per_s": round(value * nin_total, 2),
                "ir_load_s": round(t_load, 2),
                "finite_output": finite,
                "self_check": self_check,
                "max_rel_err": None if self_check is None else self_check.get("max_rel_err"),
                "batched": batched,
                "offline": offline,
                "realtime": realtime,
                "extended_layout": extended,
                "tail_ratio": args.tail_ratio,
                "reduced_ir_div": IR_DIV if IR_DIV > 1 else None,
            },
            "roofline": {
                "bound": bound,
                "kernel": (f"mac_meet_kernel (n x m block: tail stage, FFT {tail['fft_size']}, P={parts}, {tail['ksplit']} partial spectra per output, out_tile=8)"
                           if tail.get("fused_launches") and tail.get("out_tile") == 8 else
                           f"spectral_mac (tail stage, FFT {tail['fft_size']}, P={parts}, ksplit={tail['ksplit']}, out_tile={tail['out_tile']})"),
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_commit": traffic_commit,
                "traffic_kernel": traffic_kernel,
                "traffic_source": None if traffic is None else f"static: profiles/traffic_{args.workload}.json (rocprofv3 --pmc passes of an earlier run of this "
                                                               f"command, tools/pmc_traffic.sh), not measured in this run",
                "alg_bytes_per_launch": int(alg_bytes),
                "avg_launch_ms": round(avg_ms, 5),
                "launches": int(tail["mac_launches"]),
                "steady_launches": int(tail["mac_steady_launches"]),
                "all_stage_mac_ms": {str(s["fft_size"]): round(s["mac_ms"], 3) for s in stats},
            },
        }
        if bound == "launch":
            fused = int(tail.get("fused_launches", 0)) if tail.get("out_tile", 1) == 1 else 0
            if fused:
                # the whole block — forward transforms, multiply-accumulate, inverse — is ONE launch (hcv_fft_split.hip): that kernel is what
                # ran, and its duration cannot be had from inside the run without lengthening the chain; profiles/kernel_us.json holds the
                # rocprofv3 --kernel-trace average of this command (tools/r04_profiles.sh), quoted only while it is consistent with the step
                hops_blk = max(1, B // Hh)
                kname = "fused_block_hops_kernel" if hops_blk > 1 else ("fused_block_1x1_kernel" if nin == 1 and parts <= 16 else "fused_block_nx1_kernel")
                k_us, k_src = None, None
                try:
                    rec = json.load(open(os.path.join(ROOT, "profiles", "kernel_us.json"))).get(args.workload)
                    if rec and rec.get("kernel", "").startswith(kname):
                        k_us, k_src = float(rec["us"]), rec.get("source")
                except Exception:
                    pass
                step_ms = 1e3 * elapsed / args.steps
                if k_us is not None and k_us * 1e-3 > step_ms:
                    k_us, k_src = None, None            # (another box, another clock: never quote a kernel longer than the step it sits in)
                line["roofline"].update({
                    "kernel": f"{kname} (one launch per {B}-sample block: {nin} forward transform(s), multiply-accumulate over P={parts}, inverse; FFT {tail['fft_size']})",
                    "launches": fused, "avg_launch_ms": None if k_us is None else round(k_us * 1e-3, 5), "avg_launch_source": k_src,
                    "kernel_share_of_step": None if k_us is None else round(k_us * 1e-3 / step_ms, 3),
                    "achieved": None if k_us is None else round(alg_bytes / (k_us * 1e-6) / 1e9, 1),
                    "frac": None if k_us is None else round(alg_bytes / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)})
                line["roofline"].pop("all_stage_mac_ms", None)
            line["roofline"]["note"] = (f"{live_bytes / 1048576.0:.1f} MiB of live spectra stay in the 256 MiB Infinity Cache: the step is bound by its "
                                        f"kernel launches, not by HBM; `achieved` is cache bandwidth and `frac` is not an HBM fraction")
            if "profiled_ms_per_step" in timing_note:
                line["roofline"]["note"] += (f"; avg_launch_ms comes from a second pass of the same {args.steps} steps with the HIP events on "
                                             f"({timing_note['profiled_ms_per_step']} ms per step there: the event records lengthen a launch-bound "
                                             f"chain), value / ms_per_step from the bare pass")
                line["roofline"]["profiled_ms_per_step"] = timing_note["profiled_ms_per_step"]
        # SURVEY 8d's ceiling for the reference's stage list (every stage's partitions read once per hop) beside what the engine's
        # whole-hop step actually moves (one lead partition in place of the head and the shorter stages)
        sum_p, n_st = sum(p_ for _, p_ in stages), len(stages)
        bytes_per_sample = 8.0 * nin * (nout + 1) * sum_p + 12.0 * nin * n_st + 4.0 * nout * n_st
        ceiling = HBM_PEAK_GBS * 1e9 / bytes_per_sample * nout / 1e6
        line["roofline"]["survey_8d_ceiling_msamples_per_s"] = round(ceiling * world, 1)
        if bound == "hbm" and hops_per_launch > 0:
            line["roofline"]["whole_step_frac"] = round(alg_bytes / hops_per_launch * (B / Hh) / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4)
            if value / world > ceiling:
                line["roofline"]["note_ceiling"] = (f"value exceeds SURVEY 8d's {ceiling * world:.0f} Msamples/s ceiling: that figure reads every stage's partitions "
                                                    f"per hop (sum P = {sum_p}); a call made of whole tail hops is ONE uniform convolution over the lead slot and "
                                                    f"the tail's own partitions (P = {parts}), so fewer bytes move for the same output (parity: self_check)")
        if bound == "hbm":
            try:
                copy_gbs = box_copy_rate(dev)
                line["roofline"]["box_copy_GBps"] = round(copy_gbs, 1)
                line["roofline"]["achieved_over_box_copy"] = round(achieved / copy_gbs, 4)
                import ctypes
                rd = ctypes.c_double(0.0)
                aid = ctypes.CDLL(os.path.join(ROOT, "tools", "benchaid", "libhcv_benchaid.so"))      # (a bench aid beside the product, built by build())
                aid.hcv_benchaid_box_read_rate.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
                if aid.hcv_benchaid_box_read_rate(local, 4 << 30, 5, ctypes.byref(rd)) == 0 and rd.value > 0:
                    # (a read-only stream with the kernel's own kind of loads: the ceiling this box has for the multiply-accumulate)
                    line["roofline"]["box_read_GBps"] = round(rd.value, 1)
                    line["roofline"]["achieved_over_box_read"] = round(achieved / rd.value, 4)
                line["roofline"]["box_copy_note"] = ("a 2 GiB device-to-device copy (torch) on this box right after the timed region, read + write bytes over its "
                                                     "time: the box's own streaming yardstick (MI355X_MICROARCH.md: ~6.3 TB/s achievable).  The multiply-accumulate "
                                                     "is almost read-only traffic with nontemporal loads and out-runs a copy; `frac` stays achieved / 8 TB/s")
            except Exception as e:          # (context, never the measurement)
                line["roofline"]["box_copy_GBps"] = f"error: {e}"
        if args.tail_ratio:
            # the extended ladder as the headline workload (--tail-ratio): no single kernel dominates its step — the last rung's launch turns over
            # once in 64 steps and may not run at all inside the timed region — so the roofline object is the WHOLE step against SURVEY 8d's bytes
            # for the ladder's own stage list (what config.extended_layout.roofline_step carries when the ladder runs beside the headline)
            l_sum_p, l_ns = sum(s_["partitions"] for s_ in stats), len(stats)
            l_bytes = (8.0 * nin * (nout + 1) * l_sum_p + 12.0 * nin * l_ns + 4.0 * nout * l_ns) * B
            l_gbs = l_bytes / (elapsed / args.steps) / 1e9
            line["roofline"] = {"bound": "hbm", "kernel": f"whole step of the ladder (stages {[(s_['fft_size'], s_['partitions']) for s_ in stats]}): no single kernel dominates it",
                                "achieved": round(l_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(l_gbs / HBM_PEAK_GBS, 4), "traffic": None,
                                "alg_bytes_per_step": int(l_bytes), "sum_partitions": int(l_sum_p),
                                "all_stage_mac_ms": {str(s_["fft_size"]): round(s_["mac_ms"], 3) for s_ in stats},
                                "note": "bytes = SURVEY 8d's per-sample-time figure summed over the ladder's stages x the step's samples; traffic: profiles/traffic_<workload>_ladder.json"}
        if roofline_batched is not None:
            line["roofline_batched"] = roofline_batched
        if cpu is not None:
            cpu.pop("_outs", None)
            cpu.pop("_sub", None)
            line["cpu_baseline"] = cpu
        if cpu_all is not None:
            line["cpu_baseline_all_cores"] = cpu_all
        return line
    return None


def box_copy_rate(dev, gib=2.0, reps=5):
    """What THIS box streams when it does nothing else: a device-to-device copy of `gib` GiB (read + write bytes over the best of `reps`
    timings, HIP events).  Boxes of the pool differ by up to 8 % in what their HBM delivers; the roofline object carries this figure so
    that `frac` (against the 8 TB/s peak) can be read beside the box's own ceiling.  MI355X_MICROARCH.md quotes ~6.3 TB/s for it."""
    import torch
    n = int(gib * (1 << 30)) // 4
    x = torch.empty(n, dtype=torch.float32, device=dev).fill_(1.0)
    y = torch.empty_like(x)
    y.copy_(x)
    torch.cuda.synchronize(dev)
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y.copy_(x)
        e1.record()
        torch.cuda.synchronize(dev)
        best = min(best, e0.elapsed_time(e1))
    del x, y
    return 2.0 * n * 4 / (best * 1e-3) / 1e9


def also_leg(workload, device, steps=40, warmup=5, timeout=420, rt=True):
    """A further workload after the headline, in a child process (its own engine, inputs, timed region and self-check against the
    reference CPU leg): the digest of the child's bench line.  The two 64x64 shapes also run the paced real-time legs (128-, 64-
    and 32-sample calls).  Never takes the headline down."""
    rt = ["--realtime-block", "128", "--realtime-extra", "64,32"] if rt and workload in ("ns64", "c4") else ["--realtime-block", "0"]
    # (the north-star shape also on the extended ladder: what an unchanged caller of the reference API gets for it, hcv_api.hip's rule)
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", workload, "--steps", str(steps), "--warmup", str(warmup), "--also", "", "--leg",
           "--no-all-cores", "--batched-block", "0", "--extended-ratio", "4" if workload == "ns64" else "0",
           "--offline-hops", "64" if workload in ("ns64", "c4") else "0"] + rt
    env = dict(os.environ, LOCAL_RANK=str(device), RANK="0", WORLD_SIZE="1")     # the same GPU as the headline
    t0 = time.perf_counter()
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        rows = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if out.returncode != 0 or not rows:
            return {"workload": workload, "error": f"child bench rc {out.returncode}: {out.stderr[-300:]}"}
        d = json.loads(rows[-1])
    except Exception as e:
        return {"workload": workload, "error": str(e)}
    dg = digest_of(d)
    dg["seconds"] = round(time.perf_counter() - t0, 1)
    return dg


def offline_leg(conv, build, np, torch, dev, nin, nout, in_lo, stages, hops, B, steps):
    """Offline calls: a second engine whose blocks are `hops` tail hops long (maxBlock), one process_dev call per step.  The engine and
    `conv` (the timed engine, reset) are fed the same samples from silence — the one in `hops`-hop calls, the other in B-sample calls —
    until every partition of the tail is live and the offline engine's multiply-accumulate takes its steady-state (matrix-core)
    instantiation; the last call's outputs are compared.  Then `ksteps` calls are timed with the launch's HIP events on."""
    F_PEAK = 157.3          # f32 MFMA = f32 VALU peak, TFLOP/s (MI355X_MICROARCH.md)
    tail_fft, tail_p = stages[-1]
    Hh = tail_fft // 2
    OB = hops * Hh
    ocv, t_load = build(OB, 0)
    nblk = -(-(tail_p + 3) // hops) + 1                 # calls until the last one runs with every partition live
    xo = torch.from_numpy(np.stack([synth_audio(in_lo + i, 2 * OB) for i in range(nin)])).to(dev)
    yo = torch.zeros((nout, 2 * OB), device=dev, dtype=torch.float32)
    yh = torch.zeros((nout, 2 * OB), device=dev, dtype=torch.float32)
    conv.reset()
    for k in range(nblk):
        off = (k & 1) * OB
        ocv.process_dev(xo.data_ptr() + 4 * off, 2 * OB, yo.data_ptr() + 4 * off, 2 * OB, nin, nout, OB)
        for pos in range(off, off + OB, B):
            conv.process_dev(xo.data_ptr() + 4 * pos, 2 * OB, yh.data_ptr() + 4 * pos, 2 * OB, nin, nout, B)
    ocv.synchronize()
    conv.synchronize()
    last = slice(((nblk - 1) & 1) * OB, ((nblk - 1) & 1) * OB + OB)
    a, b = yo[:, last].double(), yh[:, last].double()
    err = float(((a - b).abs().amax(dim=1) / b.abs().amax(dim=1).clamp_min(1e-30)).max().item())
    ksteps = max(3, min(steps, 6))
    ocv.clear_stats()
    ocv.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(ksteps):
        off = (k & 1) * OB
        ocv.process_dev(xo.data_ptr() + 4 * off, 2 * OB, yo.data_ptr() + 4 * off, 2 * OB, nin, nout, OB)
    ocv.synchronize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / ksteps
    st = ocv.stage_stats()[-1]
    ocv.set_profiling(False)
    del ocv
    launches = max(1, st["mac_launches"])
    parts = max(st["partitions"], st.get("launch_partitions", 0))
    mac_ms = st["mac_ms"] / launches
    flops = 8.0 * Hh * parts * nin * nout * hops
    tiles = -(-hops // max(1, st["hop_tile"]))
    ot = max(1, st["out_tile"])
    nbytes = (8.0 * Hh * parts * nin * nout * tiles + 8.0 * Hh * (parts + hops) * nin * (-(-nout // ot)) + 8.0 * Hh * nout * hops * max(1, st["ksplit"]))
    f_frac = flops / (mac_ms * 1e-3) / 1e12 / F_PEAK if mac_ms > 0 else 0.0
    h_frac = nbytes / (mac_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if mac_ms > 0 else 0.0
    bound = "mfma" if f_frac >= h_frac else "hbm"
    return {"block": OB, "hops": hops, "steps": ksteps, "msamples_per_s": round(nout * OB / dt / 1e6, 1), "ms_per_call": round(1e3 * dt, 3),
            "bound": bound, "frac": round(max(f_frac, h_frac), 4),
            "kernel": f"spectral_mac_mfma_kernel (tail stage, FFT {tail_fft}, P={parts}, hop tile {st['hop_tile']}, out tile {ot}, ksplit {st['ksplit']})"
                      if st["hop_tile"] >= 32 else f"spectral_mac (hop tile {st['hop_tile']}, out tile {ot}, ksplit {st['ksplit']})",
            "mac_ms_per_launch": round(mac_ms, 4), "mac_tflops": round(flops / (mac_ms * 1e-3) / 1e12, 1) if mac_ms > 0 else None,
            "mac_f32_peak_frac": round(f_frac, 4), "mac_hbm_frac": round(h_frac, 4), "mac_share_of_call": round(mac_ms / (1e3 * dt), 3),
            "peaks": {"f32_mfma_tflops": F_PEAK, "hbm_gbs": HBM_PEAK_GBS}, "ir_load_s": round(t_load, 2),
            "max_rel_err_vs_hop_calls": float(f"{err:.3e}"), "tolerance": 1e-5, "ok": bool(err <= 1e-5),
            "checked": f"last of {nblk} calls from silence (every partition live) against {B}-sample calls of the timed engine on the same samples"}


def realtime_leg(conv, np, torch, dev, nin, nout, fs, RB, stages, seconds=1.0, extra_blocks=()):
    """Paced real-time calls on the engine the headline ran on (every partition live): call k is issued no earlier than
    k * RB / fs.  `host`: hcv_convolver_process_f32 (host pointers in, host pointers out — what HISSTools::Convolver::process
    does); `device`: hcv_convolver_process_f32_dev with sync (HBM-resident audio).  Milliseconds per call."""
    import ctypes as C
    import hisstools_library_amd as H
    from hisstools_library_amd._lib import f32p
    L = H.load()
    ncalls = max(64, int(seconds * fs / RB))
    rng = np.random.RandomState(11)
    xin = rng.uniform(-1, 1, size=(nin, RB)).astype(np.float32)
    yout = np.zeros((nout, RB), np.float32)
    ip = (f32p * nin)(*[xin[i].ctypes.data_as(f32p) for i in range(nin)])
    op = (f32p * nout)(*[yout[o].ctypes.data_as(f32p) for o in range(nout)])
    xd, yd = torch.from_numpy(xin).to(dev), torch.zeros((nout, RB), device=dev)
    budget = 1e3 * RB / fs
    out = {"block": RB, "budget_ms": round(budget, 4), "calls": ncalls}

    def paced(call):
        ts = np.zeros(ncalls)
        for _ in range(8):
            call()
        t_start = time.perf_counter()
        for k in range(ncalls):
            while time.perf_counter() < t_start + k * RB / fs:
                pass
            t0 = time.perf_counter()
            call()
            ts[k] = time.perf_counter() - t0
        ts *= 1e3
        return {"p50_ms": round(float(np.percentile(ts, 50)), 4), "p99_ms": round(float(np.percentile(ts, 99)), 4),
                "max_ms": round(float(ts.max()), 4), "realtime_factor": round(float(ncalls * budget / ts.sum()), 2),
                "over_budget": int((ts > budget).sum())}

    def host_call():
        if L.hcv_convolver_process_f32(conv.h, ip, op, nin, nout, RB) != 0:
            raise RuntimeError("process_f32 failed")

    def dev_call():
        conv.process_dev(xd.data_ptr(), RB, yd.data_ptr(), RB, nin, nout, RB, sync=True)

    out["host_pointers"] = paced(host_call)
    out["device_pointers"] = paced(dev_call)
    out["finite"] = bool(np.isfinite(yout).all() and torch.isfinite(yd).all().item())
    fed = 2 * (8 + ncalls) * RB
    # smaller hosts (64- and 32-sample callbacks: 1.33 / 0.67 ms budgets at 48 kHz), host pointers only — what such a host calls
    for EB in extra_blocks:
        if EB <= 0 or EB > RB:
            continue
        # (1.2 s: at 32 samples per call 1800 calls, so that p99 is the 18th slowest call and not one of the dozen around the 16384-point
        # stage's hops — with 600 calls the figure jumped between 0.07 and 0.26 ms from run to run)
        n_e = max(64, int(1.2 * fs / EB))
        ts = np.zeros(n_e)
        t_start = time.perf_counter()
        for k in range(n_e):
            while time.perf_counter() < t_start + k * EB / fs:
                pass
            t0 = time.perf_counter()
            if L.hcv_convolver_process_f32(conv.h, ip, op, nin, nout, EB) != 0:
                raise RuntimeError("process_f32 failed")
            ts[k] = time.perf_counter() - t0
        ts *= 1e3
        b_e = 1e3 * EB / fs
        out.setdefault("small_blocks", {})[str(EB)] = {
            "budget_ms": round(b_e, 4), "calls": n_e, "p50_ms": round(float(np.percentile(ts, 50)), 4), "p99_ms": round(float(np.percentile(ts, 99)), 4),
            "max_ms": round(float(ts.max()), 4), "over_budget": int((ts > b_e).sum())}
        fed += n_e * EB
    # the headline's step through HOST pointers (what every existing caller of HISSTools::Convolver::process does): one
    # synchronous call of `hop` samples per step — pinned staging copies, PCIe both ways and the wait included.  Never `value`.
    hop = stages[-1][0] // 2
    xh = rng.uniform(-1, 1, size=(nin, hop)).astype(np.float32)
    yh = np.zeros((nout, hop), np.float32)
    ih = (f32p * nin)(*[xh[i].ctypes.data_as(f32p) for i in range(nin)])
    oh = (f32p * nout)(*[yh[o].ctypes.data_as(f32p) for o in range(nout)])
    nsteps = 24
    # the paced legs fed 2 x (8 + ncalls) calls of RB samples: bring the stream back to a hop boundary, so that these steps are
    # the headline's (whole, aligned hops) and not a run of unaligned 8192-sample calls through every stage
    pad = (-fed) % hop
    if pad and L.hcv_convolver_process_f32(conv.h, ih, oh, nin, nout, pad) != 0:
        raise RuntimeError("process_f32 failed")

    def host_steps():
        # (the paced legs above leave the GPU mostly idle and its clocks low: step for a quarter of a second first)
        t_warm = time.perf_counter()
        while time.perf_counter() - t_warm < 0.25:
            if L.hcv_convolver_process_f32(conv.h, ih, oh, nin, nout, hop) != 0:
                raise RuntimeError("process_f32 failed")
        t0 = time.perf_counter()
        for _ in range(nsteps):
            if L.hcv_convolver_process_f32(conv.h, ih, oh, nin, nout, hop) != 0:
                raise RuntimeError("process_f32 failed")
        return (time.perf_counter() - t0) / nsteps

    dt = host_steps()
    out["host_pointer_steps"] = {"block": hop, "ms_per_step": round(1e3 * dt, 4), "msamples_per_s": round(nout * hop / dt / 1e6, 2),
                                 "note": "synchronous host-buffer calls, PCIe-inclusive; reported beside the HBM-resident value, never as it"}
    # the same with the caller's buffers registered once (hcv_host_register): the kernels work on them in place, no staging copies
    try:
        H.host_register(xh)
        H.host_register(yh)
        dt = host_steps()
        out["host_pointer_steps"]["registered_ms_per_step"] = round(1e3 * dt, 4)
        out["host_pointer_steps"]["registered_msamples_per_s"] = round(nout * hop / dt / 1e6, 2)
    finally:
        H.host_unregister(xh)
        H.host_unregister(yh)
    return out


if __name__ == "__main__":
    main()
