#!/usr/bin/env python3
"""N independent config-2-shaped convolvers on one GPU: tools/parallel_c2.py [channels ...].

BASELINE config 2 is ONE 1x1 convolver (a single 4096-point stage, 10 s IR): its block is a chain of four latency-bound launches,
and the all-host-cores CPU leg of the bench (64 independent reference convolvers, one per thread) out-runs that one chain on a
per-box basis ("What's weak" 4).  The like-for-like figure for the box is the same 64 independent convolvers on the GPU:
a parallel-mode Convolver (output o convolves input o only — Convolver(numIO, latency), Convolver.cpp:21-36) with config 2's
partitioning.  Inputs are the bench's generator (IR seed 1000*ch + ch + 1, audio seed 777 + ch); channel 0 is checked against a
float64 FFT convolution of the same data.  Prints one JSON line per channel count."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hisstools_library_amd as H
import bench

_, _, L, fs, layout = bench.WORKLOADS["c2"]
B, STEPS = 8192, 200
dev = torch.device("cuda", 0)
for n in [int(a) for a in sys.argv[1:]] or [1, 16, 64]:
    conv = H.Convolver(n, None, 0, device=0, maxBlock=B, custom=(L, *layout))
    synth = bench.IrSynth(L, dev, [(c, c) for c in range(n)])
    h0 = None
    for c in range(n):
        h = synth.get(c, c)
        if c == 0:
            h0 = h.double().cpu().numpy()
        torch.cuda.synchronize()
        assert conv.set_dev(c, c, h.data_ptr(), L, True) == 0
    synth.close()
    warm = L // B + 2
    total = (warm + 8) * B
    x = np.stack([bench.synth_audio(c, total) for c in range(n)])
    xs = torch.from_numpy(x).to(dev)
    ys = torch.zeros((n, total), device=dev)
    torch.cuda.synchronize()
    for k in range(warm + 8):                                       # ramp-up on fresh audio: every partition live, and a checkable output
        conv.process_dev(xs.data_ptr() + 4 * k * B, total, ys.data_ptr() + 4 * k * B, total, n, n, B)
    conv.synchronize()
    m = 1 << int(np.ceil(np.log2(total + L)))
    truth = np.fft.irfft(np.fft.rfft(x[0].astype(np.float64), m) * np.fft.rfft(h0, m), m)[:total]
    lat = 0 if layout[0] else layout[1] // 2                          # a chain without the time-domain head delays by half its first FFT
    y0 = ys[0].double().cpu().numpy()
    err = float(max(np.abs(y0[lat:] - truth[:total - lat]).max(), np.abs(y0[:lat]).max() if lat else 0.0) / np.abs(truth).max())
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        for k in range(STEPS):
            off = 4 * (k % 8) * B
            conv.process_dev(xs.data_ptr() + off, total, ys.data_ptr() + off, total, n, n, B)
        conv.synchronize()
        dt = (time.perf_counter() - t0) / STEPS
        best = dt if best is None else min(best, dt)
    print(json.dumps({"workload": f"{n} independent c2 convolvers (parallel-mode Convolver, one 4096-point stage, IR {L} samples), {B}-sample steps",
                      "channels": n, "ms_per_step": round(1e3 * best, 4), "msamples_per_s": round(n * B / best / 1e6, 1),
                      "live_spectra_mib": round(n * 8 * 2048 * -(-L // 2048) / 1048576.0, 1),
                      "max_rel_err_channel0_vs_float64": err}), flush=True)
    assert err < 1e-5, err
    del conv
