#!/usr/bin/env python3
"""Generate tests/golden/golden_v1.npz from the UNMODIFIED reference (oracle/_ref, compiled from /root/reference
by oracle/Makefile).  Run in the build container only:   python tests/golden/make_golden.py

Every array is float32.  Inputs are stored next to the expected outputs so the fixture is self-contained.
Cases driven through MonoConvolve/PartitionedConvolve pin the reset offset (bit-reproducible); the
NToMonoConvolve/Convolver API cannot pin it (random FFT phases, MonoConvolve.cpp:89-90), so those outputs are
reproducible to rounding only (~1e-6 of peak).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

G = {}
BE = "ref"


def audio(ch, n):
    return O.synth_audio(ch, n)


def ir(i, o, n):
    return O.synth_ir(i, o, n)


# ---- FFT format vectors (HISSTools_FFT.cpp:226-248)
for l2 in (5, 8, 10, 12, 14):
    n = 1 << l2
    x = audio(100 + l2, n)
    re, im = O.rfft(x, l2, BE)
    G[f"fft{l2}_x"], G[f"fft{l2}_re"], G[f"fft{l2}_im"] = x, re, im
    G[f"fft{l2}_inv"] = O.rifft(re, im, l2, BE)
# short / odd-length input (unzip_zero, HISSTools_FFT_Core.h:1258-1287)
x = audio(120, 77)
re, im = O.rfft(x, 8, BE)
G["fft8_odd_x"], G["fft8_odd_re"], G["fft8_odd_im"] = x, re, im

# ---- PartitionedConvolve
for name, N, L, S, ro, blocks in (("part256", 256, 1000, 3000, 0, 512), ("part1024", 1024, 3000, 6000, 77, [1, 7, 333, 2000, 64])):
    h, x = ir(1, 0, L), audio(1, S)
    p = O.PartitionedConvolve(N, L, 0, 0, backend=BE)
    p.setResetOffset(ro)
    assert p.set(h) == 0
    G[f"{name}_ir"], G[f"{name}_x"], G[f"{name}_y"] = h, x, p.run(x, blocks)
# offset / length window: IR[300 : 300+500]
h, x = ir(2, 0, 1200), audio(2, 2500)
p = O.PartitionedConvolve(256, 1024, 300, 500, backend=BE)
p.setResetOffset(0)
assert p.set(h) == 0
G["partwin_ir"], G["partwin_x"], G["partwin_y"] = h, x, p.run(x, 256)

# ---- TimeDomainConvolve
h, x = ir(3, 0, 2044), audio(3, 3000)
G["td_ir"], G["td_x"] = h, x
for Lh in (1, 16, 128, 2044):
    t = O.TimeDomainConvolve(0, Lh, backend=BE)
    t.set(h)
    G[f"td{Lh}_y"] = t.run(x, 512)

# ---- MonoConvolve (reset offset pinned to 0)
h, x = ir(4, 0, 11000), audio(4, 14000)
G["mono_ir"], G["mono_x"] = h, x
for mode in (0, 1, 2):
    m = O.MonoConvolve(16384, latency=mode, backend=BE)
    m.setResetOffset(0)
    assert m.set(h, True) == 0
    G[f"mono{mode}_y"] = m.run(x, 512)
m = O.MonoConvolve(11000, zeroLatency=False, A=512, B=2048, backend=BE)
m.setResetOffset(0)
assert m.set(h, False) == 0
G["monoc_y"] = m.run(x, [100, 900, 2048])

# ---- NToMonoConvolve 3 -> 1 (random phases)
irs = np.stack([ir(i, 0, 3000) for i in range(3)])
xs = np.stack([audio(10 + i, 4000) for i in range(3)])
c = O.NToMonoConvolve(3, 16384, 0, backend=BE)
for i in range(3):
    assert c.set(i, irs[i], True) == 0
G["n2m_irs"], G["n2m_x"], G["n2m_y"] = irs, xs, c.run(xs, 512)

# ---- Convolver 2 x 3 matrix and 3-channel parallel (random phases)
irs = np.stack([np.stack([ir(i, o, 2500) for i in range(2)]) for o in range(3)])          # [out][in][L]
xs = np.stack([audio(20 + i, 3500) for i in range(2)])
c = O.Convolver(2, 3, 0, backend=BE)
for o in range(3):
    for i in range(2):
        assert c.set(i, o, irs[o, i], True) == 0
G["conv_irs"], G["conv_x"], G["conv_y"] = irs, xs, c.run(xs, 3, 512)

irs = np.stack([ir(30 + o, o, 2000) for o in range(3)])
xs = np.stack([audio(30 + o, 3000) for o in range(3)])
c = O.Convolver(3, None, 1, backend=BE)
for o in range(3):
    assert c.set(o, o, irs[o], True) == 0
G["par_irs"], G["par_x"], G["par_y"] = irs, xs, c.run(xs, 3, 256)

# ---- spectral_processor<float>::convolve / correlate, real overloads (first "next" row); separate fixture file
S = {}
if O.have_ref_spectral():
    cases = [(10, 4), (4, 10), (7, 7), (1, 1), (1, 9), (100, 33), (33, 100), (512, 512), (1000, 129), (2, 3), (3000, 2047)]
    for ci, (n1, n2) in enumerate(cases):
        a, b = audio(200 + ci, n1), audio(300 + ci, n2)
        S[f"sp{ci}_a"], S[f"sp{ci}_b"] = a, b
        for mode in range(5):
            S[f"sp{ci}_conv{mode}"] = O.spectral_convolve(a, b, mode, BE)
            S[f"sp{ci}_corr{mode}"] = O.spectral_correlate(a, b, mode, BE)
    sout = os.path.join(ROOT, "tests", "golden", "golden_spectral_v1.npz")
    np.savez_compressed(sout, **{k: np.asarray(v, np.float32) for k, v in S.items()})
    print("wrote", sout, os.path.getsize(sout), "bytes,", sum(v.size for v in S.values()), "floats")

out = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")
np.savez_compressed(out, **{k: np.asarray(v, np.float32) for k, v in G.items()})
print("wrote", out, os.path.getsize(out), "bytes,", sum(v.size for v in G.values()), "floats")
