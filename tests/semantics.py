"""Behaviour checklist of the reference API, written once and run against any implementation
namespace: the oracle port, the compiled reference, and the HIP classes.  Returns a dict of observations
(error codes, flags) that must be identical everywhere."""
import numpy as np


def _ir(n, seed=1):
    rng = np.random.RandomState(seed)
    return (rng.uniform(-1, 1, n) * np.exp(-3.0 * np.arange(n) / max(n, 1))).astype(np.float32)


def checklist(ns):
    obs = {}
    x = np.random.RandomState(7).uniform(-1, 1, 1024).astype(np.float32)

    # --- PartitionedConvolve (PartitionedConvolve.cpp)
    p = ns.PartitionedConvolve(4096, 8192, 0, 0)
    p.setResetOffset(0)
    obs["part.setFFTSize(64)"] = p.setFFTSize(64)
    obs["part.setFFTSize(16)"] = p.setFFTSize(16)
    obs["part.setFFTSize(8192>max)"] = p.setFFTSize(8192)
    obs["part.setFFTSize(100)"] = p.setFFTSize(100)          # 12, but applies 128
    obs["part.setLength(100000)"] = p.setLength(100000)
    obs["part.setLength(0)"] = p.setLength(0)
    wrote, out = p.process(x)
    obs["part.process_without_ir.wrote"] = wrote
    obs["part.process_without_ir.untouched"] = bool(np.isnan(out).all())
    obs["part.set(ir)"] = p.set(_ir(1000))
    wrote, out = p.process(x)
    obs["part.process.wrote"] = wrote
    obs["part.first_half_fft_zero"] = bool((out[:64] == 0).all())       # fft 128 -> latency 64
    obs["part.set(too_long)"] = p.set(_ir(9000))
    obs["part.set(None)"] = p.set(None)
    obs["part.process_after_clear.wrote"] = p.process(x)[0]
    obs["part.setFFTSize_changes_then_process"] = (p.set(_ir(500)), p.setFFTSize(256), p.process(x)[0])
    p2 = ns.PartitionedConvolve(1000, 100, 0, 0)                        # non power of two max -> 1024
    obs["part.nonpow2_max.setFFTSize(1024)"] = p2.setFFTSize(1024)
    obs["part.nonpow2_max.setFFTSize(2048)"] = p2.setFFTSize(2048)
    p3 = ns.PartitionedConvolve(256, 1000, 100, 300)
    obs["part.window.set_short"] = p3.set(_ir(50))                      # total_len <= offset -> nothing loaded
    obs["part.window.wrote"] = p3.process(x)[0]

    # --- TimeDomainConvolve (TimeDomainConvolve.cpp)
    t = ns.TimeDomainConvolve(0, 128)
    obs["td.setLength(3000)"] = t.setLength(3000)
    obs["td.setLength(0)"] = t.setLength(0)
    obs["td.set(3000 taps, unlimited)"] = t.set(_ir(3000))
    obs["td.setLength(128)"] = t.setLength(128)
    obs["td.set(3000 taps, 128)"] = t.set(_ir(3000))
    wrote, out = t.process(x)
    obs["td.process.wrote"] = wrote
    obs["td.set(None)"] = t.set(None)
    wrote, out = t.process(x)
    obs["td.process_without_ir.wrote"] = wrote
    obs["td.process_without_ir.zeros"] = bool((out == 0).all())

    # --- MonoConvolve (MonoConvolve.cpp)
    for name, args in (("bad_order", dict(zeroLatency=True, A=1024, B=256)), ("too_small", dict(zeroLatency=False, A=16)),
                       ("none", dict(zeroLatency=False, A=0)), ("too_big", dict(zeroLatency=False, A=1 << 21))):
        try:
            ns.MonoConvolve(1000, **args)
            obs[f"mono.ctor.{name}"] = "ok"
        except RuntimeError as e:
            obs[f"mono.ctor.{name}"] = str(e)
    m = ns.MonoConvolve(16384, latency=0)
    out = m.process(x)
    obs["mono.process_without_ir.untouched"] = bool(np.isnan(out).all())
    obs["mono.set(20000, no resize)"] = m.set(_ir(20000), False)
    obs["mono.too_long.silent"] = bool(np.isnan(m.process(x)).all())
    obs["mono.set(20000, resize)"] = m.set(_ir(20000), True)
    obs["mono.loaded.writes"] = bool(np.isfinite(m.process(x)).all())
    obs["mono.set(None)"] = m.set(None, False)
    obs["mono.cleared.silent"] = bool(np.isnan(m.process(x)).all())
    obs["mono.set(100)"] = m.set(_ir(100), False)
    acc = np.ones(x.size, np.float32)
    m.reset()
    y = m.process(x)
    m.reset()
    ya = m.process(x, out=acc.copy(), accumulate=True)
    obs["mono.resize(5000)"] = m.resize(5000)
    obs["mono.after_resize.silent"] = bool(np.isnan(m.process(x)).all())
    obs["mono.reset"] = m.reset()
    m0 = ns.MonoConvolve(0, latency=1)
    obs["mono.maxlen0.set"] = m0.set(_ir(100), False)
    obs["mono.maxlen0.set_resize"] = m0.set(_ir(100), True)

    # --- NToMonoConvolve (NToMonoConvolve.cpp)
    c = ns.NToMonoConvolve(3, 16384, 1)
    obs["n2m.set(in=3)"] = c.set(3, _ir(10), False)
    obs["n2m.set(in=2)"] = c.set(2, _ir(10), False)
    obs["n2m.reset(in=5)"] = c.reset(5)
    obs["n2m.reset(in=0)"] = c.reset(0)
    obs["n2m.resize(in=3)"] = c.resize(3, 100)
    obs["n2m.resize(in=1)"] = c.resize(1, 100)

    # --- Convolver (Convolver.cpp)
    cv = ns.Convolver(2, 3, 0)
    obs["conv.set(in=2)"] = cv.set(2, 0, _ir(10), False)
    obs["conv.set(out=3)"] = cv.set(0, 3, _ir(10), False)
    obs["conv.set(ok)"] = cv.set(1, 2, _ir(10), False)
    obs["conv.set(too long)"] = cv.set(1, 2, _ir(17000), False)
    obs["conv.set(too long, resize)"] = cv.set(1, 2, _ir(17000), True)
    obs["conv.set_f64"] = cv.set(0, 0, _ir(10).astype(np.float64), False)
    obs["conv.reset(out=3)"] = cv.reset(0, 3)
    obs["conv.reset(in=2)"] = cv.reset(2, 0)
    obs["conv.reset(ok)"] = cv.reset(1, 1)
    obs["conv.resize(out=3)"] = cv.resize(0, 3, 100)            # sic: IN_CHAN code (Convolver.cpp:108-111)
    obs["conv.resize(in=2)"] = cv.resize(2, 0, 100)
    obs["conv.resize(ok)"] = cv.resize(0, 0, 100)
    cp = ns.Convolver(3, None, 0)
    obs["par.set(1,1)"] = cp.set(1, 1, _ir(10), False)
    obs["par.set(0,1)"] = cp.set(0, 1, _ir(10), False)
    obs["par.set(2,1)"] = cp.set(2, 1, _ir(10), False)
    obs["par.set(3,3)"] = cp.set(3, 3, _ir(10), False)
    obs["par.reset(1,2)"] = cp.reset(1, 2)
    obs["par.resize(2,2)"] = cp.resize(2, 2, 50)
    c1 = ns.Convolver(0, 2, 0)                                   # numIns < 1 is bumped to 1 (Convolver.cpp:8)
    obs["conv.zero_ins.set(0,1)"] = c1.set(0, 1, _ir(10), False)
    obs["conv.zero_ins.set(1,1)"] = c1.set(1, 1, _ir(10), False)
    return obs, (y, ya)


# What the unmodified reference answers (pinned by tests/test_semantics.py against oracle/_ref).
EXPECTED = {
    "part.setFFTSize(64)": 0, "part.setFFTSize(16)": 11, "part.setFFTSize(8192>max)": 11, "part.setFFTSize(100)": 12,
    "part.setLength(100000)": 7, "part.setLength(0)": 0,
    "part.process_without_ir.wrote": False, "part.process_without_ir.untouched": True,
    "part.set(ir)": 0, "part.process.wrote": True, "part.first_half_fft_zero": True,
    "part.set(too_long)": 4, "part.set(None)": 0, "part.process_after_clear.wrote": False,
    "part.setFFTSize_changes_then_process": (0, 0, False),
    "part.nonpow2_max.setFFTSize(1024)": 0, "part.nonpow2_max.setFFTSize(2048)": 11,
    "part.window.set_short": 0, "part.window.wrote": False,
    "td.setLength(3000)": 6, "td.setLength(0)": 0, "td.set(3000 taps, unlimited)": 5, "td.setLength(128)": 0,
    "td.set(3000 taps, 128)": 0, "td.process.wrote": True, "td.set(None)": 0,
    "td.process_without_ir.wrote": False, "td.process_without_ir.zeros": True,
    "mono.ctor.bad_order": "invalid FFT size or order", "mono.ctor.too_small": "invalid FFT size or order",
    "mono.ctor.none": "no valid FFT sizes given", "mono.ctor.too_big": "invalid FFT size or order",
    "mono.process_without_ir.untouched": True,
    "mono.set(20000, no resize)": 4, "mono.too_long.silent": True, "mono.set(20000, resize)": 0, "mono.loaded.writes": True,
    "mono.set(None)": 0, "mono.cleared.silent": True, "mono.set(100)": 0,
    "mono.resize(5000)": 0, "mono.after_resize.silent": True, "mono.reset": 0,
    "mono.maxlen0.set": 3, "mono.maxlen0.set_resize": 0,
    "n2m.set(in=3)": 1, "n2m.set(in=2)": 0, "n2m.reset(in=5)": 1, "n2m.reset(in=0)": 0, "n2m.resize(in=3)": 1, "n2m.resize(in=1)": 0,
    "conv.set(in=2)": 1, "conv.set(out=3)": 2, "conv.set(ok)": 0, "conv.set(too long)": 4, "conv.set(too long, resize)": 0,
    "conv.set_f64": 0, "conv.reset(out=3)": 2, "conv.reset(in=2)": 1, "conv.reset(ok)": 0,
    "conv.resize(out=3)": 1, "conv.resize(in=2)": 1, "conv.resize(ok)": 0,
    "par.set(1,1)": 0, "par.set(0,1)": 1, "par.set(2,1)": 1, "par.set(3,3)": 2, "par.reset(1,2)": 1, "par.resize(2,2)": 0,
    "conv.zero_ins.set(0,1)": 0, "conv.zero_ins.set(1,1)": 1,
}
