"""Scenarios of live control calls shared by tests/golden/make_golden_restart.py (which records the unmodified reference's
output) and tests/test_restart_golden.py (which replays them on the oracle and on the HIP path).  Everything is derived from
the seeded generators of oracle/oracle.py, so no inputs need to be stored."""
import numpy as np

# name -> shape, IR length, samples, and a list of (position, op, in, out, ir seed or None, ir length)
SCENARIOS = {
    "zero_latency_2x2": dict(nin=2, nout=2, latency=0, L=30000, S=90000, parallel=False, events=[
        (20011, "set", 1, 0, 7, 30000), (33000, "reset", 0, 1, None, 0), (41003, "set", 0, 0, 8, 12000), (60000, "clear", 1, 1, None, 0),
        (70777, "set", 1, 1, 9, 25000)]),
    # resize(in, out, length) silences the pair until its next set (MonoConvolve.cpp:101-110: mLength = 0)
    "resize_then_set": dict(nin=2, nout=2, latency=0, L=20000, S=70000, parallel=False, events=[
        (18001, "resize", 0, 1, None, 40000), (30000, "set", 0, 1, 10, 35000), (45005, "resize", 1, 0, None, 1000), (52000, "set", 1, 0, 11, 1000)]),
    "medium_latency_2to1": dict(nin=2, nout=1, latency=2, L=6000, S=60000, parallel=False, events=[
        (17000, "reset", 1, 0, None, 0), (30001, "set", 0, 0, 5, 6000)]),
    "parallel_3": dict(nin=3, nout=3, latency=0, L=20000, S=70000, parallel=True, events=[
        (23333, "set", 1, 1, 6, 15000), (40000, "reset", 2, 2, None, 0)]),
}


def build(ns, sc, **kw):
    """ns: oracle.oracle or hisstools_library_amd; returns (convolver with the initial IRs, inputs, script)"""
    from oracle import oracle as O
    if sc["parallel"]:
        conv = ns.Convolver(sc["nin"], None, sc["latency"], **kw)
        pairs = [(k, k) for k in range(sc["nin"])]
    else:
        conv = ns.Convolver(sc["nin"], sc["nout"], sc["latency"], **kw)
        pairs = [(i, o) for i in range(sc["nin"]) for o in range(sc["nout"])]
    for i, o in pairs:
        assert conv.set(i, o, O.synth_ir(i, o, sc["L"]), True) == 0
    xs = np.stack([O.synth_audio(100 + i, sc["S"]) for i in range(sc["nin"])])
    script = []
    for pos, op, i, o, seed, n in sc["events"]:
        if op == "set":
            h = O.synth_ir(seed, seed, n)
            script.append((pos, lambda c, i=i, o=o, h=h: c.set(i, o, h, True)))
        elif op == "reset":
            script.append((pos, lambda c, i=i, o=o: c.reset(i, o)))
        elif op == "resize":
            script.append((pos, lambda c, i=i, o=o, n=n: c.resize(i, o, n)))
        else:
            script.append((pos, lambda c, i=i, o=o: c.clear(i, o, False)))
    return conv, xs, script


def drive(conv, xs, nout, script, block):
    total = xs.shape[1]
    cuts = sorted({0, total} | {pos for pos, _ in script})
    out = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        for pos, fn in script:
            if pos == a:
                fn(conv)
        out.append(conv.run(np.ascontiguousarray(xs[:, a:b]), nout, block))
    return np.concatenate(out, axis=1)


# NToMonoConvolve (one output, reset(inChan) / set(inChan, …) mid-stream): (position, op, inChan, ir seed, ir length)
NTOMONO = dict(nin=3, latency=1, L=16000, S=60000, events=[(15001, "reset", 1, None, 0), (26000, "set", 2, 11, 9000), (40500, "set", 0, 12, 16000)])


def build_ntomono(ns, **kw):
    from oracle import oracle as O
    sc = NTOMONO
    conv = ns.NToMonoConvolve(sc["nin"], sc["L"], sc["latency"], **kw)
    for i in range(sc["nin"]):
        assert conv.set(i, O.synth_ir(i, 3, sc["L"]), True) == 0
    xs = np.stack([O.synth_audio(120 + i, sc["S"]) for i in range(sc["nin"])])
    script = []
    for pos, op, i, seed, n in sc["events"]:
        if op == "set":
            h = O.synth_ir(seed, seed, n)
            script.append((pos, lambda c, i=i, h=h: c.set(i, h, True)))
        else:
            script.append((pos, lambda c, i=i: c.reset(i)))
    return conv, xs, script


def drive_ntomono(conv, xs, script, block):
    total = xs.shape[1]
    cuts = sorted({0, total} | {pos for pos, _ in script})
    out = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        for pos, fn in script:
            if pos == a:
                fn(conv)
        out.append(conv.run(np.ascontiguousarray(xs[:, a:b]), block))
    return np.concatenate(out)
