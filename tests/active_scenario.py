"""A host that changes numIns / numOuts between process calls (Convolver.cpp:138-154 processes only the channels it is
handed; NToMonoConvolve.cpp:41 only min(inChans, activeIns) inputs).  Shared by tests/golden/make_golden_active.py, which
records the UNMODIFIED reference, and tests/test_active_counts_golden.py."""
import numpy as np

SC = dict(nin=3, nout=3, latency=0, L=9000, S=48000, cuts=[(0, 3, 3), (12000, 2, 2), (24000, 3, 3)])


def build(ns, **kw):
    from oracle import oracle as O
    conv = ns.Convolver(SC["nin"], SC["nout"], SC["latency"], **kw)
    irs = {(i, o): O.synth_ir(20 + i, 30 + o, SC["L"] - 500 * i) for i in range(SC["nin"]) for o in range(SC["nout"])}
    for (i, o), h in irs.items():
        assert conv.set(i, o, h, True) == 0
    xs = np.stack([O.synth_audio(200 + i, SC["S"]) for i in range(SC["nin"])])
    return conv, xs, irs


def drive(conv, xs, block):
    """Streams the scenario; outputs a channel count leaves out stay zero for that span (process does not touch them)."""
    S = SC["S"]
    ys = np.zeros((SC["nout"], S), np.float32)
    spans = [(a, (SC["cuts"][k + 1][0] if k + 1 < len(SC["cuts"]) else S), ni, no) for k, (a, ni, no) in enumerate(SC["cuts"])]
    for a, b, ni, no in spans:
        pos = a
        while pos < b:
            n = min(block, b - pos)
            ins = np.ascontiguousarray(xs[:, pos:pos + n])
            outs = np.zeros((SC["nout"], n), np.float32)
            conv.process(ins, outs, ni, no)
            ys[:no, pos:pos + n] = outs[:no]
            pos += n
    return ys
