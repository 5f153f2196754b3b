"""Worker of tests/test_fused_nxm_gpu.py: one engine with several outputs whose hop-sized blocks take the n x m fused block
(hcv_fused_nxm.hip), dense impulse responses on every pair, against the CPU oracle.

argv: nin nout L hops mode [engines]
  mode = host   : hop-sized host-pointer calls (synchronous)
         dev    : back-to-back asynchronous process_dev calls, repeated (every repetition must give the same bits)
         mixed  : hop-sized calls, a ragged stretch in the middle (leaving and re-entering whole-hop mode) and a live IR swap of one
                  pair (control work: the forward stream is joined), against the oracle driven through the same calls
         many   : `engines` engines at once, each with its own host thread issuing asynchronous calls (contention)
Prints one JSON line.
"""
import hashlib
import json
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import hisstools_library_amd as H
from oracle import oracle as O

nin, nout, L, hops, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
K = int(sys.argv[6]) if len(sys.argv) > 6 else 1
B = 8192
n = B * hops
dev = torch.device("cuda:0")


def rel_err(y, ref):
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(np.asarray(y, dtype=np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))


xs = np.stack([O.synth_audio(i, n) for i in range(nin)])
irs = {(i, o): O.synth_ir(i, o, L - 37 * ((i + o) % 5)) for o in range(nout) for i in range(nin)}


def make():
    c = H.Convolver(nin, nout, 0, custom=(L, True, 256, 1024, 4096, 16384), maxBlock=B)
    for (i, o), h in irs.items():
        assert c.set(i, o, h, True) == 0
    return c


ref = O.Convolver(nin, nout, 0)
ref.setResetOffset(0)
for (i, o), h in irs.items():
    assert ref.set(i, o, h, True) == 0

res = {"mode": mode}
if mode == "host":
    c = make()
    c.clear_stats()
    y = c.run(xs, nout, B)
    y_ref, _ = ref.stream_timed(xs, nout, 2048)
    res["max_err"] = max(rel_err(y[o], y_ref[o]) for o in range(nout))
    res["tail_err"] = max(rel_err(y[o][-3 * B:], y_ref[o][-3 * B:]) for o in range(nout))
    res["fused_launches"] = int(c.stage_stats()[-1]["fused_launches"])
    res["ksplit"] = int(c.stage_stats()[-1]["ksplit"])
    res["sha"] = hashlib.sha256(np.ascontiguousarray(y).tobytes()).hexdigest()
elif mode == "dev":
    c = make()
    xd = torch.from_numpy(xs).to(dev)
    yd = torch.zeros((nout, n), device=dev)
    y_ref, _ = ref.stream_timed(xs, nout, 2048)
    shas, errs = [], []
    t_blk = None
    for rep in range(3):
        c.reset()
        c.clear_stats()
        yd.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(hops):
            c.process_dev(xd.data_ptr() + 4 * k * B, n, yd.data_ptr() + 4 * k * B, n, nin, nout, B, sync=False)
        c.synchronize()
        t_blk = (time.perf_counter() - t0) / hops
        y = yd.cpu().numpy()
        shas.append(hashlib.sha256(y.tobytes()).hexdigest())
        errs.append(max(rel_err(y[o], y_ref[o]) for o in range(nout)))
    res["max_err"] = max(errs)
    res["all_same"] = len(set(shas)) == 1
    res["sha"] = shas[0]
    res["fused_launches"] = int(c.stage_stats()[-1]["fused_launches"])
    res["stood_down"] = int(c.stage_stats()[-1]["fused_stood_down"])
    res["ms_per_block"] = t_blk * 1e3
elif mode == "mixed":
    c = make()
    # hop-sized calls; a ragged stretch that adds up to whole hops again; hop-sized calls with one pair's IR swapped on the way
    plan = [B] * 14 + [300, 8192 - 300 - 128, 128, B // 2, B // 2] + [B] * 3
    swap_at = 17
    new_ir = O.synth_ir(99, 99, L)
    pos, ys, yrs = 0, [], []
    for k, b in enumerate(plan):
        if pos + b > n:
            break
        if k == swap_at:
            assert c.set(1, 0, new_ir, True) == 0 and ref.set(1, 0, new_ir, True) == 0
        ys.append(c.run(xs[:, pos:pos + b], nout, b))
        yrs.append(ref.run(xs[:, pos:pos + b], nout, min(b, 2048)))
        pos += b
    y, y_ref = np.concatenate(ys, axis=1), np.concatenate(yrs, axis=1)
    res["max_err"] = max(rel_err(y[o], y_ref[o]) for o in range(nout))
    res["fused_launches"] = int(c.stage_stats()[-1]["fused_launches"])
    res["samples"] = pos
else:
    xd = torch.from_numpy(xs).to(dev)
    y_ref, _ = ref.stream_timed(xs, nout, 2048)
    engines = [make() for _ in range(K)]
    outs = [torch.zeros((nout, n), device=dev) for _ in range(K)]
    torch.cuda.synchronize()
    start = threading.Barrier(K)
    result = [None] * K

    def drive(k):
        c, yd = engines[k], outs[k]
        shas, errs = [], []
        start.wait()
        for rep in range(3):
            c.reset()
            for j in range(hops):
                c.process_dev(xd.data_ptr() + 4 * j * B, n, yd.data_ptr() + 4 * j * B, n, nin, nout, B, sync=False)
            c.synchronize()
            y = yd.cpu().numpy()
            shas.append(hashlib.sha256(y.tobytes()).hexdigest())
            errs.append(max(rel_err(y[o], y_ref[o]) for o in range(nout)))
        result[k] = (shas, max(errs), int(c.stage_stats()[-1]["fused_launches"]), int(c.stage_stats()[-1]["fused_stood_down"]))

    ts = [threading.Thread(target=drive, args=(k,)) for k in range(K)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    res["max_err"] = max(r[1] for r in result)
    res["all_same"] = len({s for r in result for s in r[0]}) == 1
    res["sha"] = result[0][0][0]
    res["fused_launches"] = min(r[2] for r in result)
    res["stood_down"] = sum(r[3] for r in result)
print(json.dumps(res))
