"""Worker for tests/test_sharded_gloo.py: run under torch.distributed.run with the gloo backend (CPU).
The per-rank engine is the CPU oracle (injected through engine_factory) so the sharding / reduction logic of
hisstools_library_amd.sharded can be checked without a GPU."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hisstools_library_amd.sharded import ShardedConvolver, split_range  # noqa: E402
from oracle import oracle as O  # noqa: E402


class OracleEngine:
    """Adapter with the Convolver surface ShardedConvolver needs."""

    def __init__(self, numIns, numOuts, latency, device):
        self.c = O.Convolver(numIns, numOuts, latency)
        self.c.setResetOffset(0)

    def set(self, i, o, ir, resize):
        return self.c.set(i, o, ir, resize)

    def reset(self, *pair):
        return self.c.reset(*pair)

    def process(self, ins, outs):
        self.c.process(ins, outs)


def main():
    layout = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    nin, nout, L, S, B = 3, 4, 9000, 6144, 512
    if layout == "grid":
        nin, nout = 4, 1                                  # fewer outputs than ranks: inputs are split, partials all-reduced
    irs = {(i, o): O.synth_ir(i, o, L) for i in range(nin) for o in range(nout)}
    xs = np.stack([O.synth_audio(i, S) for i in range(nin)])

    full = O.Convolver(nin, nout, 0)
    full.setResetOffset(0)
    for (i, o), h in irs.items():
        assert full.set(i, o, h, True) == 0
    expect = full.run(xs, nout, B)

    sc = ShardedConvolver(nin, nout, 0, layout=layout, grid=(1, world) if layout == "grid" else None, engine_factory=OracleEngine)
    for (i, o), h in irs.items():
        assert sc.set(i, o, h, True) == 0
    assert sc.set(nin, 0, irs[(0, 0)], True) == 1 and sc.set(0, nout, irs[(0, 0)], True) == 2
    owned = sum(sc.owns(i, o) for i in range(nin) for o in range(nout))
    counts = [None] * world
    dist.all_gather_object(counts, owned)
    assert sum(counts) == nin * nout, counts                 # every pair has exactly one owner

    got = np.zeros((sc.nout_local, S), np.float32)
    for pos in range(0, S, B):
        got[:, pos:pos + B] = sc.process(xs[:, pos:pos + B])
    lo, hi = split_range(nout, sc.go, sc.row)
    peak = np.abs(expect).max()
    err = np.abs(got - expect[lo:hi]).max() / peak if hi > lo else 0.0
    tol = 0.0 if layout == "rows" else 2e-6                  # rows: same arithmetic as one process; grid: re-associated sum
    assert err <= tol, (rank, err)
    allouts = sc.gather(got)
    assert np.abs(allouts - expect).max() / peak <= tol

    # control calls while streaming: an IR swap, a restart and a clear of single pairs reach their owner and nobody else
    new_ir = O.synth_ir(7, 7, L - 500)
    script = {1024: lambda c: c.set(nin - 1, 0, new_ir, True), 2048: lambda c: c.reset(0, nout - 1), 3072: lambda c: c.clear(nin - 1, nout - 1, False)}
    full.reset()
    sc.reset()
    assert sc.reset(nin, 0) == 1 and sc.reset(0, nout) == 2
    for pos in range(0, S, B):
        if pos in script:
            assert script[pos](full) in (0, None) and script[pos](sc) == 0
        expect[:, pos:pos + B] = full.run(xs[:, pos:pos + B], nout, B)
        got[:, pos:pos + B] = sc.process(xs[:, pos:pos + B])
    err2 = np.abs(got - expect[lo:hi]).max() / peak if hi > lo else 0.0
    assert err2 <= tol, (rank, err2)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok ({layout}, err {err:.2e})")


if __name__ == "__main__":
    main()
