"""Multi-GPU sharding of the N x M convolution matrix: one process per GPU (torch.distributed, RCCL over xGMI).

(in, out) pairs are independent up to the per-output sum over inputs (NToMonoConvolve.cpp:39-42), so

* ``rows`` layout — rank r owns a contiguous block of OUTPUT rows and every input.  No collective on the data
  path: each rank produces final samples for its outputs (``gather()`` is only for callers that want every output
  everywhere).
* ``grid`` layout (Go x Gi ranks) — the output rows are split Go ways and the inputs Gi ways; a rank produces a
  PARTIAL block for its outputs from its inputs, and the Gi ranks of one output block sum their partials with one
  all-reduce per process call (the only exchange step the path has).  Needed when there are fewer outputs than
  GPUs (e.g. 8 -> 1) or for load balance.

torch.distributed is plumbing here (process groups, RCCL); the per-rank compute is the HIP engine
(hisstools_library_amd.Convolver).  ``engine_factory`` exists so the partitioning / reduction logic can be exercised
on CPU-only machines with a stand-in engine (tests/test_sharded_gloo.py injects one there).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import numpy as np


def split_range(n: int, parts: int, index: int) -> Tuple[int, int]:
    """Contiguous balanced split of range(n) into `parts`; returns [lo, hi) of block `index`."""
    base, rem = divmod(n, parts)
    lo = index * base + min(index, rem)
    return lo, lo + base + (1 if index < rem else 0)


def _default_factory(numIns: int, numOuts: int, latency: int, device: int):
    from .convolver import Convolver
    return Convolver(numIns, numOuts, latency, device=device)


class ShardedConvolver:
    def __init__(self, numIns: int, numOuts: int, latency: int = 0, layout: str = "rows", grid: Optional[Tuple[int, int]] = None,
                 rank: Optional[int] = None, world_size: Optional[int] = None, device: Optional[int] = None,
                 engine_factory: Optional[Callable] = None):
        import torch.distributed as dist
        self.dist = dist
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world_size is None else world_size
        self.numIns, self.numOuts = numIns, numOuts
        if layout == "rows":
            go, gi = self.world, 1
        elif layout == "grid":
            if grid is None or grid[0] * grid[1] != self.world:
                raise ValueError("grid layout needs grid=(Go, Gi) with Go*Gi == world size")
            go, gi = grid
        else:
            raise ValueError(layout)
        self.go, self.gi = go, gi
        self.row, self.col = divmod(self.rank, gi)                 # rank = row * Gi + col
        self.out_lo, self.out_hi = split_range(numOuts, go, self.row)
        self.in_lo, self.in_hi = split_range(numIns, gi, self.col)
        self.nout_local = self.out_hi - self.out_lo
        self.nin_local = self.in_hi - self.in_lo
        dev = (self.rank if device is None else device)
        factory = engine_factory or _default_factory
        self.engine = factory(max(self.nin_local, 1), self.nout_local, latency, dev) if self.nout_local > 0 else None
        # one reduction group per output block (the Gi ranks that share its rows); every rank must create every group
        self.row_group = None
        if gi > 1:
            for r in range(go):
                g = dist.new_group(ranks=[r * gi + c for c in range(gi)])
                if r == self.row:
                    self.row_group = g

    # ---- ownership
    def owns(self, inChan: int, outChan: int) -> bool:
        return self.out_lo <= outChan < self.out_hi and self.in_lo <= inChan < self.in_hi

    # ---- IR management: SPMD — every rank may call set() for every pair; only the owner loads it
    def set(self, inChan: int, outChan: int, ir, resize: bool = True) -> int:
        if inChan >= self.numIns:
            return 1            # CONVOLVE_ERR_IN_CHAN_OUT_OF_RANGE
        if outChan >= self.numOuts:
            return 2            # CONVOLVE_ERR_OUT_CHAN_OUT_OF_RANGE
        if not self.owns(inChan, outChan) or self.engine is None:
            return 0
        return self.engine.set(inChan - self.in_lo, outChan - self.out_lo, ir, resize)

    def reset(self, inChan: Optional[int] = None, outChan: Optional[int] = None):
        """reset() restarts every pair; reset(in, out) one pair while the others keep running (Convolver.cpp:79-97) — SPMD like
        set(): every rank may call it, the owner acts."""
        if inChan is None:
            if self.engine is not None:
                self.engine.reset()
            return 0
        if inChan >= self.numIns:
            return 1
        if outChan >= self.numOuts:
            return 2
        if not self.owns(inChan, outChan) or self.engine is None:
            return 0
        return self.engine.reset(inChan - self.in_lo, outChan - self.out_lo)

    def clear(self, inChan: int, outChan: int, resize: bool = False):
        """clear(in, out, resize) = set(in, out, nullptr, 0, resize) (Convolver.cpp:55-58)"""
        return self.set(inChan, outChan, None, resize)

    # ---- streaming
    def process(self, ins: np.ndarray) -> np.ndarray:
        """ins: [numIns][n] float32 (every rank passes the same block).  Returns this rank's output rows
        [nout_local][n] — final in the rows layout, summed over the row group in the grid layout."""
        import torch
        ins = np.ascontiguousarray(ins, dtype=np.float32)
        n = ins.shape[1]
        outs = np.zeros((self.nout_local, n), np.float32)
        if self.engine is not None and self.nin_local > 0:
            local_in = np.ascontiguousarray(ins[self.in_lo:self.in_hi])
            self.engine.process(local_in, outs)
        if self.row_group is not None:
            t = torch.from_numpy(outs)
            backend = self.dist.get_backend(self.row_group)
            if backend == "nccl":                                   # RCCL wants device buffers
                t = t.cuda()
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.row_group)
                outs = t.cpu().numpy()
            else:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.row_group)
        return outs

    def process_dev(self, ins, outs=None):
        """HBM-resident variant: `ins` is a float32 CUDA tensor [numIns][n] on this rank's GPU (same on every rank);
        returns this rank's rows as a CUDA tensor [nout_local][n].  In the grid layout the partial blocks are summed
        with ONE RCCL all-reduce on the device buffer — the only collective on the data path."""
        import torch
        n = ins.shape[1]
        if outs is None:
            outs = torch.zeros((self.nout_local, n), dtype=torch.float32, device=ins.device)
        if self.engine is not None and self.nin_local > 0:
            local_in = ins[self.in_lo:self.in_hi].contiguous()
            torch.cuda.current_stream(ins.device).synchronize()          # the engine runs on its own streams
            self.engine.process_dev(local_in.data_ptr(), n, outs.data_ptr(), n, self.nin_local, self.nout_local, n, sync=True)
        if self.row_group is not None:
            self.dist.all_reduce(outs, op=self.dist.ReduceOp.SUM, group=self.row_group)
        return outs

    def gather(self, local_outs: np.ndarray) -> np.ndarray:
        """Assemble [numOuts][n] on every rank (optional; not part of the data path in the rows layout)."""
        import torch
        n = local_outs.shape[1]
        pieces = [None] * self.world
        self.dist.all_gather_object(pieces, (self.row, self.col, np.ascontiguousarray(local_outs)))
        full = np.zeros((self.numOuts, n), np.float32)
        for row, col, block in pieces:
            if col == 0 and block.shape[0]:
                lo, hi = split_range(self.numOuts, self.go, row)
                full[lo:hi] = block
        return full
