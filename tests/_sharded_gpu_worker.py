"""Worker for tests/test_sharded_gpu.py: torch.distributed with the nccl (= RCCL) backend on real GPUs.
World size = number of visible GPUs (1 on the test box): exercises RCCL initialisation, the HIP engine per rank, the
device-resident process path and — in the grid layout — the all-reduce of the partial output block."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hisstools_library_amd.sharded import ShardedConvolver, split_range  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    layout = sys.argv[1]
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    rank, world = dist.get_rank(), dist.get_world_size()
    nin, nout, L, S, B = 4, 4, 9000, 16384, 4096
    irs = {(i, o): O.synth_ir(i, o, L) for i in range(nin) for o in range(nout)}
    xs = np.stack([O.synth_audio(i, S) for i in range(nin)])
    full = O.Convolver(nin, nout, 0)
    full.setResetOffset(0)
    for (i, o), h in irs.items():
        assert full.set(i, o, h, True) == 0
    expect = full.run(xs, nout, 2048)

    sc = ShardedConvolver(nin, nout, 0, layout=layout, grid=(1, world) if layout == "grid" else None, device=local)
    for (i, o), h in irs.items():
        assert sc.set(i, o, h, True) == 0
    xd = torch.from_numpy(xs).to(dev)
    got = torch.zeros((sc.nout_local, S), dtype=torch.float32, device=dev)
    for pos in range(0, S, B):
        got[:, pos:pos + B] = sc.process_dev(xd[:, pos:pos + B].contiguous())
    lo, hi = split_range(nout, sc.go, sc.row)
    peak = np.abs(expect).max()
    err = np.abs(got.cpu().numpy() - expect[lo:hi]).max() / peak
    assert err < 1e-5, (rank, err)
    # the host-pointer path of the same object
    sc.reset()
    host = np.concatenate([sc.process(xs[:, pos:pos + B]) for pos in range(0, S, B)], axis=1)
    assert np.abs(host - expect[lo:hi]).max() / peak < 1e-5
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} ok ({layout}, err {err:.2e})")


if __name__ == "__main__":
    main()
