"""N > 1 path on CPU: world_size-2 gloo runs of the output-row sharding (no collective) and of the input-split
layout (one all-reduce per call).  The per-rank engine is the CPU oracle; on the GPU box the same class drives the
HIP engine with the nccl (= RCCL) backend."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("layout", ["rows", "grid"])
def test_two_rank_sharding(layout):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
           str(free_port()), os.path.join(ROOT, "tests", "_sharded_worker.py"), layout]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count(" ok (") == 2


def test_split_range_covers_everything():
    from hisstools_library_amd.sharded import split_range
    for n in (1, 7, 16, 64):
        for parts in (1, 2, 3, 8):
            blocks = [split_range(n, parts, k) for k in range(parts)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[k][1] == blocks[k + 1][0] for k in range(parts - 1))
            assert max(hi - lo for lo, hi in blocks) - min(hi - lo for lo, hi in blocks) <= 1
