import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import hisstools_library_amd as H
import bench
B_ENV = int(os.environ.get("BLOCK", "8192"))
for w in sys.argv[1:]:
    nin, nout, L, fs, layout = bench.WORKLOADS[w]
    dev = torch.device("cuda", 0)
    conv = H.Convolver(nin, nout, 0, device=0, maxBlock=8192, custom=(L, *layout))
    h = torch.rand(L, device=dev) * 2 - 1
    for o in range(nout):
        for i in range(nin):
            torch.cuda.synchronize(); assert conv.set_dev(i, o, h.data_ptr(), L, True) == 0
    B = B_ENV
    xs = torch.rand((nin, B), device=dev); ys = torch.zeros((nout, B), device=dev)
    for _ in range(100): conv.process_dev(xs.data_ptr(), B, ys.data_ptr(), B, nin, nout, B)
    conv.synchronize()
    N = 2000
    t0 = time.perf_counter()
    for _ in range(N): conv.process_dev(xs.data_ptr(), B, ys.data_ptr(), B, nin, nout, B)
    t1 = time.perf_counter()
    conv.synchronize()
    t2 = time.perf_counter()
    print(f"{w}: enqueue {1e6*(t1-t0)/N:.1f} us/block, total {1e6*(t2-t0)/N:.1f} us/block")
