#!/usr/bin/env python3
"""Phase times of the four-step passes (diagnostic build -DHCV_FX_PHASE_TIMING of hcv_fftx.hip installed as the library):
average, per workgroup, time from its start to the end of the load / transform / store phase, per pass.
    python tools/micro/fx_phases.py 16 18 20 22"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hisstools_library_amd as H  # noqa: E402
import hisstools_library_amd.fft as F  # noqa: E402

lib = H.load()
dbg = lib.hcv_debug_fx_phases
dbg.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * 8)()
for l2 in [int(a) for a in sys.argv[1:]] or [20]:
    m = 1 << l2
    batch = (1 << 30) // (8 * m)
    a = torch.rand(batch * m, device="cuda") * 2 - 1
    b = torch.rand(batch * m, device="cuda") * 2 - 1
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(2):
        torch.cuda.synchronize()
        dbg(buf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        F.exec_dev(F.Op.FFT, F.Precision.F32, l2, batch, a.data_ptr(), b.data_ptr(), a.data_ptr(), b.data_ptr(), m, m, 0, st, False)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        dbg(buf)
    v = list(buf)
    nc, nr = max(1, v[3]), max(1, v[7])
    print(f"2^{l2} complex f32, batch {batch}: {ms:.3f} ms ({2 * batch * 8 * m / ms / 1e6:.0f} GB/s) | cols: {nc} workgroups, load {v[0] / nc / 100:.2f} us, "
          f"+transform {v[1] / nc / 100:.2f}, +store {v[2] / nc / 100:.2f} | rows: {nr} workgroups, load {v[4] / nr / 100:.2f} us, +transform {v[5] / nr / 100:.2f}, +store {v[6] / nr / 100:.2f}")
