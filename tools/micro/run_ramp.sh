mkdir -p gpurun_out
{
timeout 600 tools/micro/build/mac_mfma_fuzz 600 21
timeout 900 python -m pytest tests/test_offline_mfma_gpu.py tests/test_order_check_gpu.py -m gpu -q -x 2>&1 | tail -4
python - <<'PY'
import time, torch, numpy as np, sys
sys.path.insert(0, '.')
import hisstools_library_amd as H, bench
# offline convolution from silence: 16 x 16, 60 s @ 96 kHz IRs, 64-hop calls — the ramp-up IS the job for a file shorter than the IR
nin, nout, L, fs, layout = bench.WORKLOADS["c5"]
dev = torch.device("cuda", 0)
OB = 64 * 8192
c = H.Convolver(nin, nout, 0, device=0, maxBlock=OB, custom=(L, *layout))
g = torch.Generator(device=dev); g.manual_seed(1)
h = torch.rand(L, generator=g, device=dev) * 2 - 1
for o in range(nout):
    for i in range(nin):
        torch.cuda.synchronize(); assert c.set_dev(i, o, h.data_ptr(), L, True) == 0
x = torch.rand((nin, OB), device=dev) * 2 - 1; y = torch.zeros((nout, OB), device=dev)
for rep in range(2):
    c.reset(); c.synchronize(); torch.cuda.synchronize()
    ts = []
    for k in range(14):
        t0 = time.perf_counter(); c.process_dev(x.data_ptr(), OB, y.data_ptr(), OB, nin, nout, OB, sync=True); ts.append(1e3 * (time.perf_counter() - t0))
    print("ms per 64-hop call from silence:", [round(t, 1) for t in ts], "->", round(nout * OB * 11 / sum(ts[:11]) / 1e3, 1), "Msamples/s over the ramp-up (first 11 calls)")
PY
} 2>&1 | tee gpurun_out/ramp.log
