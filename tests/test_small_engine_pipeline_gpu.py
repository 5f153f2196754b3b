"""The two-stream pipeline of small engines (hcv_engine_block.hip: HCV_PIPE2 auto, sparse end events, the far wait).

Asynchronous hop-sized calls of an engine whose tail transforms are 16384 points long put the next block's forward transforms on
a second stream beside the current block's multiply-accumulate and inverse (MonoConvolve.cpp:179-201 / PartitionedConvolve.cpp:
243-385 are the reference's per-block chain; the pipeline changes when kernels run, never what they compute).  A caller that
waits for every block takes the serial chain.  Same kernels on the same data: the two schedules must agree BIT FOR BIT, and both
must equal the float64 truth within the float32 tolerance of SURVEY 8c.  Long enough runs (>= 40 blocks) reach the state the
bench measures: every second block records its end event and the transforms wait for an end three or four blocks back.
"""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = 2e-6
B = 8192


@pytest.fixture(scope="module")
def H():
    import hisstools_library_amd as H
    assert H.load().hcv_device_count() > 0, "no GPU visible: the HIP path cannot run (and there is no fallback)"
    return H


@pytest.fixture(scope="module")
def torch():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    return torch


def _truth(x, h, n):
    m = 1 << int(np.ceil(np.log2(n + h.shape[-1])))
    return np.fft.irfft(np.fft.rfft(x.astype(np.float64), m) * np.fft.rfft(h.astype(np.float64), m), m)[:n]


@pytest.mark.parametrize("nin,L,layout,latency", [
    (8, 60000, (True, 256, 1024, 4096, 16384), 0),        # config 3's shape (8 -> 1, zero latency), shorter IRs
    (1, 48000, (False, 16384, 0, 0, 0), 8192),            # config 1: MonoConvolve(48000, false, 16384)
    (3, 20000, (True, 256, 1024, 4096, 16384), 0),        # a tail of two partitions: a short second half of the chain
])
def test_pipelined_blocks_equal_serial_blocks_and_the_truth(H, torch, nin, L, layout, latency):
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(1234 + nin)
    blocks = 44
    n = blocks * B
    x = rng.uniform(-1, 1, size=(nin, n)).astype(np.float32)
    h = (rng.uniform(-1, 1, size=(nin, L)) * np.power(10.0, -3.0 * np.arange(L) / L)).astype(np.float32)
    c = H.Convolver(nin, 1, 0, custom=(L, *layout), maxBlock=B)
    hd = torch.from_numpy(h).to(dev)
    for i in range(nin):
        torch.cuda.synchronize()
        assert c.set_dev(i, 0, hd[i].data_ptr(), L, True) == 0
    xs = torch.from_numpy(x).to(dev)
    outs = {}
    for mode in ("async", "sync", "mixed"):
        c.reset()
        ys = torch.zeros((1, n), device=dev, dtype=torch.float32)
        for k in range(blocks):
            # mixed: a waiting call every seventh block drops out of the pipeline and the run re-enters it
            wait = mode == "sync" or (mode == "mixed" and k % 7 == 3)
            c.process_dev(xs.data_ptr() + 4 * k * B, n, ys.data_ptr() + 4 * k * B, n, nin, 1, B, sync=wait)
        c.synchronize()
        outs[mode] = ys[0].cpu().numpy()
    assert np.array_equal(outs["async"], outs["sync"])
    assert np.array_equal(outs["mixed"], outs["sync"])
    truth = sum(_truth(x[i], h[i], n) for i in range(nin))
    y = outs["async"]
    if latency:
        assert not y[:latency].any()
        assert rel_err(y[latency:], truth[:n - latency]) <= TOL
    else:
        assert rel_err(y, truth) <= TOL
