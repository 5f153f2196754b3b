// A library-free programme with the create / record / wait / destroy pattern an engine of round 4 had (is the abort inside
// hipStreamDestroy the runtime's, or this library's?).  Per cycle: S non-blocking streams and E events are made, small kernels run on every
// stream with event records and cross-stream waits between them, everything is synchronized, then — depending on `mode` —
//   mode 0: events destroyed first, then the streams                          (what a tidy host does)
//   mode 1: streams destroyed first, the events QUERIED and destroyed after    (an event that outlives the stream it was recorded on: what the
//           engine did with the control arena's parked blocks before round 5's fix)
//   mode 2: as 1, and a second thread creates / destroys streams of its own meanwhile (a finalizer thread beside the main one)
//   hipcc --offload-arch=gfx950 -O2 tools/micro/stream_destroy_repro.hip -o tools/micro/build/stream_destroy_repro -lpthread
//   MALLOC_CHECK_=3 MALLOC_PERTURB_=165 stream_destroy_repro <mode> <cycles> [streams] [events]
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void tiny(float *p, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static int cycle(int mode, int S, int E, float *buf)
{
    std::vector<hipStream_t> st((size_t) S);
    std::vector<hipEvent_t> ev((size_t) E);
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int k = 0; k < E; k++)
    {
        hipStream_t a = st[(size_t) (k % S)], b = st[(size_t) ((k * 7 + 3) % S)];
        hipLaunchKernelGGL(tiny, dim3(4), dim3(256), 0, a, buf + 1024 * (k % S), 1024);
        CK(hipEventRecord(ev[(size_t) k], a));
        CK(hipStreamWaitEvent(b, ev[(size_t) k], 0));
        hipLaunchKernelGGL(tiny, dim3(4), dim3(256), 0, b, buf + 1024 * ((k * 7 + 3) % S), 1024);
    }
    for (auto &s : st) CK(hipStreamSynchronize(s));
    if (mode == 0)
    {
        for (auto &e : ev) CK(hipEventDestroy(e));
        for (auto &s : st) CK(hipStreamDestroy(s));
    }
    else
    {
        for (auto &s : st) CK(hipStreamDestroy(s));
        for (auto &e : ev)
        {
            (void) hipEventQuery(e);
            CK(hipEventDestroy(e));
        }
    }
    return 0;
}

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? std::atoi(argv[1]) : 1, cycles = argc > 2 ? std::atoi(argv[2]) : 1000;
    const int S = argc > 3 ? std::atoi(argv[3]) : 12, E = argc > 4 ? std::atoi(argv[4]) : 40;
    float *buf = nullptr;
    CK(hipMalloc(&buf, sizeof(float) * 1024 * (size_t) S * 2));
    CK(hipMemset(buf, 0, sizeof(float) * 1024 * (size_t) S * 2));
    std::atomic<bool> stop { false };
    std::atomic<int> side_cycles { 0 };
    std::thread side;
    if (mode == 2)
        side = std::thread([&]()
        {
            while (!stop.load())
            {
                if (cycle(1, 3, 6, buf + 1024 * S)) break;
                side_cycles++;
            }
        });
    int rc = 0;
    for (int c = 0; c < cycles && !rc; c++) rc = cycle(mode == 2 ? 1 : mode, S, E, buf);
    stop.store(true);
    if (side.joinable()) side.join();
    CK(hipDeviceSynchronize());
    std::printf("mode %d: %d cycles of %d streams / %d events%s: %s\n", mode, cycles, S, E, mode == 2 ? " beside a second thread" : "", rc ? "FAILED" : "clean");
    return rc;
}
