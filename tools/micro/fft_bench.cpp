// Stand-alone timing of the engine's single-workgroup transforms (links libhisstools_amd.so): back-to-back launches on one
// stream, so the figure is kernel duration + one boundary — what a small engine's block pays.
//   fft_bench <log2n> <transforms> [reps]
#include "hcv_kernels.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int log2n = argc > 1 ? std::atoi(argv[1]) : 14, nt = argc > 2 ? std::atoi(argv[2]) : 8, reps = argc > 3 ? std::atoi(argv[3]) : 200;
    const int N = 1 << log2n, M = N / 2;
    std::vector<float2> tw(M);
    for (int m = 0; m < M; m++) tw[m] = make_float2((float) std::cos(-M_PI * m / M), (float) std::sin(-M_PI * m / M));
    float2 *dtw, *X, *Y;
    float *hist, *in, *out;
    const long long hlen = 4LL * N;
    CK(hipMalloc(&dtw, sizeof(float2) * M));
    CK(hipMemcpy(dtw, tw.data(), sizeof(float2) * M, hipMemcpyHostToDevice));
    CK(hipMalloc(&X, sizeof(float2) * (size_t) nt * 4 * M));
    CK(hipMalloc(&Y, sizeof(float2) * (size_t) nt * M));
    CK(hipMalloc(&hist, sizeof(float) * nt * hlen));
    CK(hipMalloc(&in, sizeof(float) * nt * M));
    CK(hipMalloc(&out, sizeof(float) * nt * M));
    CK(hipMemset(hist, 0, sizeof(float) * nt * hlen));
    CK(hipMemset(in, 0x3c, sizeof(float) * nt * M));
    CK(hipMemset(Y, 0x3c, sizeof(float2) * (size_t) nt * M));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float ms;
    for (int which = 0; which < 3; which++)
    {
        for (int k = 0; k < reps + 10; k++)
        {
            if (k == 10) CK(hipEventRecord(a, st));
            if (which == 0) CK(hcv::launch_rfft_frames_direct(log2n, hist, hlen, hlen - 1, in, M, 4LL * M, 4, 1, nt, X, 4, dtw, st));
            if (which == 1) CK(hcv::launch_rifft_emit(log2n, Y, 1, 0, 1, nt, out, M, dtw, st));
            if (which == 2) CK(hcv::launch_rfft_frames(log2n, hist, hlen, hlen - 1, 4, 1, nt, X, 4, dtw, nullptr, st));
        }
        CK(hipEventRecord(b, st));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
        const char *names[3] = { "rfft_frames_direct", "rifft_emit", "rfft_frames" };
        std::printf("N %d x %d  %-20s %.2f us per launch\n", N, nt, names[which], 1e3 * ms / reps);
    }
    return 0;
}
