// Stand-alone timing of spectral_mac launch shapes against the library's own kernels (links libhisstools_amd.so):
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I hisstools_library_amd/csrc tools/micro/mac_bench.cpp -L hisstools_library_amd -lhisstools_amd \
//         -Wl,-rpath,$PWD/hisstools_library_amd -o gpurun_out/mac_bench
//   mac_bench <nin> <nout> <P> <T> [reps] [M]
// Prints the launch plan, the average launch time and the bytes the plan moves (H once per hop tile, X once per output tile).
#include "hcv_kernels.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

// MAC_RANDOM=1: operands uniform in (-1, 1) instead of one constant (the chip clocks to its power budget: toggling operands cost the
// hop-tiled launch several per cent)
__global__ void fill_random(float *p, size_t n, unsigned seed)
{
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
    {
        unsigned h = (unsigned) i * 2654435761u ^ (unsigned) (i >> 32) * 40503u ^ seed;
        h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        p[i] = (float) (h >> 8) * (2.0f / 16777216.0f) - 1.0f;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int nin = argc > 1 ? std::atoi(argv[1]) : 16, nout = argc > 2 ? std::atoi(argv[2]) : 16, P = argc > 3 ? std::atoi(argv[3]) : 703;
    const int T = argc > 4 ? std::atoi(argv[4]) : 8, reps = argc > 5 ? std::atoi(argv[5]) : 10, M = argc > 6 ? std::atoi(argv[6]) : 8192;
    const int R = P + 2 * (T + 1);
    hcv::MacShape s;
    s.M = M; s.R = R; s.P = P; s.Pcap = P; s.nin = nin; s.nin_alloc = nin; s.nout = nout; s.diag = 0; s.T = T; s.max_ksplit = 64; s.target_blocks = 0;
    hcv::MacPlan pl;
    hcv::mac_plan(s, pl);
    const size_t hs = (size_t) nout * nin * P * M, xs = (size_t) nin * R * M, ys = (size_t) pl.ksplit * T * nout * M;
    float2 *H, *X, *Y;
    long long *hv;
    CK(hipMalloc(&H, hs * sizeof(float2)));
    CK(hipMalloc(&X, xs * sizeof(float2)));
    CK(hipMalloc(&Y, ys * sizeof(float2)));
    CK(hipMalloc(&hv, sizeof(long long) * nout * nin));
    CK(hipMemset(H, 0x3c, hs * sizeof(float2)));       // small finite floats
    CK(hipMemset(X, 0x3c, xs * sizeof(float2)));
    CK(hipMemset(hv, 0, sizeof(long long) * nout * nin));
    if (std::getenv("MAC_RANDOM") && std::atoi(std::getenv("MAC_RANDOM")))
    {
        hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (float *) H, hs * 2, 1u);
        hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (float *) X, xs * 2, 2u);
        CK(hipDeviceSynchronize());
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int k = 0; k < 3; k++) CK(hcv::launch_spectral_mac(s, pl, X, H, Y, hv, 100000 + k * T, false, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    for (int k = 0; k < reps; k++) CK(hcv::launch_spectral_mac(s, pl, X, H, Y, hv, 100000 + (3 + k) * T, false, st));
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    const int tiles = (T + pl.tt - 1) / pl.tt;
    const double bytes = 8.0 * M * P * nin * nout * tiles + 8.0 * M * (P + T) * nin * ((nout + pl.ot - 1) / pl.ot) + 8.0 * M * nout * T * pl.ksplit;
    const double flops = 8.0 * M * P * (double) nin * nout * T;
    std::printf("nin %d nout %d P %d T %d M %d | ot %d tt %d ksplit %d kper %d nt %d grid %dx%dx%d block %dx%d | %.4f ms  %.0f GB/s (%.3f of 8 TB/s)  %.1f TFLOP/s\n",
                nin, nout, P, T, M, pl.ot, pl.tt, pl.ksplit, pl.kper, pl.nt, pl.binblocks * pl.ksplit, pl.outtiles, pl.tz, pl.bx, pl.by, ms,
                bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 8e12, flops / (ms * 1e-3) / 1e12);
    return 0;
}
