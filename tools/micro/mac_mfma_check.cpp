// Stand-alone check + timing of the offline multiply-accumulate on the matrix cores (hcv_mac_mfma.hip) against the register-tiled
// kernels of the same library, on random operands:
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I hisstools_library_amd/csrc tools/micro/mac_mfma_check.cpp -L hisstools_library_amd -lhisstools_amd \
//         -Wl,-rpath,$PWD/hisstools_library_amd -o tools/micro/build/mac_mfma_check
//   mac_mfma_check <nin> <nout> <P> <T> [reps] [M] [verify 0/1]
#include "hcv_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

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
    const int nin = argc > 1 ? std::atoi(argv[1]) : 16, nout = argc > 2 ? std::atoi(argv[2]) : 16, P = argc > 3 ? std::atoi(argv[3]) : 704;
    const int T = argc > 4 ? std::atoi(argv[4]) : 64, reps = argc > 5 ? std::atoi(argv[5]) : 5, M = argc > 6 ? std::atoi(argv[6]) : 8192;
    const int verify = argc > 7 ? std::atoi(argv[7]) : 1;
    const int R = P + 2 * (T + 1) + 3;
    hcv::MacShape s;
    s.M = M; s.R = R; s.P = P; s.Pcap = P + 1; s.nin = nin; s.nin_alloc = nin; s.nout = nout; s.diag = 0; s.T = T; s.max_ksplit = 8; s.target_blocks = 0;
    s.ot_cap = 0;
    hcv::MacShape sm = s;
    sm.steady = 1;
    hcv::MacPlan pr, pm;
    hcv::mac_plan(s, pr);
    hcv::mac_plan(sm, pm);
    if (!pm.mfma) { std::printf("shape not taken by the MFMA plan\n"); return 2; }
    const size_t hs = (size_t) nout * nin * s.Pcap * M, xs = (size_t) nin * R * M;
    const size_t yr = (size_t) pr.ksplit * T * nout * M, ym = (size_t) pm.ksplit * T * nout * M;
    float2 *H, *X, *Yr, *Ym;
    long long *hv;
    CK(hipMalloc(&H, hs * sizeof(float2)));
    CK(hipMalloc(&X, xs * sizeof(float2)));
    CK(hipMalloc(&Yr, yr * sizeof(float2)));
    CK(hipMalloc(&Ym, ym * sizeof(float2)));
    CK(hipMalloc(&hv, sizeof(long long) * nout * nin));
    CK(hipMemset(hv, 0, sizeof(long long) * nout * nin));
    CK(hipMemset(Ym, 0xff, ym * sizeof(float2)));
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (float *) H, hs * 2, 1u);
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (float *) X, xs * 2, 2u);
    CK(hipDeviceSynchronize());
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const long long h_first = 100003;
    std::printf("nin %d nout %d P %d T %d M %d | tiled: ot %d tt %d ksplit %d | mfma: mt %d ksplit %d kper %d grid %dx%dx%d\n", nin, nout, P, T, M, pr.ot, pr.tt,
                pr.ksplit, pm.mfma, pm.ksplit, pm.kper, pm.binblocks * pm.ksplit, pm.outtiles, pm.tz);
    if (verify)
    {
        CK(hcv::launch_spectral_mac(s, pr, X, H, Yr, hv, h_first, false, st));
        CK(hcv::launch_spectral_mac(sm, pm, X, H, Ym, hv, h_first, false, st));
        CK(hipStreamSynchronize(st));
        std::vector<float> r(yr * 2), m(ym * 2);
        CK(hipMemcpy(r.data(), Yr, yr * sizeof(float2), hipMemcpyDeviceToHost));
        CK(hipMemcpy(m.data(), Ym, ym * sizeof(float2), hipMemcpyDeviceToHost));
        const size_t per = (size_t) T * nout * M * 2;
        double worst = 0, peak = 0;
        size_t where = 0;
        for (size_t e = 0; e < per; e++)
        {
            double vr = 0, vm = 0;
            for (int k = 0; k < pr.ksplit; k++) vr += r[(size_t) k * per + e];
            for (int k = 0; k < pm.ksplit; k++) vm += m[(size_t) k * per + e];
            peak = std::max(peak, std::fabs(vr));
            if (!(std::fabs(vr - vm) <= worst)) { worst = std::fabs(vr - vm); where = e; }
        }
        const size_t b2 = where % (2 * (size_t) M), o = (where / (2 * (size_t) M)) % nout, t = where / (2 * (size_t) M * nout);
        std::printf("  verify: max |tiled - mfma| = %.3e at hop %zu output %zu bin %zu.%zu, peak %.3e -> %.2e of peak %s\n", worst, t, o, b2 / 2, b2 & 1, peak,
                    worst / peak, worst / peak < 2e-6 ? "OK" : "MISMATCH");
    }
    for (int pass = 0; pass < 2; pass++)
    {
        const hcv::MacShape &sx = pass ? sm : s;
        const hcv::MacPlan &px = pass ? pm : pr;
        float2 *Y = pass ? Ym : Yr;
        for (int k = 0; k < 2; k++) CK(hcv::launch_spectral_mac(sx, px, X, H, Y, hv, h_first + k, false, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(a, st));
        for (int k = 0; k < reps; k++) CK(hcv::launch_spectral_mac(sx, px, X, H, Y, hv, h_first + 2 + k, false, st));
        CK(hipEventRecord(b, st));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        ms /= reps;
        const double flops = 8.0 * M * P * (double) nin * nout * T;
        const double hbytes = 8.0 * M * P * (double) nin * nout * (pass ? px.tz : (T + px.tt - 1) / px.tt);
        std::printf("  %s: %.4f ms  %.1f TFLOP/s (%.3f of 157.3)  IR stream %.0f GB/s  -> %.0f out-Msamples/s\n", pass ? "mfma " : "tiled", ms,
                    flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 157.3e12, hbytes / (ms * 1e-3) / 1e9, (double) T * M * nout / (ms * 1e-3) / 1e6);
    }
    return 0;
}
