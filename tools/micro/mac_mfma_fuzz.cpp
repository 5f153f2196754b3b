// Randomised comparison of the offline multiply-accumulate on the matrix cores (hcv_mac_mfma.hip) with the register-tiled kernels of the same
// library over shapes the engine can hand it: bins 16 .. 2048, 1 .. 20 inputs, 2 .. 40 outputs (ragged output tiles), 1 .. 60 partitions
// (k-slices that end inside an input, chunks of fewer than 16 partitions), 32 .. 150 hops (ragged hop tiles), ring lengths and first hops at random;
// one case in three is a ramp-up (a uniform first hop inside the partitions' reach: MacShape::hop_min against the checked register tiles).
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I hisstools_library_amd/csrc tools/micro/mac_mfma_fuzz.cpp -L hisstools_library_amd -lhisstools_amd \
//         -Wl,-rpath,$PWD/hisstools_library_amd -o tools/micro/build/mac_mfma_fuzz
//   mac_mfma_fuzz [cases] [seed]        exit 0 = every case within 1e-5 of the peak (two f32 evaluation orders; the checked register tiles round a product differently)
#include "hcv_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
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
    const int cases = argc > 1 ? std::atoi(argv[1]) : 200;
    std::mt19937 g(argc > 2 ? (unsigned) std::atoi(argv[2]) : 1u);
    auto pick = [&](int lo, int hi) { return lo + (int) (g() % (unsigned) (hi - lo + 1)); };
    const size_t cap_h = size_t(1) << 27, cap_x = size_t(1) << 25, cap_y = size_t(1) << 26;      // float2 elements
    float2 *H, *X, *Yr, *Ym;
    long long *hv;
    CK(hipMalloc(&H, cap_h * sizeof(float2)));
    CK(hipMalloc(&X, cap_x * sizeof(float2)));
    CK(hipMalloc(&Yr, cap_y * sizeof(float2)));
    CK(hipMalloc(&Ym, cap_y * sizeof(float2)));
    CK(hipMalloc(&hv, sizeof(long long) * 4096));
    CK(hipMemset(hv, 0, sizeof(long long) * 4096));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    std::vector<float> r, m;
    int done = 0, bad = 0;
    double worst = 0.0;
    for (int c = 0; c < cases; c++)
    {
        const int Ms[] = { 16, 32, 64, 128, 256, 1024, 2048 };
        hcv::MacShape s;
        s.M = Ms[pick(0, 6)];
        s.nin = pick(1, 20);
        s.nout = pick(2, 40);
        s.P = pick(1, 60);
        if ((long long) s.nin * s.P < 16) s.P = (16 + s.nin - 1) / s.nin;
        s.T = pick(32, 150);
        s.Pcap = s.P + pick(0, 3);
        s.nin_alloc = s.nin + pick(0, 2);
        s.R = s.P + 2 * (s.T + 1) + pick(0, 9);
        s.diag = 0;
        s.max_ksplit = pick(1, 8);
        s.target_blocks = 0;
        s.ot_cap = 0;
        s.steady = 0;
        hcv::MacShape sm = s;
        sm.steady = 1;
        hcv::MacPlan pr, pm;
        hcv::mac_plan(s, pr);
        hcv::mac_plan(sm, pm);
        const size_t hs = (size_t) s.nout * s.nin_alloc * s.Pcap * s.M, xs = (size_t) s.nin * s.R * s.M;
        const size_t per = (size_t) s.T * s.nout * s.M, yr = per * pr.ksplit, ym = per * pm.ksplit;
        if (!pm.mfma || hs > cap_h || xs > cap_x || yr > cap_y || ym > cap_y) continue;
        hipLaunchKernelGGL(fill_random, dim3(1024), dim3(256), 0, st, (float *) H, hs * 2, (unsigned) g());
        hipLaunchKernelGGL(fill_random, dim3(1024), dim3(256), 0, st, (float *) X, xs * 2, (unsigned) g());
        CK(hipMemsetAsync(Ym, 0xff, ym * sizeof(float2), st));
        const long long h_first = 1000 + pick(0, 100000);
        // one case in three: a ramp-up — every pair's first hop is hv_val, somewhere inside the reach of the launch's partitions; the register
        // tiles apply it as per-pair bounds (check = true), the matrix-core kernel stages earlier hops as zeros (hop_min)
        const bool ramp = pick(0, 2) == 0;
        const long long hv_val = ramp ? h_first + s.T - 1 - pick(0, s.P + s.T) : 0;
        if (ramp)
        {
            std::vector<long long> hh(4096, hv_val);
            CK(hipMemcpyAsync(hv, hh.data(), sizeof(long long) * 4096, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            sm.hop_min = hv_val;
        }
        CK(hcv::launch_spectral_mac(s, pr, X, H, Yr, hv, h_first, ramp, st));
        CK(hcv::launch_spectral_mac(sm, pm, X, H, Ym, hv, h_first, false, st));
        CK(hipStreamSynchronize(st));
        r.resize(yr * 2);
        m.resize(ym * 2);
        CK(hipMemcpy(r.data(), Yr, yr * sizeof(float2), hipMemcpyDeviceToHost));
        CK(hipMemcpy(m.data(), Ym, ym * sizeof(float2), hipMemcpyDeviceToHost));
        double err = 0, peak = 0;
        for (size_t e = 0; e < per * 2; e++)
        {
            double vr = 0, vm = 0;
            for (int k = 0; k < pr.ksplit; k++) vr += r[(size_t) k * per * 2 + e];
            for (int k = 0; k < pm.ksplit; k++) vm += m[(size_t) k * per * 2 + e];
            peak = std::max(peak, std::fabs(vr));
            if (e == 0 && peak == 0) peak = 1e-30;
            if (!(std::fabs(vr - vm) <= err)) err = std::fabs(vr - vm);
        }
        const double rel = err / peak;
        done++;
        worst = std::max(worst, rel);
        if (!(rel < 1e-5))         // (SURVEY 8c's bound; a missing or misplaced term would show at 1e-2: these sums have ~1000 terms of unit size)
        {
            bad++;
            std::printf("MISMATCH case %d: M %d nin %d(+%d) nout %d P %d(cap %d) T %d R %d ksplit %d/%d mt %d: %.3e of peak\n", c, s.M, s.nin, s.nin_alloc - s.nin, s.nout,
                        s.P, s.Pcap, s.T, s.R, pr.ksplit, pm.ksplit, pm.mfma, rel);
        }
    }
    std::printf("%d cases compared (of %d drawn), %d mismatches, worst %.3e of the peak\n", done, cases, bad, worst);
    return bad ? 1 : 0;
}
