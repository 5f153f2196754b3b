// K1b: the hop-tiled spectral multiply-accumulate in its steady state — one process() call spans several hops of a stage
// (offline / batched calls, the short stages of a large block), every partition of every pair live.  Same sum as
// spectral_mac_kernel (hcv_mac.hip), which keeps the partition-bound checks and the single-hop shapes:
//
//   Y[ks][t][o][b] = sum over (i, p) in this block's k-slice of  X[i][(h_t - p) mod R][b] * H[o][i][p][b]
//
// (PartitionedConvolve.cpp:387-426 per partition and pair; NToMonoConvolve.cpp:39-42 for the sum over inputs).
//
// This kernel is bound by its instruction stream, not by HBM alone: TT hops share every IR spectrum, so a thread does
// OT x TT complex multiply-adds per 16-byte load.  What that needed (measured on c5, 16x16, P = 703, 8 hops per launch:
// 2.97 ms -> 2.05 ms, the IR spectra streamed once at 5.8 TB/s):
//   * four chained FMAs per complex product, written out (the compiler may not re-associate `c += a*b - d*e`);
//   * no SLP vectorisation (Makefile: -fno-slp-vectorize for this file): v_pk_fma_f32 pairs need register-pair shuffles and,
//     at the 4 x 8 tile, spill to scratch;
//   * the window of input spectra over the hop axis ROTATES through the registers instead of shifting (loop unrolled TT times);
//   * the loads of partition p + 1 are issued before the arithmetic of partition p (two register sets alternate);
//   * bin 0 = (DC, Nyquist) is handled through the owning lane's operands, not through side accumulators in every lane.
// Compiled on its own so that the flags above do not touch the single-hop kernel, which is HBM-bound and slightly faster with
// the packed forms.

#include "hcv_kernels.h"
#include "hcv_fft_device.h"
#include "hcv_mac_params.h"

#include <type_traits>

namespace hcv
{

// Hop-tiled steady-state variant (TT > 1, every partition live): the same arithmetic with the loads of partition p + 1 —
// OT IR spectra and the one input spectrum the window gains — issued BEFORE the OT x TT complex multiply-adds of partition p.
// A wave of the plain kernel alternates "wait for the loads" and "256 FMAs"; with 128 accumulator registers only two waves
// fit a SIMD, too few to cover the one phase with the other, so the launch sat at 0.56 of the HBM rate and 0.18 of the FMA
// rate, bound by neither.  Here a wave keeps one iteration of loads in flight under its own arithmetic.
template <int OT, int TT, bool NT>
__global__ __launch_bounds__(256, 2) void spectral_mac_tiled_kernel(MacParams a)
{
    int bx = blockIdx.x;
    if (a.pin >= 0)
    {
        if ((bx & 7) != a.pin) return;
        bx >>= 3;
    }
    const int bb = bx % a.binblocks;
    const int ks = bx / a.binblocks;
    const int o0 = blockIdx.y * OT;
    const int tile = blockIdx.z * blockDim.y + threadIdx.y;
    const bool tile_live = tile * TT < a.T;
    const int t0 = tile_live ? tile * TT : 0;
    const int live_t = min(TT, a.T - t0);
    const long long h0 = a.h_first + t0;
    const int hmod = (int) (h0 % a.R);

    const int b4 = bb * blockDim.x + threadIdx.x;
    const bool binlive = b4 < a.M2;
    const int b4c = binlive ? b4 : 0;
    const bool owns_bin0 = (b4 == 0);

    const int K = a.nin * a.P;
    const int kb0 = ks * a.kper;
    const int kb1 = min(K, kb0 + a.kper);

    float4 acc[TT][OT];
#pragma unroll
    for (int t = 0; t < TT; t++)
#pragma unroll
        for (int j = 0; j < OT; j++) acc[t][j] = make_float4(0.f, 0.f, 0.f, 0.f);

    const long long pair_stride4 = (long long) a.Pcap * a.M2;
    const long long out_stride4 = (long long) a.nin_alloc * pair_stride4;

    // Bin 0 carries (DC, Nyquist): two real products, acc.x += x.x * h.x and acc.y += x.y * h.y, instead of a complex one.
    // The one lane that owns it gets there through its OPERANDS — the term x.y * h.y of the real part loses its x.y, the
    // imaginary part's x.x * h.y loses its h.y and its x.y * h.x takes h.y — so no side accumulators are carried by every lane
    // (they cost the plain kernel 32 registers at this tile size) and no lane branches.
    // one partition: acc[t][j] += W[t] * hv[j], where the window W[t] = xw[(t - ROT) mod TT] is a ROTATING view of the
    // registers — the loop below is unrolled TT times with ROT fixed in each copy, so the window never moves through registers
    // (shifting it cost one v_mov per register and partition, on the instruction stream that bounds this kernel)
    auto mac_tile = [&](const float4 (&hv)[OT], const float4 (&xw)[TT], auto rot)
    {
        constexpr int ROT = decltype(rot)::value;
        float hb[OT], hc[OT];
#pragma unroll
        for (int j = 0; j < OT; j++)
        {
            hb[j] = owns_bin0 ? 0.f : hv[j].y;
            hc[j] = owns_bin0 ? hv[j].y : hv[j].x;
        }
#pragma unroll
        for (int t = 0; t < TT; t++)
        {
            const float4 &x = xw[(t + TT - ROT) % TT];
            const float xa = owns_bin0 ? 0.f : -x.y;
            const float nw = -x.w;
#pragma unroll
            for (int j = 0; j < OT; j++)
            {
                // four chained FMAs per complex product, written out: left to the compiler `c += a * b - d * e` becomes
                // multiply + FMA + add (it may not re-associate), half as much again on the instruction that bounds this kernel
                float4 &c = acc[t][j];
                c.x = __builtin_fmaf(x.x, hv[j].x, c.x);
                c.x = __builtin_fmaf(xa, hv[j].y, c.x);
                c.y = __builtin_fmaf(x.x, hb[j], c.y);
                c.y = __builtin_fmaf(x.y, hc[j], c.y);
                c.z = __builtin_fmaf(x.z, hv[j].z, c.z);
                c.z = __builtin_fmaf(nw, hv[j].w, c.z);
                c.w = __builtin_fmaf(x.z, hv[j].w, c.w);
                c.w = __builtin_fmaf(x.w, hv[j].z, c.w);
            }
        }
    };

    if (kb0 < kb1)
    {
        const int i_first = kb0 / a.P, i_last = (kb1 - 1) / a.P;
        for (int i = i_first; i <= i_last; i++)
        {
            const int pa = (i == i_first) ? kb0 - i_first * a.P : 0;
            const int pb = (i == i_last) ? kb1 - i_last * a.P : a.P;
            const float4 *xrow = a.X + (long long) (a.diag ? min(o0, a.nout - 1) : i) * a.R * a.M2;
            const float4 *hrow[OT];
#pragma unroll
            for (int j = 0; j < OT; j++) hrow[j] = a.H + (long long) min(o0 + j, a.nout - 1) * out_stride4 + (long long) i * pair_stride4;

            float4 xw[TT];                                        // xw[t] = X[h0 + t - pa] at ROT = 0
#pragma unroll
            for (int t = 0; t < TT; t++)
            {
                int slot = hmod + min(t, live_t - 1) - pa;
                if (slot < 0) slot += a.R;
                if (slot >= a.R) slot -= a.R;
                xw[t] = xrow[(unsigned) slot * (unsigned) a.M2 + (unsigned) b4c];
            }

            auto load_h = [&](float4 (&hv)[OT], int p)
            {
                const unsigned hoff = (unsigned) p * (unsigned) a.M2 + (unsigned) b4c;
#pragma unroll
                for (int j = 0; j < OT; j++) hv[j] = NT ? load_nt(hrow[j] + hoff) : hrow[j][hoff];
            };
            auto load_x = [&](int p) -> float4
            {
                int slot = hmod - p - 1;                          // the hop the window gains at p + 1
                if (slot < 0) slot += a.R;
                return xrow[(unsigned) slot * (unsigned) a.M2 + (unsigned) b4c];
            };

            // two register sets alternate: the one a step does not compute with is being loaded (TT is even)
            float4 hA[OT], hB[OT];
            float4 xA, xB;
            load_h(hA, pa);
            xA = load_x(pa);
            for (int p = pa; p < pb; p += TT)
            {
                auto step = [&](auto rot)
                {
                    constexpr int R_ = decltype(rot)::value;
                    if (p + R_ >= pb) return;                     // (wave-uniform)
                    float4 (&hc_)[OT] = (R_ & 1) ? hB : hA;
                    float4 (&hn_)[OT] = (R_ & 1) ? hA : hB;
                    float4 &xc_ = (R_ & 1) ? xB : xA;
                    float4 &xn_ = (R_ & 1) ? xA : xB;
                    if (p + R_ + 1 < pb)
                    {
                        load_h(hn_, p + R_ + 1);
                        xn_ = load_x(p + R_ + 1);
                    }
                    mac_tile(hc_, xw, rot);
                    xw[(TT - 1 - R_ + TT) % TT] = xc_;            // the oldest slot takes the new hop: W'[0]
                };
                if constexpr (TT >= 2) { step(std::integral_constant<int, 0>()); step(std::integral_constant<int, 1>()); }
                if constexpr (TT >= 4) { step(std::integral_constant<int, 2>()); step(std::integral_constant<int, 3>()); }
                if constexpr (TT >= 8)
                {
                    step(std::integral_constant<int, 4>()); step(std::integral_constant<int, 5>());
                    step(std::integral_constant<int, 6>()); step(std::integral_constant<int, 7>());
                }
            }
        }
    }

    if (tile_live && binlive)
    {
        float4 *y = a.Y + (long long) ks * a.ks_stride4 + b4;
#pragma unroll
        for (int t = 0; t < TT; t++)
            if (t < live_t)
            {
#pragma unroll
                for (int j = 0; j < OT; j++)
                    if (o0 + j < a.nout) y[((long long) (t0 + t) * a.nout + (o0 + j)) * a.M2] = acc[t][j];
            }
    }
}

template <int OT, int TT>
static hipError_t launch_tiled(bool nt, dim3 grid, dim3 block, const MacParams &a, hipStream_t st)
{
    if (nt)
        hipLaunchKernelGGL((spectral_mac_tiled_kernel<OT, TT, true>), grid, block, 0, st, a);
    else
        hipLaunchKernelGGL((spectral_mac_tiled_kernel<OT, TT, false>), grid, block, 0, st, a);
    return hipGetLastError();
}

hipError_t launch_mac_tiled(int ot, int tt, bool nt, dim3 grid, dim3 block, const MacParams &a, hipStream_t st)
{
    switch (ot * 16 + tt)
    {
        case 4 * 16 + 2: return launch_tiled<4, 2>(nt, grid, block, a, st);
        case 1 * 16 + 2: return launch_tiled<1, 2>(nt, grid, block, a, st);
        case 4 * 16 + 4: return launch_tiled<4, 4>(nt, grid, block, a, st);
        case 1 * 16 + 4: return launch_tiled<1, 4>(nt, grid, block, a, st);
        case 4 * 16 + 8: return launch_tiled<4, 8>(nt, grid, block, a, st);
        case 1 * 16 + 8: return launch_tiled<1, 8>(nt, grid, block, a, st);
        default: return hipErrorInvalidValue;
    }
}

// (Engine::init, once per device: HIP loads a translation unit's code object at the first launch of one of its kernels — 0.3 - 0.8 ms on the
// calling thread, which for the kernels of a control section or a restart is the audio thread in mid-stream; asking for a kernel's attributes loads it now)
void preload_mac_tiled()
{
    hipFuncAttributes fa;
    (void) hipFuncGetAttributes(&fa, reinterpret_cast<const void *>((spectral_mac_tiled_kernel<4, 8, true>)));
    (void) hipGetLastError();
}

} // namespace hcv
