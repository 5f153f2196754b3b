// K1: spectral multiply-accumulate — the HBM-bound roofline kernel of the partitioned convolution.
//
//   Y[ks][t][o][b] = sum over (i, p) in this block's k-slice of  X[i][(h_t - p) mod R][b] * H[o][i][p][b],   h_t = h_first + t
//
// replaces PartitionedConvolve::processPartition (PartitionedConvolve.cpp:387-426) called P times per hop per
// (in,out) pair, plus the accumulation over inputs that NToMonoConvolve::process does in the time domain
// (NToMonoConvolve.cpp:39-42).
//
// Register tile per thread:  2 bins (one 16-byte load)  x  OT outputs  x  TT hops.
//   * the OT outputs share the input spectrum X in registers  -> X traffic is 1/OT of H traffic
//   * the TT hops share the IR spectrum H in registers        -> when a process() call spans several hops of a stage
//     (short stages, or batched/offline calls) H is read once per TT hops.  Along p the X operands of consecutive
//     hops form a sliding window (hop t at partition p needs spectrum h_t - p), so each p step loads ONE new X
//     value and shifts the window — an FIR over the hop axis.
//   * H is streamed with nontemporal 16-byte loads when every element is used once per launch, keeping the caches
//     for X.
// The reduction over (input, partition) stays in registers; long reductions are split over blockIdx.x (split-K,
// summed by reduce_partials_kernel).  For short spectra (< 256 float4 per spectrum) the spare threads of the
// workgroup (threadIdx.y) take further hop tiles.
//
// bin 0 carries (DC, Nyquist) and needs two real products instead of a complex one
// (PartitionedConvolve.cpp:398-406, 424-425): the one lane that owns bin 0 tracks the Nyquist products in a side
// accumulator and repairs its bin after the loop.
//
// hv[o][i] is the first hop whose input a pair may see (per-pair reset); CHECK=false is the steady state where
// every pair sees all P partitions.

#include "hcv_kernels.h"
#include "hcv_fft_device.h"
#include "hcv_mac_params.h"
#include "hcv_order_check.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#ifndef HCV_MAC_UNROLL
#define HCV_MAC_UNROLL 1
#endif

namespace hcv
{

__device__ __forceinline__ void cmac2(float4 &acc, const float4 &x, const float4 &h)
{
    acc.x += x.x * h.x - x.y * h.y;
    acc.y += x.x * h.y + x.y * h.x;
    acc.z += x.z * h.z - x.w * h.w;
    acc.w += x.z * h.w + x.w * h.z;
}

//
// INWG: split-K INSIDE the workgroup, for the engines whose whole reduction is a few hundred terms (8 -> 1 with 5 s IRs: 240):
// the workgroup is 64 lanes (128 bins, one 1 KiB run per load) x blockDim.y k-slices, one wave each; the slices' sums meet in LDS
// and are added up in slice order by the first wave, which writes the finished spectrum — no partial sums through memory and no
// reduce_partials launch (4.4 us of a 27 us chain).
template <int OT, int TT, bool CHECK, bool NT, bool INWG = false>
__global__ __launch_bounds__(INWG ? 1024 : 256) void spectral_mac_kernel(MacParams a)
{
    int bx = blockIdx.x;
    if (a.pin >= 0)
    {
        if ((bx & 7) != a.pin) return;
        bx >>= 3;
    }
    const int bb = INWG ? bx : bx % a.binblocks;
    const int ks = INWG ? (int) threadIdx.y : bx / a.binblocks;
    const int o0 = blockIdx.y * OT;
    const int tile = INWG ? (int) blockIdx.z : (int) (blockIdx.z * blockDim.y + threadIdx.y);      // hop tile of this thread row
    const bool tile_live = tile * TT < a.T;
    const int t0 = tile_live ? tile * TT : 0;
    const int live_t = min(TT, a.T - t0);                        // hops of the tile that exist
    const long long h0 = a.h_first + t0;
    const int hmod = (int) (h0 % a.R);

    const int b4 = bb * blockDim.x + threadIdx.x;                // float4 index inside the spectrum
    const bool binlive = b4 < a.M2;
    const int b4c = binlive ? b4 : 0;
    const bool owns_bin0 = (b4 == 0);

    // this block's k-slice [kb0, kb1) of the flattened (i, p) reduction
    const int K = a.nin * a.P;
    const int kb0 = ks * a.kper;
    const int kb1 = min(K, kb0 + a.kper);

    float4 acc[TT][OT];
    float ny[TT][OT];                                            // bin 0 only: sum of the Nyquist products x.y * h.y
#pragma unroll
    for (int t = 0; t < TT; t++)
#pragma unroll
        for (int j = 0; j < OT; j++)
        {
            acc[t][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            ny[t][j] = 0.f;
        }

    const long long pair_stride4 = (long long) a.Pcap * a.M2;
    const long long out_stride4 = (long long) a.nin_alloc * pair_stride4;


    if (kb0 < kb1)
    {
        const int i_first = kb0 / a.P, i_last = (kb1 - 1) / a.P;
        for (int i = i_first; i <= i_last; i++)
        {
            const int pa = (i == i_first) ? kb0 - i_first * a.P : 0;
            const int pb = (i == i_last) ? kb1 - i_last * a.P : a.P;

            int lim[OT];
            if (CHECK)
            {
#pragma unroll
                for (int j = 0; j < OT; j++)
                {
                    long long d = h0 - a.hv[(long long) min(o0 + j, a.nout - 1) * a.nin_alloc + i];
                    lim[j] = d > 0x3fffffff ? 0x3fffffff : (d < -0x3fffffff ? -0x3fffffff : (int) d);
                }
            }

            // diag (parallel) mode runs with OT == 1: the input row is the output's own
            const float4 *xrow = a.X + (long long) (a.diag ? min(o0, a.nout - 1) : i) * a.R * a.M2;
            // wave-uniform 64-bit bases per (output, input); the per-lane part of every address is a 32-bit offset
            // (one pair's spectra and one input's ring are far below 2^32 float4).  Dead outputs of a ragged last
            // tile re-read the last live one.
            const float4 *hrow[OT];
#pragma unroll
            for (int j = 0; j < OT; j++) hrow[j] = a.H + (long long) min(o0 + j, a.nout - 1) * out_stride4 + (long long) i * pair_stride4;

            // small tiles (one or two outputs, one hop: the n x 1 and 1 x 1 engines) are latency-bound chains of short k-slices: with
            // eight partitions unrolled their sixteen loads are in flight together; the large tiles keep their registers
            constexpr int kUnroll = (OT <= 2 && TT == 1) ? 8 : HCV_MAC_UNROLL;
            // sliding window over the hop axis: xw[t] = X[h0 + t - p]
            float4 xw[TT];
            if (TT > 1)
            {
#pragma unroll
                for (int t = 0; t < TT; t++)
                {
                    int slot = hmod + min(t, live_t - 1) - pa;
                    if (slot < 0) slot += a.R;
                    if (slot >= a.R) slot -= a.R;
                    xw[t] = xrow[(unsigned) slot * (unsigned) a.M2 + (unsigned) b4c];
                }
            }

#pragma unroll kUnroll
            for (int p = pa; p < pb; p++)
            {
                float4 hval[OT];
                const unsigned hoff = (unsigned) p * (unsigned) a.M2 + (unsigned) b4c;
#pragma unroll
                for (int j = 0; j < OT; j++) hval[j] = NT ? load_nt(hrow[j] + hoff) : hrow[j][hoff];

                float4 xn;
                {
                    int slot = hmod - p - (TT > 1 ? 1 : 0);      // TT > 1: the hop the window gains at p + 1
                    if (slot < 0) slot += a.R;
                    xn = xrow[(unsigned) slot * (unsigned) a.M2 + (unsigned) b4c];
                }
                if (TT == 1) xw[0] = xn;

#pragma unroll
                for (int t = 0; t < TT; t++)
#pragma unroll
                    for (int j = 0; j < OT; j++)
                    {
                        bool ok = true;
                        if (CHECK) ok = p <= lim[j] + t;
                        if (ok)
                        {
                            cmac2(acc[t][j], xw[t], hval[j]);
                            if (owns_bin0) ny[t][j] += xw[t].y * hval[j].y;
                        }
                    }

                if (TT > 1)
                {
#pragma unroll
                    for (int t = TT - 1; t > 0; t--) xw[t] = xw[t - 1];
                    xw[0] = xn;
                }
            }
        }
    }

    if (owns_bin0)
    {
#pragma unroll
        for (int t = 0; t < TT; t++)
#pragma unroll
            for (int j = 0; j < OT; j++)
            {
                acc[t][j].x += ny[t][j];                        // sum(x.x*h.x - x.y*h.y) + sum(x.y*h.y) = DC products
                acc[t][j].y = ny[t][j];                         // Nyquist products
            }
    }

    if constexpr (INWG)
    {
        extern __shared__ __attribute__((aligned(16))) float4 red[];                 // [k-slice][t][j][lane]
        const int W = blockDim.x;
#pragma unroll
        for (int t = 0; t < TT; t++)
#pragma unroll
            for (int j = 0; j < OT; j++) red[((threadIdx.y * TT + t) * OT + j) * W + threadIdx.x] = acc[t][j];
        __syncthreads();
        if (threadIdx.y == 0 && tile_live && binlive)
        {
#pragma unroll
            for (int t = 0; t < TT; t++)
                if (t < live_t)
                {
#pragma unroll
                    for (int j = 0; j < OT; j++)
                        if (o0 + j < a.nout)
                        {
                            float4 sum = acc[t][j];
                            for (int k = 1; k < (int) blockDim.y; k++)
                            {
                                const float4 p = red[((k * TT + t) * OT + j) * W + threadIdx.x];
                                sum.x += p.x; sum.y += p.y; sum.z += p.z; sum.w += p.w;
                            }
                            a.Y[((long long) (t0 + t) * a.nout + (o0 + j)) * a.M2 + b4] = sum;
                        }
                }
        }
        return;
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
                    if (o0 + j < a.nout)
                    {
                        float4 *d = y + ((long long) (t0 + t) * a.nout + (o0 + j)) * a.M2;
                        *d = acc[t][j];
                    }
            }
    }
}

// ------------------------------------------------------------------------------------------------ launch plan

void mac_plan(const MacShape &s, MacPlan &pl)
{
    pl.mfma = 0;
    if (mac_mfma_applies(s))
    {
        mac_mfma_plan(s, pl);               // offline calls: the matrix cores (hcv_mac_mfma.hip)
        return;
    }
    constexpr int target_blocks = 768;      // workgroups to aim for (3 per CU)
    // hop tiling pays through H reuse on long reductions; the short head stages run leaner (fewer registers, so they
    // co-reside with the tail's workgroups)
    const int tt_cap = s.P >= 32 ? 8 : 4;
    const int M2 = s.M / 2;

    int tt = 1;
    while (tt * 2 <= s.T && tt * 2 <= tt_cap) tt *= 2;
    pl.tt = tt;

    int ot;
    if (s.diag)
        ot = 1;
    else if (tt == 1)
        ot = s.nout >= 8 ? 8 : s.nout >= 4 ? 4 : s.nout >= 2 ? 2 : 1;
    else
        ot = s.nout >= 3 ? 4 : 1;                              // TT > 1 kernels exist for OT in {1, 4}
    while (s.ot_cap > 0 && ot > s.ot_cap && ot > 1) ot >>= 1;
    if (tt > 1 && ot == 2) ot = 1;
    pl.ot = ot;

    pl.bx = M2 < 256 ? M2 : 256;
    const int tiles = (s.T + tt - 1) / tt;
    pl.by = std::max(1, std::min(256 / pl.bx, tiles));
    pl.tz = (tiles + pl.by - 1) / pl.by;
    pl.binblocks = (M2 + pl.bx - 1) / pl.bx;
    pl.outtiles = (s.nout + pl.ot - 1) / pl.ot;

    const long long K = (long long) (s.diag ? 1 : s.nin) * s.P;
    const long long base = (long long) pl.binblocks * pl.outtiles * pl.tz;
    // workgroups to aim for = what is resident at once: 3 per CU for the single-hop tile (132 registers), 2 per CU for the
    // large hop tiles (4 x 4 and 4 x 8 need 176 - 233): a third, half-empty round of workgroups cost the 4 x 8 tile 8 %
    const long long tgt = s.target_blocks > 0 ? s.target_blocks : (pl.ot == 4 && pl.tt >= 4) ? 512 : target_blocks;
    long long want = std::max<long long>(1, tgt / base);
    long long maxsplit = K / 8;                                 // keep every k-slice at least 8 long (4 and 2, and at most 8 / 4 slices for the
                                                                // ladder's pivot stage so that the inverse folds the sum: c5 ladder 0.1348 -> 0.144 / 0.144 / 0.144 / 0.161 ms)
    if (maxsplit < 1) maxsplit = 1;
    if (want > maxsplit) want = maxsplit;
    if (want < 1) want = 1;
    if (s.max_ksplit > 0 && want > s.max_ksplit) want = s.max_ksplit;
    pl.kper = (int) std::max<long long>(1, (K + want - 1) / want);      // (K = 0: no live input, the launch writes zeros)
    pl.ksplit = (int) ((K + pl.kper - 1) / pl.kper);
    if (pl.ksplit < 1) pl.ksplit = 1;
    // small engines: one hop, one or two outputs, a reduction of 32 .. a few hundred terms whose
    // spectra stay in the caches — the k-slices become the waves of one workgroup.  (Hop-tiled launches of such engines — the
    // 1 x 1 / 4096-point workload: four hops per block, 235 partitions — measured SLOWER this way, 0.0185 -> 0.0227 ms per block:
    // sixteen workgroups of eight waves walking 30 partitions each; they keep the hop-tiled kernel and its reduction launch.)
    pl.inwg = 0;
    const double h_bytes = 8.0 * s.M * (double) K * s.nout;
    if (pl.tt == 1 && tiles == 1 && pl.ot <= 2 && pl.ksplit > 1 && K >= 32 && K <= 1024 && M2 % 64 == 0 &&
        h_bytes <= 32.0 * 1048576.0 && s.target_blocks <= 0)
    {
        int kw = 16;
        while (kw > 2 && K / kw < 6) kw >>= 1;                    // at least six terms per slice
        while (kw > 2 && 1024 * pl.ot * pl.tt * kw > 48 * 1024) kw >>= 1;      // the slices' sums within the default 48 KiB of dynamic LDS
        pl.inwg = kw;
        pl.bx = 64;
        pl.by = 1;
        pl.tz = 1;
        pl.binblocks = M2 / 64;
        pl.kper = (int) ((K + kw - 1) / kw);
        pl.ksplit = 1;
    }
    // every H element is read exactly once per launch when a single hop tile covers the call: stream it
    // ... and they outgrow the caches: spectra of at most 24 MB stay in the XCDs' L2s from block to block (workgroup
    // b of every launch lands on the same XCD), where nontemporal loads would push them out (8 -> 1 / 5 s: MAC 7.7 -> 6.6 us)
    const double spectra_mb = 8.0 * s.M * (double) (s.diag ? 1 : s.nin) * s.P * s.nout / 1048576.0;
    pl.nt = (tiles == 1 && spectra_mb > 24.0) ? 1 : 0;
}

template <int OT, int TT>
static hipError_t launch_mac_tile(const MacParams &a, const MacPlan &pl, bool check, hipStream_t st)
{
    dim3 grid(pl.binblocks * pl.ksplit * (a.pin >= 0 ? 8 : 1), pl.outtiles, pl.tz);
    dim3 block(pl.bx, pl.by);
    // steady-state launches of the hop-tiled shapes take the software-pipelined kernel (hcv_mac_tiled.hip, its own translation
    // unit: it is compiled without the SLP vectoriser)
    if constexpr (OT <= 2 && TT == 1)
    {
        if (pl.inwg > 0)
        {
            const dim3 g2(pl.binblocks * (a.pin >= 0 ? 8 : 1), pl.outtiles, 1), b2(64, pl.inwg);
            const size_t lds = sizeof(float4) * 64 * OT * TT * (size_t) pl.inwg;
            if (check)
                hipLaunchKernelGGL((spectral_mac_kernel<OT, TT, true, false, true>), g2, b2, lds, st, a);
            else if (pl.nt)
                hipLaunchKernelGGL((spectral_mac_kernel<OT, TT, false, true, true>), g2, b2, lds, st, a);
            else
                hipLaunchKernelGGL((spectral_mac_kernel<OT, TT, false, false, true>), g2, b2, lds, st, a);
            return hipGetLastError();
        }
    }
    if (TT > 1 && !check) return launch_mac_tiled(OT, TT, pl.nt != 0, grid, block, a, st);
    if (check)
        hipLaunchKernelGGL((spectral_mac_kernel<OT, TT, true, false>), grid, block, 0, st, a);
    else if (pl.nt)
        hipLaunchKernelGGL((spectral_mac_kernel<OT, TT, false, true>), grid, block, 0, st, a);
    else
        hipLaunchKernelGGL((spectral_mac_kernel<OT, TT, false, false>), grid, block, 0, st, a);
    return hipGetLastError();
}

hipError_t launch_spectral_mac(const MacShape &s, const MacPlan &pl, const float2 *X, const float2 *H, float2 *Y, const long long *hv,
                               long long h_first, bool check, hipStream_t st)
{
    if (s.T <= 0 || s.nout <= 0) return hipSuccess;
    ORD_ACCESS(st, X, std::max<long long>(0, h_first - (s.P - 1)), h_first + s.T, (long long) s.R, false, "input-spectrum ring slots (multiply-accumulate)");
    ORD_ACCESS(st, Y, 0, 1, 0, true, "partial spectra (multiply-accumulate)");
    MacParams a;
    a.X = reinterpret_cast<const float4 *>(X);
    a.H = reinterpret_cast<const float4 *>(H);
    a.Y = reinterpret_cast<float4 *>(Y);
    a.hv = hv;
    a.h_first = h_first;
    a.M2 = s.M / 2;
    a.R = s.R;
    a.P = s.P;
    a.Pcap = s.Pcap;
    a.T = s.T;
    a.nin = s.diag ? 1 : s.nin;
    a.nin_alloc = s.nin_alloc;
    a.nout = s.nout;
    a.diag = s.diag;
    a.ksplit = pl.ksplit;
    a.kper = pl.kper;
    a.binblocks = pl.binblocks;
    a.ks_stride4 = (long long) s.T * s.nout * (s.M / 2);
    a.hop_min = s.hop_min;
    if (pl.mfma)
    {
        if (check) return hipErrorInvalidValue;             // (planned for a steady launch only: MacShape::steady)
        a.pin = -1;
        return launch_mac_mfma(pl, a, st);
    }
    a.pin = (pl.outtiles == 1 && pl.tz == 1) ? xcd_pin_for((long long) pl.binblocks * pl.ksplit * (pl.inwg > 0 ? pl.inwg / 4 : 1)) : -1;
    const int key = pl.ot * 16 + pl.tt;
    switch (key)
    {
        case 8 * 16 + 1: return launch_mac_tile<8, 1>(a, pl, check, st);
        case 4 * 16 + 1: return launch_mac_tile<4, 1>(a, pl, check, st);
        case 2 * 16 + 1: return launch_mac_tile<2, 1>(a, pl, check, st);
        case 1 * 16 + 1: return launch_mac_tile<1, 1>(a, pl, check, st);
        case 4 * 16 + 2: return launch_mac_tile<4, 2>(a, pl, check, st);
        case 1 * 16 + 2: return launch_mac_tile<1, 2>(a, pl, check, st);
        case 4 * 16 + 4: return launch_mac_tile<4, 4>(a, pl, check, st);
        case 1 * 16 + 4: return launch_mac_tile<1, 4>(a, pl, check, st);
        case 4 * 16 + 8: return launch_mac_tile<4, 8>(a, pl, check, st);
        case 1 * 16 + 8: return launch_mac_tile<1, 8>(a, pl, check, st);
        default: return hipErrorInvalidValue;
    }
}

// (Engine::init, once per device: HIP loads a translation unit's code object at the first launch of one of its kernels — 0.3 - 0.8 ms on the
// calling thread, which for the kernels of a control section or a restart is the audio thread in mid-stream; asking for a kernel's attributes loads it now)
void preload_mac()
{
    hipFuncAttributes fa;
    (void) hipFuncGetAttributes(&fa, reinterpret_cast<const void *>((spectral_mac_kernel<8, 1, false, true, false>)));
    (void) hipGetLastError();
}

} // namespace hcv
