// Device-side pieces of the residue-split real transforms (see hcv_fft_split.hip for the mathematics), shared by the translation
// units that build fused blocks out of them (hcv_fft_split.hip: one-output engines; hcv_fused_nxm.hip: n x m matrices).
#pragma once

#include "hcv_fft_device.h"
#include "hcv_fused_sync.h"

#include <type_traits>

namespace hcv
{

namespace
{
    // exp(-2 pi i m / (2 S)) for m in [0, 2 S) from the S-entry table `tl` (in LDS: a pass's three twiddles cost three LDS reads
    // instead of three trips to the L2 — with five passes per transform those trips were most of the kernel)
    template <int LOG2S> __device__ __forceinline__ float2 lroot(const float2 *tl, int m)
    {
        constexpr int S = 1 << LOG2S;
        const float2 w = tl[m & (S - 1)];
        return (m & S) ? make_float2(-w.x, -w.y) : w;
    }

    // one radix-4 Stockham pass of the S-point transform on TG threads (LdsFFT::pass4 with the twiddles in LDS; LOG2P = log2 of
    // the pass's stride p, 0 for the first pass whose twiddles are all 1)
    template <int LOG2S, int TG, int LOG2P, class Src, class Dst>
    __device__ __forceinline__ void split_pass4(const Src &src, const Dst &dst, int tid, const float2 *tl)
    {
        constexpr int S = 1 << LOG2S, NB4 = S / 4, BPT = (NB4 + TG - 1) / TG, P = 1 << LOG2P;
        float2 u[BPT][4];
#pragma unroll
        for (int b = 0; b < BPT; b++)
        {
            const int i = tid + b * TG;
            if (NB4 % TG == 0 || i < NB4)
            {
#pragma unroll
                for (int r = 0; r < 4; r++) u[b][r] = src(i + r * NB4);
            }
        }
        if (Src::is_lds && Dst::is_lds) __syncthreads();
#pragma unroll
        for (int b = 0; b < BPT; b++)
        {
            const int i = tid + b * TG;
            if (NB4 % TG == 0 || i < NB4)
            {
                const int k = i & (P - 1);
                const int j = ((i - k) << 2) + k;
                if (LOG2P > 0)
                {
                    const int step = k * ((2 * S) / (4 * P));            // exp(-2 pi i k r / (4 P)) = root(r * step)
                    u[b][1] = cmul(u[b][1], lroot<LOG2S>(tl, step));
                    u[b][2] = cmul(u[b][2], lroot<LOG2S>(tl, 2 * step));
                    u[b][3] = cmul(u[b][3], lroot<LOG2S>(tl, 3 * step));
                }
                radix4(u[b][0], u[b][1], u[b][2], u[b][3]);
#pragma unroll
                for (int r = 0; r < 4; r++) dst(j + r * P, u[b][r]);
            }
        }
        if (Dst::is_lds) __syncthreads();
    }

    template <int LOG2S, int TG, class Src, class Dst>
    __device__ __forceinline__ void split_pass2(const Src &src, const Dst &dst, int tid, const float2 *tl)
    {
        constexpr int S = 1 << LOG2S, NB2 = S / 2, BPT = (NB2 + TG - 1) / TG, P = S / 2;       // (always the last pass)
        float2 u[BPT][2];
#pragma unroll
        for (int b = 0; b < BPT; b++)
        {
            const int i = tid + b * TG;
            if (NB2 % TG == 0 || i < NB2)
            {
                u[b][0] = src(i);
                u[b][1] = src(i + NB2);
            }
        }
#pragma unroll
        for (int b = 0; b < BPT; b++)
        {
            const int i = tid + b * TG;
            if (NB2 % TG == 0 || i < NB2)
            {
                const float2 u0 = u[b][0], u1 = cmul(u[b][1], lroot<LOG2S>(tl, 2 * i));     // exp(-2 pi i k / (2 P)), k = i: entry k (2 S) / (2 P) = 2 k
                dst(i, make_float2(u0.x + u1.x, u0.y + u1.y));
                dst(i + P, make_float2(u0.x - u1.x, u0.y - u1.y));
            }
        }
    }

    template <int LOG2S, int TG, int LOG2P, class St> struct SubFFT
    {
        __device__ static __forceinline__ void run(LdsBuf<float2> s, const St &st, int tid, const float2 *tl)
        {
            const LdsIO<float2> io = { s };
            if constexpr (LOG2P + 2 == LOG2S)
                split_pass4<LOG2S, TG, LOG2P>(io, st, tid, tl);                  // the last pass delivers
            else if constexpr (LOG2P + 1 == LOG2S)
                split_pass2<LOG2S, TG>(io, st, tid, tl);
            else
            {
                split_pass4<LOG2S, TG, LOG2P>(io, io, tid, tl);
                SubFFT<LOG2S, TG, LOG2P + 2, St>::run(s, st, tid, tl);
            }
        }
    };

    // S-point forward transform of the sequence in `s` (synchronised), natural order, results handed to `st`; `tl` = the
    // (2 S)-th roots in LDS (synchronised)
    template <int LOG2S, int TG, class St>
    __device__ __forceinline__ void sub_fft(LdsBuf<float2> s, const St &st, int tid, const float2 *tl)
    {
        SubFFT<LOG2S, TG, 0, St>::run(s, st, tid, tl);
    }

    // the S-entry table of the sub-transform, global -> LDS (16-byte moves; the caller synchronises)
    template <int LOG2S, int TG> __device__ __forceinline__ void stage_table(float2 *tl, const float2 *__restrict__ tws, int tid)
    {
        constexpr int V = (1 << LOG2S) / 2;                 // float4 count
        const float4 *src = reinterpret_cast<const float4 *>(tws);
        float4 *dst = reinterpret_cast<float4 *>(tl);
        for (int e = tid; e < V; e += TG) dst[e] = src[e];
    }

    // forward: bin R k + r (and its mirror) of the packed half spectrum
    template <int LOG2S, int LOG2R, bool AGENT = false> struct SplitSpectrumStore
    {
        static constexpr bool is_lds = false;
        float2 *dst;
        int r;
        __device__ __forceinline__ void operator()(int k, float2 v) const
        {
            constexpr int S = 1 << LOG2S, R = 1 << LOG2R, HALF = S / 2;
            v.x += v.x;
            v.y += v.y;
            if (r == 0)
            {
                if (k == 0) put1<AGENT>(&dst[0].x, v.x);             // 2 X[0]
                else if (k == HALF) put1<AGENT>(&dst[0].y, v.x);     // 2 X[N/2]
                else if (k < HALF) put2<AGENT>(dst + R * k, v);
            }
            else if (k < HALF)
                put2<AGENT>(dst + R * k + r, v);
            else if (r != R / 2)
                put2<AGENT>(dst + R * (S - 1 - k) + (R - r), make_float2(v.x, -v.y));
        }
    };

    // inverse: samples R n1 + 2 j, R n1 + 2 j + 1 of the frame (second half only); v = (x_b, x_a): the transform ran on exchanged re / im
    template <int LOG2S, int LOG2R> struct SplitSampleStore
    {
        static constexpr bool is_lds = false;
        float *row;             // sample e of the frame lands at row[e]
        float scale;
        int j;
        __device__ __forceinline__ void operator()(int n1, float2 v) const
        {
            constexpr int S = 1 << LOG2S, R = 1 << LOG2R;
            if (n1 < S / 2) return;
            *reinterpret_cast<float2 *>(row + R * n1 + 2 * j) = make_float2(v.y * scale, v.x * scale);
        }
    };
}

// Residue r of one frame's spectrum, by the 256 threads of a workgroup.  lds / tl / wr = the workgroup's LDS (lds_padded(S), S and R
// float2).  DIRECT: the new hop of the frame comes from the caller's block (positions >= n0) and the workgroup of residue 0 files
// it in the history ring, as rfft_frames_direct_kernel does.  AGENT: see put2.
template <int LOG2N, int LOG2R, bool DIRECT, bool AGENT, int TG = 256>
__device__ __forceinline__ void rfft_split_body(float2 *lds, float2 *tl, float2 *wr, int tid, int r, float *hrow, const float *irow, long long base, long long n0,
                                                long long hist_mask, float2 *dstX, const float2 *__restrict__ tw, const float2 *__restrict__ tws)
{
    constexpr int R = 1 << LOG2R, LOG2S = LOG2N - LOG2R, S = 1 << LOG2S;
    const LdsBuf<float2> s = { lds };
    stage_table<LOG2S, TG>(tl, tws, tid);
    if (tid < R)
    {
        float sn, cs;
        sincospif(-2.0f * (float) tid / (float) R, &sn, &cs);        // W_R^tid
        wr[tid] = make_float2(cs, sn);
    }
    __syncthreads();

    for (int v = tid; v < S / 4; v += TG)
    {
        float4 f[R];
#pragma unroll
        for (int qq = 0; qq < R; qq++)
        {
            const int e = 4 * v + S * qq;
            const long long pos = base + e;
            const float *src = (DIRECT && pos >= n0) ? irow + (pos - n0) : hrow + (pos & hist_mask);
            f[qq] = *reinterpret_cast<const float4 *>(src);
        }
        if (DIRECT && r == 0)
        {
#pragma unroll
            for (int qq = R / 2; qq < R; qq++)
            {
                // (AGENT: the ring's new hop is read by launches that may have started before this one ends — written through, like the spectrum)
                float2 *hp = reinterpret_cast<float2 *>(hrow + ((base + 4 * v + S * qq) & hist_mask));
                put2<AGENT>(hp, make_float2(f[qq].x, f[qq].y));
                put2<AGENT>(hp + 1, make_float2(f[qq].z, f[qq].w));
            }
        }
        float2 a[4] = { make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f) };
#pragma unroll
        for (int qq = 0; qq < R; qq++)
        {
            const float2 c = wr[(qq * r) & (R - 1)];
            a[0].x += f[qq].x * c.x; a[0].y += f[qq].x * c.y;
            a[1].x += f[qq].y * c.x; a[1].y += f[qq].y * c.y;
            a[2].x += f[qq].z * c.x; a[2].y += f[qq].z * c.y;
            a[3].x += f[qq].w * c.x; a[3].y += f[qq].w * c.y;
        }
#pragma unroll
        for (int jj = 0; jj < 4; jj++)
        {
            const int n = 4 * v + jj;
            s[n] = cmul(a[jj], root<LOG2N - 1>(tw, n * r));         // n r < S (R / 2 + 1) <= M + S: root() covers [0, 2M)
        }
    }
    __syncthreads();
    const SplitSpectrumStore<LOG2S, LOG2R, AGENT> st = { dstX, r };
    sub_fft<LOG2S, TG>(s, st, tid, tl);
}

// Sample classes n2 = 2 j, 2 j + 1 of one frame, by the 256 threads of a workgroup.  dyn = the workgroup's LDS: M (spectrum) +
// lds_padded(S) + S + R float2.  Ysrc: the spectrum, or the first of `ksplit` partial sums `ks_stride` float2 apart (added up while
// staged).  AGENT: the spectrum was written by other workgroups of this launch (see put2).  `row`: sample e of the frame lands at row[e].
template <int LOG2N, int LOG2R, bool AGENT, int TG = 256>
__device__ __forceinline__ void rifft_split_body(float2 *dyn, int tid, int j, const float2 *__restrict__ Ysrc, int ksplit, long long ks_stride, float *row,
                                                 const float2 *__restrict__ tw, const float2 *__restrict__ tws)
{
    constexpr int N = 1 << LOG2N, M = N / 2, R = 1 << LOG2R, LOG2S = LOG2N - LOG2R, S = 1 << LOG2S;
    float2 *spec = dyn;                                  // [M] the packed half spectrum
    const LdsBuf<float2> s = { dyn + M };                // [lds_padded(S)] the class pair's transform
    float2 *tl = dyn + M + lds_padded(S);                // [S] the sub-transform's twiddles
    float2 *wr = tl + S;                                 // [R] W_R^j
    if (tid < R)
    {
        float sn, cs;
        sincospif(-2.0f * (float) tid / (float) R, &sn, &cs);
        wr[tid] = make_float2(cs, sn);
    }
    stage_table<LOG2S, TG>(tl, tws, tid);
    if constexpr (AGENT)
    {
        // float2 per thread; the k-slices of this launch's multiply-accumulate (`ks_stride` float2 apart, a power of two of them) are added up
        // in slice order.  EVERY slice's load of a chunk of elements is issued before the first add: an agent-scope load goes to the memory
        // side (2 - 3 us), and a chain of slice after slice was most of the n x m block's inverse phase
        constexpr int V = M / TG;
        auto stage = [&](auto ks_c)
        {
            constexpr int KS = decltype(ks_c)::value, E = (32 / KS) < V ? (32 / KS) : V;
            static_assert(V % E == 0, "");
#pragma unroll
            for (int e0 = 0; e0 < V; e0 += E)
            {
                float2 b[KS][E];
#pragma unroll
                for (int ks = 0; ks < KS; ks++)
#pragma unroll
                    for (int e = 0; e < E; e++) b[ks][e] = get2<true>(Ysrc + (long long) ks * ks_stride + tid + (e0 + e) * TG);
#pragma unroll
                for (int e = 0; e < E; e++)
                {
                    float2 acc = b[0][e];
#pragma unroll
                    for (int ks = 1; ks < KS; ks++)
                    {
                        acc.x += b[ks][e].x;
                        acc.y += b[ks][e].y;
                    }
                    spec[tid + (e0 + e) * TG] = acc;
                }
            }
        };
        if (ksplit == 1) stage(std::integral_constant<int, 1>{});
        else if (ksplit == 2) stage(std::integral_constant<int, 2>{});
        else if (ksplit == 4) stage(std::integral_constant<int, 4>{});
        else stage(std::integral_constant<int, 8>{});
    }
    else
    {
        // stage (and add up) the spectrum: 16-byte loads.  A power of two of slices: EVERY slice's loads of a chunk of elements are in flight
        // before the first add (sixteen loads per thread at a time) — slice after slice, each waiting for the one before, was 21 us of
        // the n x m block's inverse with four slices against 8 with one; other counts take the slices in turn
        const float4 *src = reinterpret_cast<const float4 *>(Ysrc);
        float4 *dst4 = reinterpret_cast<float4 *>(spec);
        constexpr int V = M / 2 / TG;                    // float4 per thread
        auto stage = [&](auto ks_c)
        {
            constexpr int KS = decltype(ks_c)::value, E0 = 16 / KS, E = E0 < 1 ? 1 : (E0 < V ? E0 : V);
            static_assert(V % E == 0, "");
#pragma unroll
            for (int e0 = 0; e0 < V; e0 += E)
            {
                float4 b[KS][E];
#pragma unroll
                for (int ks = 0; ks < KS; ks++)
#pragma unroll
                    for (int e = 0; e < E; e++) b[ks][e] = src[ks * (ks_stride / 2) + tid + (e0 + e) * TG];
#pragma unroll
                for (int e = 0; e < E; e++)
                {
                    float4 acc = b[0][e];
#pragma unroll
                    for (int ks = 1; ks < KS; ks++)
                    {
                        acc.x += b[ks][e].x; acc.y += b[ks][e].y; acc.z += b[ks][e].z; acc.w += b[ks][e].w;
                    }
                    dst4[tid + (e0 + e) * TG] = acc;
                }
            }
        };
        if (ksplit == 1) stage(std::integral_constant<int, 1>{});
        else if (ksplit == 2) stage(std::integral_constant<int, 2>{});
        else if (ksplit == 4) stage(std::integral_constant<int, 4>{});
        else if (ksplit == 8) stage(std::integral_constant<int, 8>{});
        else
        {
            float4 acc[V];
#pragma unroll
            for (int e = 0; e < V; e++) acc[e] = src[tid + e * TG];
            for (int ks = 1; ks < ksplit; ks++)
            {
                float4 b[V];
#pragma unroll
                for (int e = 0; e < V; e++) b[e] = src[ks * (ks_stride / 2) + tid + e * TG];
#pragma unroll
                for (int e = 0; e < V; e++)
                {
                    acc[e].x += b[e].x; acc[e].y += b[e].y; acc[e].z += b[e].z; acc[e].w += b[e].w;
                }
            }
#pragma unroll
            for (int e = 0; e < V; e++) dst4[tid + e * TG] = acc[e];
        }
    }
    __syncthreads();

    const int na = 2 * j, nb = 2 * j + 1;
    for (int b1 = tid; b1 < S; b1 += TG)
    {
        float2 ca = make_float2(0.f, 0.f), cb = make_float2(0.f, 0.f);
#pragma unroll
        for (int b2 = 0; b2 < R; b2++)
        {
            float2 y;
            if (b2 < R / 2)
            {
                y = spec[b1 + S * b2];
                if (b2 == 0 && b1 == 0) y.y = 0.f;                   // Yfull[0] = DC (real); .y of bin 0 is the Nyquist value
            }
            else if (b1 == 0)
            {
                if (b2 == R / 2) y = make_float2(spec[0].y, 0.f);   // Yfull[N/2]
                else
                {
                    y = spec[S * (R - b2)];
                    y.y = -y.y;
                }
            }
            else
            {
                y = spec[S * (R - b2) - b1];
                y.y = -y.y;
            }
            // V_R^(n b2) = conj W_R^(n b2)
            const float2 va = wr[(na * b2) & (R - 1)], vb = wr[(nb * b2) & (R - 1)];
            ca.x += y.x * va.x + y.y * va.y; ca.y += y.y * va.x - y.x * va.y;
            cb.x += y.x * vb.x + y.y * vb.y; cb.y += y.y * vb.x - y.x * vb.y;
        }
        // g = V_N^(n2 b1) c = conj(W_N^(n2 b1)) c
        const float2 wa = root<LOG2N - 1>(tw, na * b1), wb = root<LOG2N - 1>(tw, nb * b1);
        const float2 ga = make_float2(ca.x * wa.x + ca.y * wa.y, ca.y * wa.x - ca.x * wa.y);
        const float2 gb = make_float2(cb.x * wb.x + cb.y * wb.y, cb.y * wb.x - cb.x * wb.y);
        // u = ga + i gb, handed to the forward transform with re / im exchanged (the inverse by the swap trick)
        s[b1] = make_float2(ga.y + gb.x, ga.x - gb.y);
    }
    __syncthreads();
    const SplitSampleStore<LOG2S, LOG2R> st = { row, 1.f / (float) (8 * M), j };
    sub_fft<LOG2S, TG>(s, st, tid, tl);
}

} // namespace hcv
