// Device-side FFT building blocks shared by the LDS-resident transforms (hcv_kernels.hip) and the
// four-step transforms for N > 32768 (hcv_bigfft.hip).  gfx950 / wave64 only.
#pragma once

#include <hip/hip_runtime.h>

namespace hcv
{

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// 16-byte streaming load that bypasses cache retention (global_load_dwordx4 ... nt)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_nt(const float4 *p)
{
    v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ double2 cmul(double2 a, double2 b)
{
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// tw holds the N-th roots of unity exp(-2*pi*i*m/N) for m in [0, N/2); the second half of the circle
// is the negated first half.  C = float2 (the convolution path) or double2 (the double FFT surface).
template <int LOG2M, class C>
__device__ __forceinline__ C root(const C *__restrict__ tw, int m)
{
    constexpr int M = 1 << LOG2M;                    // N/2 table entries
    C w = tw[m & (M - 1)];
    return (m & M) ? C(-w.x, -w.y) : w;
}

// ------------------------------------------------------------------------------------------------
// In-LDS complex FFT of M = 2^LOG2M points, forward sign, unnormalised, natural order in and out.
// Stockham autosort, radix-4 passes plus one radix-2 pass when LOG2M is odd.  TG threads cooperate on one
// transform; every pass is "read my butterflies into registers / barrier / write results / barrier", so a
// single M-point LDS buffer suffices (64 KiB at N = 16384 in float).
// ------------------------------------------------------------------------------------------------

template <int LOG2M, int TG, class C = float2>
struct LdsFFT
{
    static constexpr int M = 1 << LOG2M;
    static constexpr int NB4 = M / 4;
    static constexpr int BPT4 = (NB4 + TG - 1) / TG;
    static constexpr int NB2 = M / 2;
    static constexpr int BPT2 = (NB2 + TG - 1) / TG;

    __device__ static __forceinline__ void run(C *s, int tid, const C *__restrict__ tw)
    {
        int p = 1;
#pragma unroll 1
        for (int pass = 0; pass < LOG2M / 2; pass++, p <<= 2)
        {
            C u[BPT4][4];
#pragma unroll
            for (int b = 0; b < BPT4; b++)
            {
                int i = tid + b * TG;
                if (NB4 % TG == 0 || i < NB4)
                {
#pragma unroll
                    for (int r = 0; r < 4; r++) u[b][r] = s[i + r * NB4];
                }
            }
            __syncthreads();
#pragma unroll
            for (int b = 0; b < BPT4; b++)
            {
                int i = tid + b * TG;
                if (NB4 % TG == 0 || i < NB4)
                {
                    int k = i & (p - 1);
                    int j = ((i - k) << 2) + k;
                    // twiddle exp(-2 pi i k r / (4p)) = root(k * r * (2M / 4p))
                    int step = k * ((2 * M) / (4 * p));
                    C u0 = u[b][0];
                    C u1 = cmul(u[b][1], root<LOG2M>(tw, step));
                    C u2 = cmul(u[b][2], root<LOG2M>(tw, 2 * step));
                    C u3 = cmul(u[b][3], root<LOG2M>(tw, 3 * step));
                    C a = C(u0.x + u2.x, u0.y + u2.y);
                    C c = C(u0.x - u2.x, u0.y - u2.y);
                    C e = C(u1.x + u3.x, u1.y + u3.y);
                    C d = C(u1.y - u3.y, u3.x - u1.x);     // -i * (u1 - u3)
                    s[j] = C(a.x + e.x, a.y + e.y);
                    s[j + p] = C(c.x + d.x, c.y + d.y);
                    s[j + 2 * p] = C(a.x - e.x, a.y - e.y);
                    s[j + 3 * p] = C(c.x - d.x, c.y - d.y);
                }
            }
            __syncthreads();
        }
        if (LOG2M & 1)
        {
            C u[BPT2][2];
#pragma unroll
            for (int b = 0; b < BPT2; b++)
            {
                int i = tid + b * TG;
                if (NB2 % TG == 0 || i < NB2)
                {
                    u[b][0] = s[i];
                    u[b][1] = s[i + NB2];
                }
            }
            __syncthreads();
#pragma unroll
            for (int b = 0; b < BPT2; b++)
            {
                int i = tid + b * TG;
                if (NB2 % TG == 0 || i < NB2)
                {
                    int k = i & (p - 1);
                    int j = ((i - k) << 1) + k;
                    C u0 = u[b][0];
                    C u1 = cmul(u[b][1], root<LOG2M>(tw, k * ((2 * M) / (2 * p))));
                    s[j] = C(u0.x + u1.x, u0.y + u1.y);
                    s[j + p] = C(u0.x - u1.x, u0.y - u1.y);
                }
            }
            __syncthreads();
        }
    }
};

// threads cooperating on one transform and transforms per 256-thread workgroup
template <int LOG2M> struct FFTGeom
{
    static constexpr int M = 1 << LOG2M;
    static constexpr int TG = (M / 4) < 256 ? (M / 4) : 256;
    static constexpr int G = 256 / TG;
    static constexpr int THREADS = TG * G;
};


} // namespace hcv
