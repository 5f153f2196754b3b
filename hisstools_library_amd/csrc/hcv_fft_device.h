// Device-side FFT building blocks shared by the LDS-resident transforms (hcv_kernels.hip) and the
// four-step transforms for N > 32768 (hcv_bigfft.hip).  gfx950 / wave64 only.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

namespace hcv
{

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------

// Complex float arithmetic is written on 2-vectors (re, im): the adds are then one v_pk_add_f32 each and the products two
// packed instructions with the broadcast / swap in their operand selects.  Left as scalar component arithmetic the SLP
// vectoriser pairs components of DIFFERENT values and pays for it in register moves (a third of the vector instructions of
// the LDS transform kernels).  HCV_FFT_PACKED=0 keeps the component forms.
#ifndef HCV_FFT_PACKED
#define HCV_FFT_PACKED 1
#endif
#ifndef HCV_FFT_LDS_BASE
#define HCV_FFT_LDS_BASE 1      // radix-16 / radix-32 passes address LDS as one padded base + constant multiples (LdsFFT::pass16)
#endif
#ifndef HCV_FFT_RADIX8
#define HCV_FFT_RADIX8 1        // log2 sizes 4n + 3, float, last pass stored directly: one radix-8 tail pass instead of radix-4 +
                                // radix-2 (LdsFFT::run).  Complex 2^11 +4 %, real inverse 2^12 +6-8 %; NOT where the last pass goes
                                // back to LDS (the real forward transform: -4 %) and not in double (156 registers instead of 130:
                                // real forward 2^12 -12 %): profiles/r02_fft_radix8_ab.txt
#endif
#ifndef HCV_FFT_RADIX32
#define HCV_FFT_RADIX32 1       // odd log2 sizes: a radix-32 first pass instead of a radix-2 tail (LdsFFT::run)
#endif
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f cmulv(v2f a, v2f b) { return a.xx * b + a.yy * v2f{ -b.y, b.x }; }

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
#if HCV_FFT_PACKED
    const v2f r = cmulv(v2f{ a.x, a.y }, v2f{ b.x, b.y });
    return make_float2(r.x, r.y);
#else
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
#endif
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
// Stockham autosort: radix-16 passes (two radix-4 layers fused in registers, so the data crosses LDS once per FOUR
// butterfly levels), then one radix-4 and / or radix-2 pass for the remaining levels.  TG threads cooperate on one
// transform; every pass is "read my butterflies into registers / barrier / write results / barrier", so a single
// M-point LDS buffer (LdsBuf, padded) suffices (68 KiB at N = 16384 in float).  Twiddles come from the table: a radix-16 butterfly
// needs six loads (w, w^2, w^3 and w^4, w^8, w^12) and none in the first pass, where every twiddle is 1.
// ------------------------------------------------------------------------------------------------

// LDS buffer of one transform with one spare element after every 16: the radix-16 passes touch LDS at element strides
// of 16 * p, which would land 64 lanes on two banks; the skew spreads them over all of them (<= 2-way conflicts).
__host__ __device__ constexpr int lds_padded(int points) { return points + (points >> 4); }

// Pitch (in elements) between the columns / rows of a four-step tile in LDS.  lds_padded(P) alone is a multiple of 16 elements:
// the transposing phases, where adjacent lanes sit on adjacent tile lines (2 or 4 lines apart with vector I/O), would then put
// 8 - 16 lanes on the same bank.  One more element makes the line stride odd in 8-byte units, and those phases conflict-free.
__host__ __device__ constexpr int fourstep_pitch(int points) { return lds_padded(points) + 1; }

template <class C> struct LdsBuf
{
    C *base;
    __device__ __forceinline__ C &operator[](int i) const { return base[i + (i >> 4)]; }
};

template <class C> __device__ __forceinline__ void radix4(C &x0, C &x1, C &x2, C &x3)
{
    const C a = C(x0.x + x2.x, x0.y + x2.y);
    const C c = C(x0.x - x2.x, x0.y - x2.y);
    const C e = C(x1.x + x3.x, x1.y + x3.y);
    const C d = C(x1.y - x3.y, x3.x - x1.x);                 // -i * (x1 - x3)
    x0 = C(a.x + e.x, a.y + e.y);
    x1 = C(c.x + d.x, c.y + d.y);
    x2 = C(a.x - e.x, a.y - e.y);
    x3 = C(c.x - d.x, c.y - d.y);
}

// v * (cr + i ci) for a compile-time-known constant
template <class C, class R> __device__ __forceinline__ C mulk(C v, R cr, R ci)
{
    return C(v.x * cr - v.y * ci, v.x * ci + v.y * cr);
}

// 16-point DFT of u[0..15] in place: on return u[q] holds bin q.  Input index r = 4a + b, bin q = q1 + 4 q2:
// radix-4 over a, twiddle W16^(b q1), radix-4 over b.  wb (b = 1..3) are extra per-column factors folded into the
// middle twiddle (the Stockham pass twiddle w^b); pass nullptr-like ones via ONES = true to skip them.
template <bool ONES, class C> __device__ __forceinline__ void dft16(C *u, C w1, C w2, C w3)
{
    typedef decltype(u[0].x) R;
    const R c1 = (R) 0.92387953251128675613, s1 = (R) 0.38268343236508977173, h = (R) 0.70710678118654752440;
#pragma unroll
    for (int b = 0; b < 4; b++) radix4(u[b], u[4 + b], u[8 + b], u[12 + b]);          // u[4 q1 + b] = T_b[q1]
    if (!ONES)
    {
#pragma unroll
        for (int q1 = 0; q1 < 4; q1++)
        {
            u[4 * q1 + 1] = cmul(u[4 * q1 + 1], w1);
            u[4 * q1 + 2] = cmul(u[4 * q1 + 2], w2);
            u[4 * q1 + 3] = cmul(u[4 * q1 + 3], w3);
        }
    }
    // W16^(b q1), W16 = exp(-2 pi i / 16)
    u[4 + 1] = mulk(u[4 + 1], c1, -s1);                       // b q1 = 1
    u[4 + 2] = mulk(u[4 + 2], h, -h);                         // 2
    u[4 + 3] = mulk(u[4 + 3], s1, -c1);                       // 3
    u[8 + 1] = mulk(u[8 + 1], h, -h);                         // 2
    u[8 + 2] = C(u[8 + 2].y, -u[8 + 2].x);                    // 4: -i
    u[8 + 3] = mulk(u[8 + 3], -h, -h);                        // 6
    u[12 + 1] = mulk(u[12 + 1], s1, -c1);                     // 3
    u[12 + 2] = mulk(u[12 + 2], -h, -h);                      // 6
    u[12 + 3] = mulk(u[12 + 3], -c1, s1);                     // 9
#pragma unroll
    for (int q1 = 0; q1 < 4; q1++) radix4(u[4 * q1], u[4 * q1 + 1], u[4 * q1 + 2], u[4 * q1 + 3]);   // u[4 q1 + q2] = X[q1 + 4 q2]
}

// The radix-16 and radix-4 butterflies for float on 2-vectors (see cmul above).
#if HCV_FFT_PACKED
__device__ __forceinline__ v2f mulkv(v2f v, float cr, float ci) { return v.xx * v2f{ cr, ci } + v.yy * v2f{ -ci, cr }; }
__device__ __forceinline__ void radix4v(v2f &x0, v2f &x1, v2f &x2, v2f &x3)
{
    const v2f a = x0 + x2, c = x0 - x2, e = x1 + x3, f = x1 - x3;
    const v2f d = v2f{ f.y, -f.x };                          // -i * (x1 - x3)
    x0 = a + e;
    x1 = c + d;
    x2 = a - e;
    x3 = c - d;
}
__device__ __forceinline__ void radix4(float2 &x0, float2 &x1, float2 &x2, float2 &x3)
{
    v2f a = v2f{ x0.x, x0.y }, b = v2f{ x1.x, x1.y }, c = v2f{ x2.x, x2.y }, d = v2f{ x3.x, x3.y };
    radix4v(a, b, c, d);
    x0 = make_float2(a.x, a.y);
    x1 = make_float2(b.x, b.y);
    x2 = make_float2(c.x, c.y);
    x3 = make_float2(d.x, d.y);
}
__device__ __forceinline__ float2 mulk(float2 v, float cr, float ci)
{
    const v2f r = mulkv(v2f{ v.x, v.y }, cr, ci);
    return make_float2(r.x, r.y);
}
template <bool ONES> __device__ __forceinline__ void dft16(float2 *u, float2 w1, float2 w2, float2 w3)
{
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    v2f v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = v2f{ u[i].x, u[i].y };
#pragma unroll
    for (int b = 0; b < 4; b++) radix4v(v[b], v[4 + b], v[8 + b], v[12 + b]);
    if (!ONES)
    {
        const v2f t1 = v2f{ w1.x, w1.y }, t2 = v2f{ w2.x, w2.y }, t3 = v2f{ w3.x, w3.y };
#pragma unroll
        for (int q1 = 0; q1 < 4; q1++)
        {
            v[4 * q1 + 1] = cmulv(v[4 * q1 + 1], t1);
            v[4 * q1 + 2] = cmulv(v[4 * q1 + 2], t2);
            v[4 * q1 + 3] = cmulv(v[4 * q1 + 3], t3);
        }
    }
    v[4 + 1] = mulkv(v[4 + 1], c1, -s1);
    v[4 + 2] = mulkv(v[4 + 2], h, -h);
    v[4 + 3] = mulkv(v[4 + 3], s1, -c1);
    v[8 + 1] = mulkv(v[8 + 1], h, -h);
    v[8 + 2] = v2f{ v[8 + 2].y, -v[8 + 2].x };
    v[8 + 3] = mulkv(v[8 + 3], -h, -h);
    v[12 + 1] = mulkv(v[12 + 1], s1, -c1);
    v[12 + 2] = mulkv(v[12 + 2], -h, -h);
    v[12 + 3] = mulkv(v[12 + 3], -c1, s1);
#pragma unroll
    for (int q1 = 0; q1 < 4; q1++) radix4v(v[4 * q1], v[4 * q1 + 1], v[4 * q1 + 2], v[4 * q1 + 3]);
#pragma unroll
    for (int i = 0; i < 16; i++) u[i] = make_float2(v[i].x, v[i].y);
}
#endif

// Where a pass takes its inputs from / puts its outputs: the LDS buffer itself ...
template <class C> struct LdsIO
{
    static constexpr bool is_lds = true;
    LdsBuf<C> s;
    __device__ __forceinline__ C operator()(int n) const { return s[n]; }
    __device__ __forceinline__ void operator()(int k, C v) const { s[k] = v; }
};
// ... or any functor with `static constexpr bool is_lds = false`, `C operator()(int n)` (element n of the input) or
// `void operator()(int k, C v)` (bin k of the result): the first pass then reads HBM straight into its butterfly
// registers (all of a thread's loads in flight at once) and the last one writes its results without touching LDS.

template <int LOG2M, int TG, class C = float2>
struct LdsFFT
{
    static constexpr int M = 1 << LOG2M;
    static constexpr int N16 = LOG2M / 4;                    // radix-16 passes
    static constexpr int REM = LOG2M % 4;                    // 2, 3: one radix-4 pass; odd: one radix-2 pass
    static constexpr int NB16 = M / 16 > 0 ? M / 16 : 1;
    static constexpr int BPT16 = (NB16 + TG - 1) / TG;
    static constexpr int NB4 = M / 4 > 0 ? M / 4 : 1;
    static constexpr int BPT4 = (NB4 + TG - 1) / TG;
    static constexpr int NB2 = M / 2 > 0 ? M / 2 : 1;
    static constexpr int BPT2 = (NB2 + TG - 1) / TG;

    template <bool FIRST, class Src, class Dst>
    __device__ static __forceinline__ void pass16(const Src &src, const Dst &dst, int tid, const C *__restrict__ tw, int p)
    {
        C u[BPT16][16];
        // the pass twiddles come from the table in global memory (L1 / L2 hits, hundreds of cycles): issue those loads FIRST,
        // in front of the LDS reads and the barrier, so that they are in flight while the data is fetched — behind the barrier
        // their latency stood in line with the arithmetic in every pass of a single-transform workgroup
        C tw6[BPT16][6];
        if (!FIRST)
        {
#pragma unroll
            for (int b = 0; b < BPT16; b++)
            {
                const int i = tid + b * TG;
                if (NB16 % TG == 0 || i < NB16)
                {
                    const int k = i & (p - 1);
                    const int step = k * ((2 * M) / (16 * p));
                    tw6[b][0] = root<LOG2M>(tw, step);
                    tw6[b][1] = root<LOG2M>(tw, 2 * step);
                    tw6[b][2] = root<LOG2M>(tw, 3 * step);
                    tw6[b][3] = root<LOG2M>(tw, 4 * step);
                    tw6[b][4] = root<LOG2M>(tw, 8 * step);
                    tw6[b][5] = root<LOG2M>(tw, 12 * step);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < BPT16; b++)
        {
            const int i = tid + b * TG;
            if (NB16 % TG == 0 || i < NB16)
            {
                if constexpr (std::is_same<Src, LdsIO<C>>::value && NB16 % 16 == 0 && HCV_FFT_LDS_BASE)
                {
                    // (the padded index of i + r NB16 is that of i plus r (NB16 + NB16 / 16) — NB16 is a multiple of 16 —
                    // so the sixteen reads are one base address and compile-time offsets instead of a shift and an add each)
                    const C *bp = src.s.base + (i + (i >> 4));
#pragma unroll
                    for (int r = 0; r < 16; r++) u[b][r] = bp[r * (NB16 + NB16 / 16)];
                }
                else
                {
#pragma unroll
                    for (int r = 0; r < 16; r++) u[b][r] = src(i + r * NB16);
                }
            }
        }
        if (Src::is_lds && Dst::is_lds) __syncthreads();
#pragma unroll
        for (int b = 0; b < BPT16; b++)
        {
            const int i = tid + b * TG;
            if (NB16 % TG == 0 || i < NB16)
            {
                const int k = i & (p - 1);
                const int j = ((i - k) << 4) + k;
                if (FIRST)
                    dft16<true>(u[b], C(), C(), C());
                else
                {
                    // pass twiddle w^r, w = exp(-2 pi i k / (16 p)) = root(k * 2M / (16 p)); r = 4a + b: w^(4a) here, w^b inside
                    const C w4 = tw6[b][3], w8 = tw6[b][4], w12 = tw6[b][5];
#pragma unroll
                    for (int c = 0; c < 4; c++)
                    {
                        u[b][4 + c] = cmul(u[b][4 + c], w4);
                        u[b][8 + c] = cmul(u[b][8 + c], w8);
                        u[b][12 + c] = cmul(u[b][12 + c], w12);
                    }
                    dft16<false>(u[b], tw6[b][0], tw6[b][1], tw6[b][2]);
                }
                if constexpr (std::is_same<Dst, LdsIO<C>>::value && HCV_FFT_LDS_BASE)
                {
                    // (likewise the writes: bin q goes to j + q p; p = 1 in the first pass with j a multiple of 16, else p is a
                    // multiple of 16: one padded base and q times a per-pass stride)
                    C *bp = dst.s.base + (j + (j >> 4));
                    const int pp = FIRST ? 1 : p + (p >> 4);
#pragma unroll
                    for (int q2 = 0; q2 < 4; q2++)
#pragma unroll
                        for (int q1 = 0; q1 < 4; q1++) bp[(q1 + 4 * q2) * pp] = u[b][4 * q1 + q2];
                }
                else
                {
#pragma unroll
                    for (int q2 = 0; q2 < 4; q2++)
#pragma unroll
                        for (int q1 = 0; q1 < 4; q1++) dst(j + (q1 + 4 * q2) * p, u[b][4 * q1 + q2]);
                }
            }
        }
        if (Dst::is_lds) __syncthreads();
    }

    // Radix-32 FIRST pass (p = 1, every pass twiddle 1) for LOG2M = 4 n + 1: X[q] = E[q] + W32^q O[q], X[q + 16] = E[q] - W32^q O[q]
    // with E / O the 16-point DFTs of a butterfly's even / odd inputs.  It takes the place of the first radix-16 pass AND the
    // radix-2 tail, i.e. one trip through LDS and two barriers less (the engine's 16384-point hop transform: four passes -> three).
    // Half of the TG threads own a butterfly (M / 32 of them), 64 data registers each.
    static constexpr int NB32 = M / 32 > 0 ? M / 32 : 1;
    static constexpr int BPT32 = (NB32 + TG - 1) / TG;
    template <class Src, class Dst>
    __device__ static __forceinline__ void pass32_first(const Src &src, const Dst &dst, int tid)
    {
        typedef decltype(C().x) R;
        // W32^q = (cos, -sin)(2 pi q / 32), q = 0 .. 15
        constexpr double CS[16] = { 1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708, 0.70710678118654752440,
                                    0.55557023301960222474, 0.38268343236508977173, 0.19509032201612826785, 0.0, -0.19509032201612826785,
                                    -0.38268343236508977173, -0.55557023301960222474, -0.70710678118654752440, -0.83146961230254523708,
                                    -0.92387953251128675613, -0.98078528040323044913 };
        constexpr double SN[16] = { 0.0, 0.19509032201612826785, 0.38268343236508977173, 0.55557023301960222474, 0.70710678118654752440,
                                    0.83146961230254523708, 0.92387953251128675613, 0.98078528040323044913, 1.0, 0.98078528040323044913,
                                    0.92387953251128675613, 0.83146961230254523708, 0.70710678118654752440, 0.55557023301960222474,
                                    0.38268343236508977173, 0.19509032201612826785 };
        C e[BPT32][16], o[BPT32][16];
#pragma unroll
        for (int b = 0; b < BPT32; b++)
        {
            const int i = tid + b * TG;
            if (i < NB32)
            {
#pragma unroll
                for (int a = 0; a < 16; a++)
                {
                    e[b][a] = src(i + (2 * a) * NB32);
                    o[b][a] = src(i + (2 * a + 1) * NB32);
                }
            }
        }
        if (Src::is_lds && Dst::is_lds) __syncthreads();
#pragma unroll
        for (int b = 0; b < BPT32; b++)
        {
            const int i = tid + b * TG;
            if (i < NB32)
            {
                dft16<true>(e[b], C(), C(), C());
                dft16<true>(o[b], C(), C(), C());
                const int j = i << 5;
#pragma unroll
                for (int q2 = 0; q2 < 4; q2++)
#pragma unroll
                    for (int q1 = 0; q1 < 4; q1++)
                    {
                        const int q = q1 + 4 * q2, at = 4 * q1 + q2;             // (dft16 leaves bin q1 + 4 q2 in slot 4 q1 + q2)
                        const C t = (q == 0) ? o[b][at] : mulk(o[b][at], (R) CS[q], (R) -SN[q]);
                        if constexpr (std::is_same<Dst, LdsIO<C>>::value && HCV_FFT_LDS_BASE)
                        {
                            C *bp = dst.s.base + (j + (j >> 4));                   // (j is a multiple of 32: offsets q + (q >> 4))
                            bp[q] = C(e[b][at].x + t.x, e[b][at].y + t.y);
                            bp[q + 17] = C(e[b][at].x - t.x, e[b][at].y - t.y);
                        }
                        else
                        {
                            dst(j + q, C(e[b][at].x + t.x, e[b][at].y + t.y));
                            dst(j + q + 16, C(e[b][at].x - t.x, e[b][at].y - t.y));
                        }
                    }
            }
        }
        if (Dst::is_lds) __syncthreads();
    }

    template <class Src, class Dst>
    __device__ static __forceinline__ void pass4(const Src &src, const Dst &dst, int tid, const C *__restrict__ tw, int p)
    {
        C u[BPT4][4];
        C tw3[BPT4][3];                                     // (table loads in front of the LDS reads and the barrier, as in pass16)
#pragma unroll
        for (int b = 0; b < BPT4; b++)
        {
            const int i = tid + b * TG;
            if (NB4 % TG == 0 || i < NB4)
            {
                const int step = (i & (p - 1)) * ((2 * M) / (4 * p));
                tw3[b][0] = root<LOG2M>(tw, step);
                tw3[b][1] = root<LOG2M>(tw, 2 * step);
                tw3[b][2] = root<LOG2M>(tw, 3 * step);
            }
        }
#pragma unroll
        for (int b = 0; b < BPT4; b++)
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
        for (int b = 0; b < BPT4; b++)
        {
            const int i = tid + b * TG;
            if (NB4 % TG == 0 || i < NB4)
            {
                const int k = i & (p - 1);
                const int j = ((i - k) << 2) + k;
                // twiddle exp(-2 pi i k r / (4p)) = root(k * r * (2M / 4p))
                u[b][1] = cmul(u[b][1], tw3[b][0]);
                u[b][2] = cmul(u[b][2], tw3[b][1]);
                u[b][3] = cmul(u[b][3], tw3[b][2]);
                radix4(u[b][0], u[b][1], u[b][2], u[b][3]);
#pragma unroll
                for (int r = 0; r < 4; r++) dst(j + r * p, u[b][r]);
            }
        }
        if (Dst::is_lds) __syncthreads();
    }

    // Radix-8 pass for LOG2M = 4 n + 3: the radix-4 and radix-2 tails in one trip through LDS.  X[q] = E[q] + W8^q O[q],
    // X[q + 4] = E[q] - W8^q O[q] with E / O the 4-point DFTs of the (twiddled) even / odd inputs; two butterflies per thread.
    static constexpr int NB8 = M / 8 > 0 ? M / 8 : 1;
    static constexpr int BPT8 = (NB8 + TG - 1) / TG;
    template <class Src, class Dst>
    __device__ static __forceinline__ void pass8(const Src &src, const Dst &dst, int tid, const C *__restrict__ tw, int p)
    {
        typedef decltype(C().x) R;
        const R h = (R) 0.70710678118654752440;
        C u[BPT8][8];
        C tw7[BPT8][7];                                     // (table loads in front of the LDS reads and the barrier, as in pass16)
#pragma unroll
        for (int b = 0; b < BPT8; b++)
        {
            const int i = tid + b * TG;
            if (NB8 % TG == 0 || i < NB8)
            {
                const int step = (i & (p - 1)) * ((2 * M) / (8 * p));
#pragma unroll
                for (int r = 1; r < 8; r++) tw7[b][r - 1] = root<LOG2M>(tw, r * step);
            }
        }
#pragma unroll
        for (int b = 0; b < BPT8; b++)
        {
            const int i = tid + b * TG;
            if (NB8 % TG == 0 || i < NB8)
            {
#pragma unroll
                for (int r = 0; r < 8; r++) u[b][r] = src(i + r * NB8);
            }
        }
        if (Src::is_lds && Dst::is_lds) __syncthreads();
#pragma unroll
        for (int b = 0; b < BPT8; b++)
        {
            const int i = tid + b * TG;
            if (NB8 % TG == 0 || i < NB8)
            {
                const int k = i & (p - 1);
                const int j = ((i - k) << 3) + k;
#pragma unroll
                for (int r = 1; r < 8; r++) u[b][r] = cmul(u[b][r], tw7[b][r - 1]);
                radix4(u[b][0], u[b][2], u[b][4], u[b][6]);          // E[0..3] in slots 0, 2, 4, 6
                radix4(u[b][1], u[b][3], u[b][5], u[b][7]);          // O[0..3] in slots 1, 3, 5, 7
                const C o0 = u[b][1];
                const C o1 = mulk(u[b][3], h, -h);
                const C o2 = C(u[b][5].y, -u[b][5].x);
                const C o3 = mulk(u[b][7], -h, -h);
                const C e0 = u[b][0], e1 = u[b][2], e2 = u[b][4], e3 = u[b][6];
                dst(j, C(e0.x + o0.x, e0.y + o0.y));
                dst(j + p, C(e1.x + o1.x, e1.y + o1.y));
                dst(j + 2 * p, C(e2.x + o2.x, e2.y + o2.y));
                dst(j + 3 * p, C(e3.x + o3.x, e3.y + o3.y));
                dst(j + 4 * p, C(e0.x - o0.x, e0.y - o0.y));
                dst(j + 5 * p, C(e1.x - o1.x, e1.y - o1.y));
                dst(j + 6 * p, C(e2.x - o2.x, e2.y - o2.y));
                dst(j + 7 * p, C(e3.x - o3.x, e3.y - o3.y));
            }
        }
        if (Dst::is_lds) __syncthreads();
    }

    template <class Src, class Dst>
    __device__ static __forceinline__ void pass2(const Src &src, const Dst &dst, int tid, const C *__restrict__ tw, int p)
    {
        C u[BPT2][2];
        C tw1[BPT2];
#pragma unroll
        for (int b = 0; b < BPT2; b++)
        {
            const int i = tid + b * TG;
            if (NB2 % TG == 0 || i < NB2) tw1[b] = root<LOG2M>(tw, (i & (p - 1)) * ((2 * M) / (2 * p)));
        }
#pragma unroll
        for (int b = 0; b < BPT2; b++)
        {
            const int i = tid + b * TG;
            if (NB2 % TG == 0 || i < NB2)
            {
                u[b][0] = src(i);
                u[b][1] = src(i + NB2);
            }
        }
        if (Src::is_lds && Dst::is_lds) __syncthreads();
#pragma unroll
        for (int b = 0; b < BPT2; b++)
        {
            const int i = tid + b * TG;
            if (NB2 % TG == 0 || i < NB2)
            {
                const int k = i & (p - 1);
                const int j = ((i - k) << 1) + k;
                const C u0 = u[b][0];
                const C u1 = cmul(u[b][1], tw1[b]);
                dst(j, C(u0.x + u1.x, u0.y + u1.y));
                dst(j + p, C(u0.x - u1.x, u0.y - u1.y));
            }
        }
        if (Dst::is_lds) __syncthreads();
    }

    // Transform with the input taken from `ld` and the result handed to `st`; s is the scratch between the passes.
    // If ld reads LDS (LdsIO or a functor over it) the caller has already synchronised after filling it.
    template <class Ld, class St>
    __device__ static __forceinline__ void run(const Ld &ld, const St &st, LdsBuf<C> s, int tid, const C *__restrict__ tw)
    {
        const LdsIO<C> io = { s };
        constexpr bool TAIL4 = REM >= 2, TAIL2 = (REM & 1) != 0;
        int p = 1;
#if HCV_FFT_RADIX32
        if constexpr (TAIL2 && !TAIL4 && N16 >= 1)
        {
            // LOG2M = 4 n + 1 (32, 512, 8192 points): radix-32 first, then n - 1 radix-16 passes.  (For 4 n + 3 — radix-32,
            // radix-16s, radix-4 against radix-16s, radix-4, radix-2 — it measured equal to 5 % slower: 2^11 complex +1 %, the real
            // 2^12 -5 %, c2 0.031 -> 0.0326 ms; those sizes keep their tails.)
            if constexpr (N16 == 1)
            {
                pass32_first(ld, st, tid);
                return;
            }
            else
            {
                pass32_first(ld, io, tid);
                p = 32;
#pragma unroll 1
                for (int pass = 0; pass < N16 - 2; pass++, p <<= 4) pass16<false>(io, io, tid, tw, p);
                pass16<false>(io, st, tid, tw, p);
                return;
            }
        }
#endif
        if constexpr (N16 > 0)
        {
            if constexpr (N16 == 1 && !TAIL4 && !TAIL2)
            {
                pass16<true>(ld, st, tid, tw, 1);
                return;
            }
            else
            {
                pass16<true>(ld, io, tid, tw, 1);
                p = 16;
                constexpr int INNER = (TAIL4 || TAIL2) ? N16 - 1 : N16 - 2;     // LDS -> LDS radix-16 passes
#pragma unroll 1
                for (int pass = 0; pass < INNER; pass++, p <<= 4) pass16<false>(io, io, tid, tw, p);
                if constexpr (!TAIL4 && !TAIL2)
                {
                    pass16<false>(io, st, tid, tw, p);
                    return;
                }
            }
        }
#if HCV_FFT_RADIX8
        if constexpr (TAIL4 && TAIL2 && N16 >= 1 && !St::is_lds && sizeof(C) == 8)
        {
            pass8(io, st, tid, tw, p);
            return;
        }
#endif
        if constexpr (TAIL4)
        {
            if constexpr (N16 == 0)
            {
                if constexpr (TAIL2) pass4(ld, io, tid, tw, p); else pass4(ld, st, tid, tw, p);
            }
            else
            {
                if constexpr (TAIL2) pass4(io, io, tid, tw, p); else pass4(io, st, tid, tw, p);
            }
            p <<= 2;
        }
        if constexpr (TAIL2)
        {
            if constexpr (N16 == 0 && !TAIL4) pass2(ld, st, tid, tw, p); else pass2(io, st, tid, tw, p);
        }
    }

    // in place on data already in LDS (and synchronised); leaves the result in LDS, synchronised
    __device__ static __forceinline__ void run(LdsBuf<C> s, int tid, const C *__restrict__ tw)
    {
        const LdsIO<C> io = { s };
        run(io, io, s, tid, tw);
    }
};

// Threads cooperating on one transform and transforms per workgroup: one radix-16 butterfly per thread, up to 1024
// threads (16 waves keep enough loads in flight when a single 8192- or 16384-point transform owns the CU's LDS), in
// workgroups of at least WG threads.  (Measured alternatives: M/4 threads for short transforms and 64-thread workgroups —
// equal on the engine's workloads, slower on batched transforms.)
template <int LOG2M, int WG = 256> struct FFTGeom
{
    static constexpr int M = 1 << LOG2M;
    static constexpr int TG = (M / 16) < 1 ? 1 : ((M / 16) < 1024 ? (M / 16) : 1024);
    static constexpr int THREADS = TG > WG ? TG : WG;
    static constexpr int G = THREADS / TG;
};


// Tile of one four-step workgroup: TILE adjacent columns (or rows) of P points each, all transformed at once by
// TILE * TG threads (at most 1024).  TILE aims at 128-byte runs in each of the split arrays, within 128 KiB of LDS
// (136 KiB with the bank padding).
#ifndef HCV_FX_TILE_CAP64
#define HCV_FX_TILE_CAP64 1
#endif
#ifndef HCV_FX_TILE_BYTES
#define HCV_FX_TILE_BYTES 256
#endif
#ifndef HCV_FX_TILE_MINRUN
#define HCV_FX_TILE_MINRUN 64       // bytes of complex elements per tile line below which a 64 KiB tile is not worth its second workgroup
#endif
#ifndef HCV_FX_XCD_ORDER
#define HCV_FX_XCD_ORDER 1          // workgroup b of a pass takes tile (b % 8) * (tiles / 8) + b / 8: the tiles an XCD holds at one time
                                    // are NEIGHBOURS, so the partial cache lines of their strided runs meet in that XCD's L2
#endif
template <int P, int ELEM_BYTES> struct FourStepTile
{
    static constexpr int TG = P / 16 < 256 ? P / 16 : 256;
    static constexpr int WANT = HCV_FX_TILE_BYTES / ELEM_BYTES;         // complex elements: 128 bytes per split array
    // LDS budget per tile: 64 KiB (two workgroups per CU) while that still leaves 64-byte runs, else 128 KiB (one)
    static constexpr int CAP64 = 64 * 1024 / (P * ELEM_BYTES), CAP128 = 128 * 1024 / (P * ELEM_BYTES);
    static constexpr int CAP = (HCV_FX_TILE_CAP64 && CAP64 * ELEM_BYTES >= HCV_FX_TILE_MINRUN) ? CAP64 : CAP128;
    static constexpr int TILE = CAP < WANT ? CAP : WANT;
    static constexpr int THREADS = TILE * TG < 1024 ? TILE * TG : 1024;
    static constexpr int G = THREADS / TG;                              // sub-transforms in flight
    static_assert(TILE % G == 0 && THREADS % 64 == 0, "tile geometry");
    // workgroups of this tile a CU holds (160 KiB of LDS, 2048 threads): the register budget the kernels ask for
    static constexpr int LDS_BYTES = TILE * (P + (P >> 4) + 1) * ELEM_BYTES;
    static constexpr int BY_LDS = 160 * 1024 / LDS_BYTES, BY_THREADS = 2048 / THREADS;
    static constexpr int PER_CU = BY_LDS < 1 ? 1 : (BY_LDS < BY_THREADS ? BY_LDS : BY_THREADS);
    static constexpr int WAVES_PER_SIMD = PER_CU * THREADS / 256 < 1 ? 1 : PER_CU * THREADS / 256;   // (second launch bound, HIP semantics)
};

// Workgroups are handed to the eight XCDs round robin (linear workgroup id mod 8).  With the plain order the tiles that run at one
// time on ONE XCD are eight apart, and each 32- or 64-byte run of a strided tile line is a partial cache line in that XCD's L2
// whose other parts are fetched (and written back) by other XCDs.
__device__ __forceinline__ int fourstep_tile_of(int b, int tiles)
{
#if HCV_FX_XCD_ORDER
    if ((tiles & 7) == 0) return (b & 7) * (tiles >> 3) + (b >> 3);
#endif
    return b;
}

} // namespace hcv
