// Residue-split real transforms: ONE hop transform spread over R/2 + 1 (forward) or R/2 (inverse) workgroups that never talk to
// each other — for the engines whose block is a handful of transforms (1 x 1, 8 -> 1), where a single-workgroup transform is bound
// by ONE CU's instruction issue (the 16384-point real transform: 1024 threads, ~2500 vector instructions each, 8-14 us on one of
// 256 CUs; profiles/r02d_c{1,3}_kernel_summary.txt).  Replaces, for those blocks, hisstools_rfft / hisstools_rifft as
// PartitionedConvolve::process calls them per hop (PartitionedConvolve.cpp:304-307,350-360; HISSTools_FFT.cpp:226-248).
//
// Forward (decimation in frequency, first stage only): with N = R S and W_K = exp(-2 pi i / K),
//     X[R k + r] = FFT_S( y_r )[k],      y_r[n] = W_N^(n r) * sum_{q < R} f[n + S q] W_R^(q r),     n < S
// so residue class r of the spectrum is ONE S-point complex transform of a sequence every workgroup can form from the real frame f
// by itself (R real-times-constant products per point).  f is real, so X[N - b] = conj X[b]: the workgroup of residue r also
// delivers residue R - r (bins R (S - 1 - k) + R - r = conj of its k >= S/2 half), and r = 0 .. R/2 cover the half spectrum —
// R/2 + 1 workgroups doing together the work of the one packed M-point transform they replace, none of it twice except half of
// the two self-paired classes.  Output = the engine's packed format (x2 scale; bin 0 = (DC, Nyquist)), natural bin order: nothing
// downstream changes.
//
// Inverse (decimation in time, last stage only): with e = R n1 + n2, b = b1 + S b2 and V_K = exp(+2 pi i / K),
//     x[R n1 + n2] = IFFT_S( g_n2 )[n1],   g_n2[b1] = V_N^(n2 b1) * sum_{b2 < R} Yfull[b1 + S b2] V_R^(n2 b2)
// so the samples of residue class n2 are one S-point transform of a sequence formed from the whole (Hermitian-extended) spectrum;
// x is real, so two classes share a complex transform (u = g_a + i g_b): R/2 workgroups per output.  Only the second half of the
// frame is delivered (overlap-save), scaled 1 / (4N) (scaleStore, PartitionedConvolve.cpp:232-241).
//
// The S-point transforms run in LDS as radix-4 Stockham passes on all 256 threads, their twiddle table staged in LDS beside the data.

#include "hcv_engine.h"
#include "hcv_fft_device.h"

#include <cstdlib>
#include <string>

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
    template <int LOG2S, int LOG2R> struct SplitSpectrumStore
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
                if (k == 0) dst[0].x = v.x;                          // 2 X[0]
                else if (k == HALF) dst[0].y = v.x;                  // 2 X[N/2]
                else if (k < HALF) dst[R * k] = v;
            }
            else if (k < HALF)
                dst[R * k + r] = v;
            else if (r != R / 2)
                dst[R * (S - 1 - k) + (R - r)] = make_float2(v.x, -v.y);
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

// One workgroup = (transform q = (t, i), residue r).  DIRECT: the new hop of the frame comes from the caller's block (positions
// >= n0) and the workgroup of residue 0 files it in the history ring, as rfft_frames_direct_kernel does.
template <int LOG2N, int LOG2R, bool DIRECT>
__global__ __launch_bounds__(256) void rfft_split_kernel(float *__restrict__ hist, long long hist_stride, long long hist_mask, const float *__restrict__ in,
                                                          long long in_stride, long long n0, long long h_first, int nin, float2 *__restrict__ X, int Rring,
                                                          const float2 *__restrict__ tw, const float2 *__restrict__ tws, int pin)
{
    constexpr int N = 1 << LOG2N, M = N / 2, R = 1 << LOG2R, LOG2S = LOG2N - LOG2R, S = 1 << LOG2S, NW = R / 2 + 1, TG = 256;
    __shared__ __attribute__((aligned(16))) float2 lds[lds_padded(S)];
    __shared__ __attribute__((aligned(16))) float2 tl[S];
    __shared__ float2 wr[R];
    const int tid = threadIdx.x;
    int bx = blockIdx.x;
    if (pin >= 0)
    {
        if ((bx & 7) != pin) return;
        bx >>= 3;
    }
    const int r = bx % NW, q = bx / NW;
    const int t = q / nin, i = q % nin;
    const long long h = h_first + t;
    const LdsBuf<float2> s = { lds };
    stage_table<LOG2S, TG>(tl, tws, tid);
    if (tid < R)
    {
        float sn, cs;
        sincospif(-2.0f * (float) tid / (float) R, &sn, &cs);        // W_R^tid
        wr[tid] = make_float2(cs, sn);
    }
    __syncthreads();

    float *hrow = hist + (long long) i * hist_stride;
    const float *irow = in + (long long) i * in_stride;
    const long long base = (h - 1) * (long long) M;
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
                *reinterpret_cast<float4 *>(hrow + ((base + 4 * v + S * qq) & hist_mask)) = f[qq];
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
    const int slot = (int) (h % Rring);
    const SplitSpectrumStore<LOG2S, LOG2R> st = { X + ((long long) i * Rring + slot) * M, r };
    sub_fft<LOG2S, TG>(s, st, tid, tl);
}

// One workgroup = (transform q = (t, o), sample classes n2 = 2 j, 2 j + 1).  Y: [ksplit][T][nout][M] partial sums (added up while
// the spectrum is staged into LDS).
template <int LOG2N, int LOG2R>
__global__ __launch_bounds__(256) void rifft_split_emit_kernel(const float2 *__restrict__ Y, int ksplit, long long ks_stride, int nout, float *__restrict__ out,
                                                                long long out_stride, const float2 *__restrict__ tw, const float2 *__restrict__ tws, int pin)
{
    constexpr int N = 1 << LOG2N, M = N / 2, R = 1 << LOG2R, LOG2S = LOG2N - LOG2R, S = 1 << LOG2S, NW = R / 2, TG = 256;
    extern __shared__ __attribute__((aligned(16))) float2 dyn[];
    float2 *spec = dyn;                                  // [M] the packed half spectrum
    const LdsBuf<float2> s = { dyn + M };                // [lds_padded(S)] the class pair's transform
    float2 *tl = dyn + M + lds_padded(S);                // [S] the sub-transform's twiddles
    float2 *wr = tl + S;                                 // [R] W_R^j
    const int tid = threadIdx.x;
    int bx = blockIdx.x;
    if (pin >= 0)
    {
        if ((bx & 7) != pin) return;
        bx >>= 3;
    }
    const int j = bx % NW, q = bx / NW;
    const int t = q / nout, o = q % nout;
    if (tid < R)
    {
        float sn, cs;
        sincospif(-2.0f * (float) tid / (float) R, &sn, &cs);
        wr[tid] = make_float2(cs, sn);
    }
    stage_table<LOG2S, TG>(tl, tws, tid);
    {
        // stage (and add up) the spectrum: 16-byte loads, every load of a slice in flight before the adds
        const float4 *src = reinterpret_cast<const float4 *>(Y + ((long long) t * nout + o) * M);
        float4 *dst4 = reinterpret_cast<float4 *>(spec);
        constexpr int V = M / 2 / TG;                    // float4 per thread
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
    const SplitSampleStore<LOG2S, LOG2R> st = { out + (long long) o * out_stride + (long long) t * M - M, 1.f / (float) (8 * M), j };
    sub_fft<LOG2S, TG>(s, st, tid, tl);
}

// ------------------------------------------------------------------------------------------------ launchers

static const float2 *sub_table(int log2s)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::string err;
    return twiddles(dev, log2s + 1, &err);               // (2 S)-th roots, S entries: what LdsFFT<LOG2S> indexes
}

// HCV_FFT_SPLIT: 0 = never, 1 = wherever a split kernel exists, unset = blocks of at most kSplitMaxTransforms transforms (16384 or 4096 points)
static int split_mode()
{
    static const int m = std::getenv("HCV_FFT_SPLIT") ? std::atoi(std::getenv("HCV_FFT_SPLIT")) : -1;
    return m;
}
constexpr int kSplitMaxTransforms = 16;

bool fft_split_applies(int log2n, int transforms)
{
    const int m = split_mode();
    if (m == 0 || transforms <= 0) return false;
    if (log2n != 14 && log2n != 12) return false;
    if (m > 0) return transforms <= 1024;
    return transforms <= kSplitMaxTransforms;
}

static int split_radix_log2(int log2n)
{
    static const int r14 = std::getenv("HCV_FFT_SPLIT_R14") ? std::atoi(std::getenv("HCV_FFT_SPLIT_R14")) : 4;
    static const int r12 = std::getenv("HCV_FFT_SPLIT_R12") ? std::atoi(std::getenv("HCV_FFT_SPLIT_R12")) : 3;
    if (log2n == 14) return (r14 == 3 || r14 == 5) ? r14 : 4;
    return (r12 == 2 || r12 == 4) ? r12 : 3;
}

void fft_split_prepare(int log2n)
{
    if (log2n == 14 || log2n == 12) (void) sub_table(log2n - split_radix_log2(log2n));
}

template <int LOG2N, int LOG2R>
static hipError_t launch_rfft_split_t(float *hist, long long hist_stride, long long hist_mask, const float *in, long long in_stride, long long n0,
                                      long long h_first, int T, int nin, float2 *X, int R, const float2 *tw, hipStream_t st)
{
    const float2 *tws = sub_table(LOG2N - LOG2R);
    if (!tws) return hipErrorInvalidValue;
    constexpr int NW = (1 << LOG2R) / 2 + 1;
    const int pin = xcd_pin_for((long long) NW * T * nin);
    hipLaunchKernelGGL((rfft_split_kernel<LOG2N, LOG2R, true>), dim3(NW * T * nin * (pin >= 0 ? 8 : 1)), dim3(256), 0, st, hist, hist_stride, hist_mask, in, in_stride,
                       n0, h_first, nin, X, R, tw, tws, pin);
    return hipGetLastError();
}

hipError_t launch_rfft_frames_direct_split(int log2n, float *hist, long long hist_stride, long long hist_mask, const float *in, long long in_stride, long long n0,
                                           long long h_first, int T, int nin, float2 *X, int R, const float2 *tw, hipStream_t st)
{
    const int lr = split_radix_log2(log2n);
#define HCV_SPLIT_F(LN, LR) if (log2n == LN && lr == LR) return launch_rfft_split_t<LN, LR>(hist, hist_stride, hist_mask, in, in_stride, n0, h_first, T, nin, X, R, tw, st)
    HCV_SPLIT_F(14, 3); HCV_SPLIT_F(14, 4); HCV_SPLIT_F(14, 5); HCV_SPLIT_F(12, 2); HCV_SPLIT_F(12, 3); HCV_SPLIT_F(12, 4);
#undef HCV_SPLIT_F
    return hipErrorInvalidValue;
}

template <int LOG2N, int LOG2R>
static hipError_t launch_rifft_split_t(const float2 *Y, int ksplit, long long ks_stride, int T, int nout, float *out, long long out_stride, const float2 *tw,
                                       hipStream_t st)
{
    const float2 *tws = sub_table(LOG2N - LOG2R);
    if (!tws) return hipErrorInvalidValue;
    constexpr int M = 1 << (LOG2N - 1), S = 1 << (LOG2N - LOG2R), R = 1 << LOG2R, NW = R / 2;
    constexpr size_t lds = sizeof(float2) * (size_t) (M + lds_padded(S) + S + R);
    if (lds > 48 * 1024)
    {
        // (more than the default dynamic LDS must be asked for, once per device and instantiation)
        static bool allowed[64] = {};
        int dev = 0;
        (void) hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !allowed[dev])
        {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rifft_split_emit_kernel<LOG2N, LOG2R>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e != hipSuccess) return e;
            if (dev >= 0 && dev < 64) allowed[dev] = true;
        }
    }
    const int pin = xcd_pin_for((long long) NW * T * nout);
    hipLaunchKernelGGL((rifft_split_emit_kernel<LOG2N, LOG2R>), dim3(NW * T * nout * (pin >= 0 ? 8 : 1)), dim3(256), lds, st, Y, ksplit, ks_stride, nout, out, out_stride,
                       tw, tws, pin);
    return hipGetLastError();
}

hipError_t launch_rifft_emit_split(int log2n, const float2 *Y, int ksplit, long long ks_stride, int T, int nout, float *out, long long out_stride,
                                   const float2 *tw, hipStream_t st)
{
    const int lr = split_radix_log2(log2n);
#define HCV_SPLIT_I(LN, LR) if (log2n == LN && lr == LR) return launch_rifft_split_t<LN, LR>(Y, ksplit, ks_stride, T, nout, out, out_stride, tw, st)
    HCV_SPLIT_I(14, 3); HCV_SPLIT_I(14, 4); HCV_SPLIT_I(14, 5); HCV_SPLIT_I(12, 2); HCV_SPLIT_I(12, 3); HCV_SPLIT_I(12, 4);
#undef HCV_SPLIT_I
    return hipErrorInvalidValue;
}

} // namespace hcv
