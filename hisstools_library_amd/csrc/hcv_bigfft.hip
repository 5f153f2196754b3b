// Real FFTs for N = 2^16 .. 2^20 (PartitionedConvolve.h:19 allows FFT sizes up to 2^20): the complex transform of
// M = N/2 points no longer fits the CU's LDS, so it runs as a four-step transform through HBM
//
//     M = M1 * M2,   n = M2*n1 + n2,   k = k1 + M1*k2
//     1. "cols":  for every n2: M1-point FFT over n1 (LDS), times the twiddle W_M^(n2*k1)      Z[n1][n2] -> T[k1][n2]
//     2. "rows":  for every k1: M2-point FFT over n2 (LDS)                                     T[k1][n2] -> Z[k1 + M1*k2]
//
// with the same in-LDS Stockham kernels (hcv_fft_device.h) for the M1- and M2-point pieces (both <= 1024 points),
// the same packed, doubled half-spectrum format and the same scaling as the LDS-resident transforms.  These sizes
// mean hops of >= 32768 samples; they are rare, so the passes favour simplicity over the last percent.

#include "hcv_kernels.h"
#include "hcv_fft_device.h"

#include <algorithm>

namespace hcv
{

__device__ __forceinline__ float2 root_rt(const float2 *__restrict__ tw, int idx, int M)
{
    float2 w = tw[idx & (M - 1)];
    return (idx & M) ? make_float2(-w.x, -w.y) : w;
}

void big_fft_split(int log2n, int &l1, int &l2)
{
    const int lm = log2n - 1;
    l2 = (lm + 1) / 2;
    l1 = lm - l2;
}

// ------------------------------------------------------------------------------------------------ load

struct BigLoad
{
    const float *src;
    long long stride, mask, h_first, count;
    int nin, mode;                 // 0 = history frames, 1 = IR partitions, 2 = plain rows
};

// Z[q - q0][n] = (x[2n], x[2n+1]) of transform q, zero padded
__global__ void big_load_kernel(BigLoad a, float2 *__restrict__ Z, int M, int q0)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= M) return;
    const int q = q0 + blockIdx.y;
    float2 v;
    if (a.mode == 0)
    {
        const int t = q / a.nin, i = q % a.nin;
        const long long pos = ((a.h_first + t - 1) * (long long) M + 2LL * n) & a.mask;
        v = *reinterpret_cast<const float2 *>(a.src + (long long) i * a.stride + pos);
    }
    else if (a.mode == 1)
    {
        const long long s0 = (long long) q * M + 2LL * n;
        v.x = (2 * n < M && s0 < a.count) ? a.src[s0] : 0.f;
        v.y = (2 * n + 1 < M && s0 + 1 < a.count) ? a.src[s0 + 1] : 0.f;
    }
    else
    {
        const float *row = a.src + (long long) q * a.stride;
        v.x = (2LL * n < a.count) ? row[2 * n] : 0.f;
        v.y = (2LL * n + 1 < a.count) ? row[2 * n + 1] : 0.f;
    }
    Z[(long long) blockIdx.y * M + n] = v;
}

// ------------------------------------------------------------------------------------------------ four-step passes

// tiles as in hcv_fftx.hip (FourStepTile): runs of 256 bytes per row of the tile, one thread group per column / row, up to
// 1024 threads
template <int L1>
__global__ __launch_bounds__((FourStepTile<(1 << L1), 8>::THREADS)) void big_cols_kernel(const float2 *__restrict__ Zin, float2 *__restrict__ Tout, int M2, int M,
                                                       const float2 *__restrict__ tw1, const float2 *__restrict__ twN)
{
    constexpr int M1 = 1 << L1;
    typedef FourStepTile<M1, 8> Tile;
    constexpr int TG = Tile::TG, G = Tile::G, BIG_COLS = Tile::TILE, NT = Tile::THREADS;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];   // [BIG_COLS][M1]

    const int col0 = fourstep_tile_of(blockIdx.x, gridDim.x) * BIG_COLS;
    const float2 *zin = Zin + (long long) blockIdx.y * M;
    float2 *tout = Tout + (long long) blockIdx.y * M;

    for (int e = threadIdx.x; e < BIG_COLS * M1; e += NT)
    {
        const int c = e % BIG_COLS, n1 = e / BIG_COLS;
        LdsBuf<float2>{ lds + c * fourstep_pitch(M1) }[n1] = zin[(long long) n1 * M2 + col0 + c];
    }
    __syncthreads();
    const int g = threadIdx.x / TG, t = threadIdx.x % TG;
    for (int c0 = 0; c0 < BIG_COLS; c0 += G) LdsFFT<L1, TG>::run(LdsBuf<float2>{ lds + (c0 + g) * fourstep_pitch(M1) }, t, tw1);
    // W_M^(n2 k1) = (N-th root)^(2 n2 k1), computed rather than gathered from the N/2-entry table (one scattered 8-byte load
    // per element of the tile bound this pass).  A thread's elements are its column at k1 = k10 + i DK: W^(n2 k10) and
    // S, S^2, S^4, S^8 (S = W^(n2 DK)) each from sincospi of an exactly reduced fraction (within 2 ulp), the sixteen twiddles
    // as products of at most four of them — five calls instead of sixteen (as fx_cols_kernel, hcv_fftx.hip).
    constexpr int NI = BIG_COLS * M1 / NT, DK = NT / BIG_COLS;
    static_assert((BIG_COLS * M1) % NT == 0 && NT % BIG_COLS == 0 && (NI & (NI - 1)) == 0 && NI <= 16, "tile geometry");
    auto root = [&](int idx)
    {
        float sn, cs;
        sincospif(-(float) (idx & (2 * M - 1)) * (1.0f / (float) M), &sn, &cs);       // (M is a power of two: the reciprocal and the product are exact, and a float division is ten instructions)
        return make_float2(cs, sn);
    };
    const int c = (int) threadIdx.x % BIG_COLS, k10 = (int) threadIdx.x / BIG_COLS;
    const int n2 = col0 + c;
    float2 w[NI];
    w[0] = root(2 * n2 * k10);
#pragma unroll
    for (int bit = 1; bit < NI; bit *= 2)
    {
        const float2 sb = root(2 * n2 * DK * bit);                  // (not S squared: that would double S's error every time)
#pragma unroll
        for (int i = 0; i < bit; i++) w[bit + i] = cmul(w[i], sb);
    }
#pragma unroll
    for (int i = 0; i < NI; i++)
    {
        const int k1 = k10 + i * DK;
        tout[(long long) k1 * M2 + n2] = cmul(LdsBuf<float2>{ lds + c * fourstep_pitch(M1) }[k1], w[i]);
    }
}

template <int L2>
__global__ __launch_bounds__((FourStepTile<(1 << L2), 8>::THREADS)) void big_rows_kernel(const float2 *__restrict__ Tin, float2 *__restrict__ Zout, int M1, int M,
                                                       const float2 *__restrict__ tw2)
{
    constexpr int M2 = 1 << L2;
    typedef FourStepTile<M2, 8> Tile;
    constexpr int TG = Tile::TG, G = Tile::G, BIG_ROWS = Tile::TILE, NT = Tile::THREADS;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];   // [BIG_ROWS][M2]

    const int row0 = fourstep_tile_of(blockIdx.x, gridDim.x) * BIG_ROWS;
    const float2 *tin = Tin + (long long) blockIdx.y * M + (long long) row0 * M2;
    float2 *zout = Zout + (long long) blockIdx.y * M;

    for (int e = threadIdx.x; e < BIG_ROWS * M2; e += NT) LdsBuf<float2>{ lds + (e / M2) * fourstep_pitch(M2) }[e % M2] = tin[e];
    __syncthreads();
    const int g = threadIdx.x / TG, t = threadIdx.x % TG;
    for (int r0 = 0; r0 < BIG_ROWS; r0 += G)
    {
        LdsFFT<L2, TG>::run(LdsBuf<float2>{ lds + (r0 + g) * fourstep_pitch(M2) }, t, tw2);
    }
    for (int e = threadIdx.x; e < BIG_ROWS * M2; e += NT)
    {
        const int r = e % BIG_ROWS, k2 = e / BIG_ROWS;
        zout[(long long) (row0 + r) + (long long) M1 * k2] = LdsBuf<float2>{ lds + r * fourstep_pitch(M2) }[k2];
    }
}

// ------------------------------------------------------------------------------------------------ real passes

struct BigStore
{
    float2 *dst;
    long long stride, h_first;
    int nin, R, mode;              // 0 = dst + q*stride, 1 = input-spectrum ring slot of frame q = (t, i)
};

__device__ __forceinline__ float2 *big_dst(const BigStore &s, int q, int M)
{
    if (s.mode == 0) return s.dst + (long long) q * s.stride;
    const int t = q / s.nin, i = q % s.nin;
    const int slot = (int) ((s.h_first + t) % s.R);
    return s.dst + ((long long) i * s.R + slot) * M;
}

// forward: Z = FFT_M(x_even + i x_odd) -> packed, doubled half spectrum (maths of real_post_store in hcv_kernels.hip)
__global__ void big_real_post_kernel(const float2 *__restrict__ Z, BigStore s, int M, int q0, const float2 *__restrict__ twN)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > M / 2) return;
    const float2 *z = Z + (long long) blockIdx.y * M;
    float2 *dst = big_dst(s, q0 + blockIdx.y, M);
    if (k == 0)
    {
        const float2 v = z[0];
        const float t1 = v.x + v.y, t2 = v.x - v.y;
        dst[0] = make_float2(t1 + t1, t2 + t2);
        return;
    }
    const int m = M - k;
    const float2 w = twN[k];
    const float2 z1 = z[k], z2 = z[m];
    const float r3 = z1.x + z2.x, i3 = z1.y + z2.y, r4 = z1.x - z2.x, i4 = z1.y - z2.y;
    const float u1 = (w.x * i3) + (w.y * r4);
    const float u2 = (w.y * i3) - (w.x * r4);
    dst[k] = make_float2(r3 + u1, u2 + i4);
    dst[m] = make_float2(r3 - u1, u2 - i4);
}

// inverse: packed spectrum -> (im, re)-swapped complex input of the forward transform (real_pre_inverse)
__global__ void big_real_pre_kernel(const float2 *__restrict__ Y, long long y_stride, float2 *__restrict__ Z, int M, int q0,
                                    const float2 *__restrict__ twN)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > M / 2) return;
    const float2 *y = Y + (long long) (q0 + blockIdx.y) * y_stride;
    float2 *z = Z + (long long) blockIdx.y * M;
    if (k == 0)
    {
        const float2 v = y[0];
        z[0] = make_float2(v.x - v.y, v.x + v.y);
        return;
    }
    const int m = M - k;
    const float2 w = twN[k];
    const float c = -w.x, sn = w.y;
    const float2 z1 = y[k], z2 = y[m];
    const float r3 = z1.x + z2.x, i3 = z1.y + z2.y, r4 = z1.x - z2.x, i4 = z1.y - z2.y;
    const float u1 = (c * i3) + (sn * r4);
    const float u2 = (sn * i3) - (c * r4);
    z[k] = make_float2(u2 + i4, r3 + u1);
    z[m] = make_float2(u2 - i4, r3 - u1);
}

// inverse epilogue A: valid half-frame, scaled by 1/(4N), added to the timeline at the hop's emission time
__global__ void big_overlap_add_kernel(const float2 *__restrict__ Z, int M, int q0, int nout, long long h_first, float *__restrict__ timeline,
                                       long long tl_stride, long long tl_mask)
{
    const int k = M / 2 + blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    const int q = q0 + blockIdx.y;
    const int t = q / nout, o = q % nout;
    const float2 v = Z[(long long) blockIdx.y * M + k];             // (x[2k+1], x[2k])
    const float scale = 1.f / (8.f * (float) M);
    const long long pos = ((h_first + t + 1) * (long long) M + 2LL * k - M) & tl_mask;
    float2 *d = reinterpret_cast<float2 *>(timeline + (long long) o * tl_stride + pos);
    float2 cur = *d;
    cur.x += v.y * scale;
    cur.y += v.x * scale;
    *d = cur;
}

// inverse epilogue B: plain rows of 2M samples, unnormalised
__global__ void big_store_rows_kernel(const float2 *__restrict__ Z, int M, int q0, float *__restrict__ dst)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    const float2 v = Z[(long long) blockIdx.y * M + k];
    reinterpret_cast<float2 *>(dst + (long long) (q0 + blockIdx.y) * 2 * M)[k] = make_float2(v.y, v.x);
}

// ------------------------------------------------------------------------------------------------ host side

static hipError_t big_cfft(int log2n, float2 *a, float2 *b, int batch, const BigFFTWork &w, const float2 *twN, hipStream_t st)
{
    int l1, l2;
    big_fft_split(log2n, l1, l2);
    const int M = 1 << (log2n - 1), M1 = 1 << l1, M2 = 1 << l2;
    size_t lds1 = 0, lds2 = 0;
    dim3 gc, gr, bc, br;
#define HCV_BIG_COLS(L)                                                                                                \
    case L:                                                                                                            \
        lds1 = sizeof(float2) * FourStepTile<(1 << L), 8>::TILE * fourstep_pitch(M1);                                      \
        gc = dim3(M2 / FourStepTile<(1 << L), 8>::TILE, batch);                                                        \
        bc = dim3(FourStepTile<(1 << L), 8>::THREADS);                                                                 \
        if (lds1 > 48 * 1024) (void) hipFuncSetAttribute(reinterpret_cast<const void *>(big_cols_kernel<L>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds1); \
        hipLaunchKernelGGL(big_cols_kernel<L>, gc, bc, lds1, st, a, b, M2, M, w.tw1, twN);                           \
        break;
    switch (l1)
    {
        HCV_BIG_COLS(7) HCV_BIG_COLS(8) HCV_BIG_COLS(9)
        default: return hipErrorInvalidValue;
    }
#undef HCV_BIG_COLS
#define HCV_BIG_ROWS(L)                                                                                                \
    case L:                                                                                                            \
        lds2 = sizeof(float2) * FourStepTile<(1 << L), 8>::TILE * fourstep_pitch(M2);                                      \
        gr = dim3(M1 / FourStepTile<(1 << L), 8>::TILE, batch);                                                        \
        br = dim3(FourStepTile<(1 << L), 8>::THREADS);                                                                 \
        if (lds2 > 48 * 1024) (void) hipFuncSetAttribute(reinterpret_cast<const void *>(big_rows_kernel<L>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds2); \
        hipLaunchKernelGGL(big_rows_kernel<L>, gr, br, lds2, st, b, a, M1, M, w.tw2);                                 \
        break;
    switch (l2)
    {
        HCV_BIG_ROWS(8) HCV_BIG_ROWS(9) HCV_BIG_ROWS(10)
        default: return hipErrorInvalidValue;
    }
#undef HCV_BIG_ROWS
    return hipGetLastError();
}

static int big_chunk(const BigFFTWork &w, int M) { return (int) std::max<size_t>(1, w.elems / (size_t) M); }

static hipError_t big_forward(int log2n, const BigLoad &ld, const BigStore &sv, int batch, const float2 *twN, const BigFFTWork &w, hipStream_t st)
{
    if (!w.a || !w.b) return hipErrorInvalidValue;
    const int M = 1 << (log2n - 1);
    const int chunk = big_chunk(w, M);
    for (int q0 = 0; q0 < batch; q0 += chunk)
    {
        const int nb = std::min(chunk, batch - q0);
        hipLaunchKernelGGL(big_load_kernel, dim3((M + 255) / 256, nb), dim3(256), 0, st, ld, w.a, M, q0);
        hipError_t e = big_cfft(log2n, w.a, w.b, nb, w, twN, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(big_real_post_kernel, dim3((M / 2 + 1 + 255) / 256, nb), dim3(256), 0, st, w.a, sv, M, q0, twN);
    }
    return hipGetLastError();
}

hipError_t big_rfft_frames(int log2n, const float *hist, long long hist_stride, long long hist_mask, long long h_first, int T, int nin, float2 *X, int R,
                           const float2 *tw, const BigFFTWork &w, hipStream_t st)
{
    BigLoad ld = { hist, hist_stride, hist_mask, h_first, 0, nin, 0 };
    BigStore sv = { X, 0, h_first, nin, R, 1 };
    return big_forward(log2n, ld, sv, T * nin, tw, w, st);
}

hipError_t big_rfft_ir(int log2n, const float *src, long long count, int P, float2 *dst, const float2 *tw, const BigFFTWork &w, hipStream_t st)
{
    BigLoad ld = { src, 0, 0, 0, count, 1, 1 };
    BigStore sv = { dst, (long long) 1 << (log2n - 1), 0, 1, 1, 0 };
    return big_forward(log2n, ld, sv, P, tw, w, st);
}

hipError_t big_rfft_rows(int log2n, const float *src, long long src_stride, long long in_len, int batch, float2 *dst, const float2 *tw,
                         const BigFFTWork &w, hipStream_t st)
{
    BigLoad ld = { src, src_stride, 0, 0, in_len, 1, 2 };
    BigStore sv = { dst, (long long) 1 << (log2n - 1), 0, 1, 1, 0 };
    return big_forward(log2n, ld, sv, batch, tw, w, st);
}

hipError_t big_rifft_overlap_add(int log2n, const float2 *Y, long long h_first, int T, int nout, float *timeline, long long tl_stride, long long tl_mask,
                                 const float2 *tw, const BigFFTWork &w, hipStream_t st)
{
    if (!w.a || !w.b) return hipErrorInvalidValue;
    const int M = 1 << (log2n - 1), batch = T * nout;
    const int chunk = big_chunk(w, M);
    for (int q0 = 0; q0 < batch; q0 += chunk)
    {
        const int nb = std::min(chunk, batch - q0);
        hipLaunchKernelGGL(big_real_pre_kernel, dim3((M / 2 + 1 + 255) / 256, nb), dim3(256), 0, st, Y, (long long) M, w.a, M, q0, tw);
        hipError_t e = big_cfft(log2n, w.a, w.b, nb, w, tw, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(big_overlap_add_kernel, dim3((M / 2 + 255) / 256, nb), dim3(256), 0, st, w.a, M, q0, nout, h_first, timeline, tl_stride, tl_mask);
    }
    return hipGetLastError();
}

hipError_t big_rifft_rows(int log2n, const float2 *src, int batch, float *dst, const float2 *tw, const BigFFTWork &w, hipStream_t st)
{
    if (!w.a || !w.b) return hipErrorInvalidValue;
    const int M = 1 << (log2n - 1);
    const int chunk = big_chunk(w, M);
    for (int q0 = 0; q0 < batch; q0 += chunk)
    {
        const int nb = std::min(chunk, batch - q0);
        hipLaunchKernelGGL(big_real_pre_kernel, dim3((M / 2 + 1 + 255) / 256, nb), dim3(256), 0, st, src, (long long) M, w.a, M, q0, tw);
        hipError_t e = big_cfft(log2n, w.a, w.b, nb, w, tw, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(big_store_rows_kernel, dim3((M + 255) / 256, nb), dim3(256), 0, st, w.a, M, q0, dst);
    }
    return hipGetLastError();
}

// (Engine::init, once per device: HIP loads a translation unit's code object at the first launch of one of its kernels — 0.3 - 0.8 ms on the
// calling thread, which for the kernels of a control section or a restart is the audio thread in mid-stream; asking for a kernel's attributes loads it now)
void preload_bigfft()
{
    hipFuncAttributes fa;
    (void) hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(big_load_kernel));
    (void) hipGetLastError();
}

} // namespace hcv
