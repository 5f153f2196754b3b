// MI355X (gfx950 / CDNA4) kernels for the partitioned-convolution hot path.
//
//   K2/K9  rfft_frames / rfft_ir      real -> packed half spectrum, LDS Stockham, x2 scale
//   K1     spectral_mac               Y[t][o] = sum_i sum_p X[i][h-p] * H[o][i][p]   (the HBM roofline kernel)
//   K3/K4  rifft_overlap_add          packed spectrum -> time, 1/(4N) scale, valid half added to the output timeline
//   K5     fir_head                   direct-form FIR head (time-domain stage)
//   K7     scatter_input / emit       ring bookkeeping on device
//
// Spectrum format (identical to the reference's vDSP-compatible packing, HISSTools_FFT_Core.h:934-988,
// but stored INTERLEAVED as float2 per bin so one dwordx4 load carries two bins):
//   bin 0      = (2*X[0], 2*X[N/2])          DC and Nyquist packed
//   bin k>0    = (2*Re X[k], 2*Im X[k])      k < N/2
//
// Everything here is written for wave64 / gfx950 only.

#include "hcv_kernels.h"
#include "hcv_fft_device.h"

#include <hip/hip_runtime.h>

#include "hcv_fft_frames_device.h"
#include "hcv_order_check.h"

#include <algorithm>
#include <cstdlib>

namespace hcv
{

// Index of the thread group (= transform) within the workgroup.  Groups of whole waves are wave-uniform: saying so
// keeps every per-transform pointer in scalar registers and the per-lane addresses 32-bit.
template <int TG> __device__ __forceinline__ int wave_uniform_group()
{
    const int g = threadIdx.x / TG;
    if (TG % 64 == 0) return __builtin_amdgcn_readfirstlane(g);
    return g;
}

// ------------------------------------------------------------------------------------------------
// K2: forward real FFT of overlap-save frames straight out of the input history ring.
//   transform q = (t, i): frame = hist[i][(h-1)*H .. (h+1)*H), h = h_first + t   (natural [older | newer] order:
//   the valid half of the matching inverse is the SECOND one).
//   out: X[(i * R + (h mod R)) * M + k]
// ------------------------------------------------------------------------------------------------

struct FrameLoad
{
    static constexpr bool is_lds = false;
    const float *row;
    long long base, mask;
    bool live;
    __device__ __forceinline__ float2 operator()(int k) const
    {
        if (!live) return make_float2(0.f, 0.f);
        return *reinterpret_cast<const float2 *>(row + ((base + 2LL * k) & mask));
    }
};

template <int LOG2M>
__global__ __launch_bounds__(FFTGeom<LOG2M>::THREADS, 4) void rfft_frames_direct_kernel(float *__restrict__ hist, long long hist_stride, long long hist_mask,
                                                                 const float *__restrict__ in, long long in_stride, long long n0,
                                                                 long long h_first, int T, int nin, float2 *__restrict__ X, int R,
                                                                 const float2 *__restrict__ tw)
{
    using Gm = FFTGeom<LOG2M>;
    constexpr int M = Gm::M, TG = Gm::TG, G = Gm::G;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];

    const int g = wave_uniform_group<TG>(), tid = threadIdx.x % TG;
    const int q = blockIdx.x * G + g;
    const bool live = q < T * nin;
    const int t = live ? q / nin : 0, i = live ? q % nin : 0;
    const long long h = h_first + t;
    LdsBuf<float2> s = { lds + g * lds_padded(M) };

    const DirectFrameLoad ld = { hist + (long long) i * hist_stride, in + (long long) i * in_stride, (h - 1) * (long long) M, hist_mask, n0, M / 2, live };
    LdsFFT<LOG2M, TG>::run(ld, LdsIO<float2>{ s }, s, tid, tw);
    if (live)
    {
        int slot = (int) (h % R);
        real_post_store<LOG2M, TG>(s, tid, tw, X + ((long long) i * R + slot) * M);
    }
}

template <int LOG2M>
__global__ __launch_bounds__(FFTGeom<LOG2M>::THREADS, 4) void rfft_frames_kernel(const float *__restrict__ hist, long long hist_stride, long long hist_mask,
                                                          long long h_first, int T, int nin, float2 *__restrict__ X, int R,
                                                          const float2 *__restrict__ tw)
{
    using Gm = FFTGeom<LOG2M>;
    constexpr int M = Gm::M, TG = Gm::TG, G = Gm::G;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];

    const int g = wave_uniform_group<TG>(), tid = threadIdx.x % TG;
    const int q = blockIdx.x * G + g;
    const bool live = q < T * nin;
    const int t = live ? q / nin : 0, i = live ? q % nin : 0;
    const long long h = h_first + t;
    LdsBuf<float2> s = { lds + g * lds_padded(M) };

    // hop = M samples, frame = 2M samples = M float2; positions are even, so a float2 never straddles the ring's wrap
    const FrameLoad ld = { hist + (long long) i * hist_stride, (h - 1) * (long long) M, hist_mask, live };
    LdsFFT<LOG2M, TG>::run(ld, LdsIO<float2>{ s }, s, tid, tw);
    if (live)
    {
        int slot = (int) (h % R);
        real_post_store<LOG2M, TG>(s, tid, tw, X + ((long long) i * R + slot) * M);
    }
}

// ------------------------------------------------------------------------------------------------
// K9: IR partition FFT.  Partition p of one (in,out) pair: ir[p*M .. p*M + M) zero-padded to 2M samples.
//   src points at the first sample of the stage's IR segment, `count` samples are valid.
//   dst = H spectra of that pair, [P][M] float2.  Partitions past the IR are written as zeros.
// ------------------------------------------------------------------------------------------------

// (x[2k], x[2k+1]) of a row with `valid` samples (<= 0: none), zero beyond
struct SampleLoad
{
    static constexpr bool is_lds = false;
    const float *row;
    long long valid;
    __device__ __forceinline__ float2 operator()(int k) const
    {
        const long long a = 2LL * k;
        return make_float2(a < valid ? row[a] : 0.f, a + 1 < valid ? row[a + 1] : 0.f);
    }
};

template <int LOG2M>
__global__ __launch_bounds__(FFTGeom<LOG2M>::THREADS, 4) void rfft_ir_kernel(const float *__restrict__ src, long long count, int P, float2 *__restrict__ dst,
                                                      const float2 *__restrict__ tw)
{
    using Gm = FFTGeom<LOG2M>;
    constexpr int M = Gm::M, TG = Gm::TG, G = Gm::G;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];

    const int g = wave_uniform_group<TG>(), tid = threadIdx.x % TG;
    const int p = blockIdx.x * G + g;
    const bool live = p < P;
    LdsBuf<float2> s = { lds + g * lds_padded(M) };

    // samples 2k, 2k+1 of the zero-padded partition; only the first M samples of the 2M can be non-zero
    const SampleLoad ld = { src + (long long) p * M, live ? std::min<long long>(M, count - (long long) p * M) : 0 };
    LdsFFT<LOG2M, TG>::run(ld, LdsIO<float2>{ s }, s, tid, tw);
    if (live) real_post_store<LOG2M, TG>(s, tid, tw, dst + (long long) p * M);
}

// Generic real FFT of `batch` independent rows (plumbing for the hisstools_rfft surface): row b has
// in_len valid samples at src + b * src_stride, zero padded to 2M; dst row stride M float2.
template <int LOG2M>
__global__ __launch_bounds__(FFTGeom<LOG2M>::THREADS, 4) void rfft_rows_kernel(const float *__restrict__ src, long long src_stride, long long in_len, int batch,
                                                        float2 *__restrict__ dst, const float2 *__restrict__ tw)
{
    using Gm = FFTGeom<LOG2M>;
    constexpr int M = Gm::M, TG = Gm::TG, G = Gm::G;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];

    const int g = wave_uniform_group<TG>(), tid = threadIdx.x % TG;
    const int b = blockIdx.x * G + g;
    const bool live = b < batch;
    LdsBuf<float2> s = { lds + g * lds_padded(M) };
    const SampleLoad ld = { src + (long long) (live ? b : 0) * src_stride, live ? std::min<long long>(in_len, 2LL * M) : 0 };
    LdsFFT<LOG2M, TG>::run(ld, LdsIO<float2>{ s }, s, tid, tw);
    if (live) real_post_store<LOG2M, TG>(s, tid, tw, dst + (long long) b * M);
}

// ------------------------------------------------------------------------------------------------
// Inverse real FFT core: packed spectrum in LDS (s[0..M)) -> 2M real samples left in LDS as
// s[k] = (x[2k+1], x[2k])  (note the swap: the inverse complex FFT is run on exchanged re/im, the
// trick of HISSTools_FFT_Core.h:1341-1346).  Unnormalised, as hisstools_rifft.
// ------------------------------------------------------------------------------------------------

// First-pass source of the inverse transforms: element n of the exchanged-re/im complex input, computed on the fly from
// the packed spectrum held in LDS (pass_real_trig_table<true>, HISSTools_FFT_Core.h:934-988).
template <int LOG2M>
struct PreLoad
{
    static constexpr bool is_lds = true;
    LdsBuf<float2> s;
    const float2 *__restrict__ tw;
    __device__ __forceinline__ float2 operator()(int n) const
    {
        constexpr int M = 1 << LOG2M;
        if (n == 0)
        {
            const float2 z = s[0];
            return make_float2(z.x - z.y, z.x + z.y);         // (im, re): re = x + y, im = x - y
        }
        const bool lo = n <= M / 2;
        const int k = lo ? n : M - n, m = M - k;
        const float2 w = tw[k];
        const float c = -w.x, sn = w.y;
        const float2 z1 = s[k], z2 = s[m];
        const float r3 = z1.x + z2.x, i3 = z1.y + z2.y, r4 = z1.x - z2.x, i4 = z1.y - z2.y;
        const float u1 = (c * i3) + (sn * r4);
        const float u2 = (sn * i3) - (c * r4);
        return lo ? make_float2(u2 + i4, r3 + u1) : make_float2(u2 - i4, r3 - u1);
    }
};

// Packed spectrum (the sum of `ksplit` partials) of one transform into LDS: every thread first issues all of its loads,
// then adds and stores — under a saturated HBM a load costs microseconds, so they must not queue behind one another.
template <int LOG2M, int TG, bool AGENT = false>
__device__ __forceinline__ void stage_spectrum(LdsBuf<float2> s, int tid, const float2 *__restrict__ src, int ksplit, long long ks_stride, bool live)
{
    constexpr int M = 1 << LOG2M, EPT = (M + TG - 1) / TG;
    float2 v[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++)
    {
        const int k = tid + e * TG;
        v[e] = (live && (M % TG == 0 || k < M)) ? get2<AGENT>(src + k) : make_float2(0.f, 0.f);
    }
    for (int ks = 1; ks < ksplit; ks++)
    {
        float2 b[EPT];
#pragma unroll
        for (int e = 0; e < EPT; e++)
        {
            const int k = tid + e * TG;
            b[e] = (live && (M % TG == 0 || k < M)) ? get2<AGENT>(src + ks * ks_stride + k) : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int e = 0; e < EPT; e++)
        {
            v[e].x += b[e].x;
            v[e].y += b[e].y;
        }
    }
#pragma unroll
    for (int e = 0; e < EPT; e++)
    {
        const int k = tid + e * TG;
        if (M % TG == 0 || k < M) s[k] = v[e];
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// K3 + K4 + K6: inverse FFT of the accumulated spectra, scale by 1/(4N), and ADD the valid half-frame into
// the per-output timeline ring at the hop's emission time ((h+1)*H .. (h+2)*H) — latency N/2, as the reference).
//   Y: [ksplit][T][nout][M] float2 partial sums (summed here)
// ------------------------------------------------------------------------------------------------

// last-pass sink: result k = (x[2k+1], x[2k]); the second half of the frame is added to the timeline ring
struct OverlapAddStore
{
    static constexpr bool is_lds = false;
    float *row;
    long long base, mask;
    float scale;
    int first;
    bool live;
    __device__ __forceinline__ void operator()(int k, float2 v) const
    {
        if (!live || k < first) return;
        // The timeline has one writer per sample at a time, so the add needs no atomicity — but a plain load / add / store
        // would chain the eight updates of a thread (the compiler must assume they alias), each a round trip to an HBM
        // that the tail MAC keeps saturated.  No-return hardware float atomics are fire-and-forget.
        float *d = row + ((base + 2LL * k) & mask);
        unsafeAtomicAdd(d, v.y * scale);
        unsafeAtomicAdd(d + 1, v.x * scale);
    }
};

template <int LOG2M>
__global__ __launch_bounds__(FFTGeom<LOG2M>::THREADS, 4) void rifft_overlap_add_kernel(const float2 *__restrict__ Y, int ksplit, long long ks_stride, long long h_first,
                                                                int T, int nout, float *__restrict__ timeline, long long tl_stride,
                                                                long long tl_mask, const float2 *__restrict__ tw)
{
    using Gm = FFTGeom<LOG2M>;
    constexpr int M = Gm::M, TG = Gm::TG, G = Gm::G;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];

    const int g = wave_uniform_group<TG>(), tid = threadIdx.x % TG;
    const int q = blockIdx.x * G + g;
    const bool live = q < T * nout;
    const int t = live ? q / nout : 0, o = live ? q % nout : 0;
    LdsBuf<float2> s = { lds + g * lds_padded(M) };

    stage_spectrum<LOG2M, TG>(s, tid, Y + ((long long) t * nout + o) * M, ksplit, ks_stride, live);
    // valid samples are e in [M, 2M): float2 index k in [M/2, M); scale 1 / (4N), N = 2M  (scaleStore, PartitionedConvolve.cpp:232-241)
    const OverlapAddStore st = { timeline + (long long) o * tl_stride, (h_first + t + 1) * (long long) M - M, tl_mask, 1.f / (float) (8 * M), M / 2, live };
    LdsFFT<LOG2M, TG>::run(PreLoad<LOG2M>{ s, tw }, st, s, tid, tw);
}

// The same inverse delivering its valid half straight into the caller's output block (scaleStore + the framing copy of
// PartitionedConvolve.cpp:232-241,359-360 in the transform's last pass): for blocks whose samples come from ONE zero-latency
// transform per output and hop — whole-hop mode — so that neither a timeline ring nor an emit launch stands between the
// inverse and the caller.
struct EmitStore
{
    static constexpr bool is_lds = false;
    float *row;             // out[o] + t * M - M: sample e of the frame lands at row[e]
    float scale;
    int first;
    bool live;
    __device__ __forceinline__ void operator()(int k, float2 v) const
    {
        if (!live || k < first) return;
        *reinterpret_cast<float2 *>(row + 2LL * k) = make_float2(v.y * scale, v.x * scale);
    }
};

template <int LOG2M>
__global__ __launch_bounds__(FFTGeom<LOG2M>::THREADS, 4) void rifft_emit_kernel(const float2 *__restrict__ Y, int ksplit, long long ks_stride, int T, int nout,
                                                         float *__restrict__ out, long long out_stride, const float2 *__restrict__ tw)
{
    using Gm = FFTGeom<LOG2M>;
    constexpr int M = Gm::M, TG = Gm::TG, G = Gm::G;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];

    const int g = wave_uniform_group<TG>(), tid = threadIdx.x % TG;
    const int q = blockIdx.x * G + g;
    const bool live = q < T * nout;
    const int t = live ? q / nout : 0, o = live ? q % nout : 0;
    LdsBuf<float2> s = { lds + g * lds_padded(M) };

    stage_spectrum<LOG2M, TG>(s, tid, Y + ((long long) t * nout + o) * M, ksplit, ks_stride, live);
    const EmitStore st = { out + (long long) o * out_stride + (long long) t * M - M, 1.f / (float) (8 * M), M / 2, live };
    LdsFFT<LOG2M, TG>::run(PreLoad<LOG2M>{ s, tw }, st, s, tid, tw);
}

struct SwapStore
{
    static constexpr bool is_lds = false;
    float2 *d;
    bool live;
    __device__ __forceinline__ void operator()(int k, float2 v) const
    {
        if (live) d[k] = make_float2(v.y, v.x);
    }
};

// plain inverse for the hisstools_rifft plumbing surface: src rows of M float2 -> dst rows of 2M floats
template <int LOG2M>
__global__ __launch_bounds__(FFTGeom<LOG2M>::THREADS, 4) void rifft_rows_kernel(const float2 *__restrict__ src, int batch, float *__restrict__ dst,
                                                         const float2 *__restrict__ tw)
{
    using Gm = FFTGeom<LOG2M>;
    constexpr int M = Gm::M, TG = Gm::TG, G = Gm::G;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int g = wave_uniform_group<TG>(), tid = threadIdx.x % TG;
    const int b = blockIdx.x * G + g;
    const bool live = b < batch;
    LdsBuf<float2> s = { lds + g * lds_padded(M) };
    stage_spectrum<LOG2M, TG>(s, tid, src + (long long) (live ? b : 0) * M, 1, 0, live);
    const SwapStore st = { reinterpret_cast<float2 *>(dst + (long long) (live ? b : 0) * 2 * M), live };
    LdsFFT<LOG2M, TG>::run(PreLoad<LOG2M>{ s, tw }, st, s, tid, tw);
}

// Split-K epilogue: Y[0][e] = sum_ks Y[ks][e].  One float4 per thread, the ksplit strided loads of a thread are
// independent (issued back to back), neighbouring threads are contiguous: a plain coalesced streaming reduction.
__global__ __launch_bounds__(256) void reduce_partials_kernel(float4 *Y, int ksplit, long long ks_stride4, long long n4, int pin, float4 *dst)
{
    int bx = blockIdx.x;
    if (pin >= 0)
    {
        if ((bx & 7) != pin) return;
        bx >>= 3;
    }
    const long long e = bx * (long long) blockDim.x + threadIdx.x;
    if (e >= n4) return;
    if (ksplit <= 33)
    {
        // few slices (the small engines' launch-bound chains): every slice's load in flight at once — the slices were written a
        // moment ago by workgroups all over the chip and come from the memory side, two microseconds a trip — then the same sums
        // in the same order as the loop below
        float4 v[33];
#pragma unroll
        for (int k = 0; k < 33; k++)
            if (k < ksplit) v[k] = Y[(long long) k * ks_stride4 + e];
        float4 s = v[0];
        const int quads = (ksplit - 1) / 4;
#pragma unroll
        for (int g = 0; g < 8; g++)
            if (g < quads)
            {
                const float4 a = v[1 + 4 * g], b = v[2 + 4 * g], c = v[3 + 4 * g], d = v[4 + 4 * g];
                s.x += (a.x + b.x) + (c.x + d.x);
                s.y += (a.y + b.y) + (c.y + d.y);
                s.z += (a.z + b.z) + (c.z + d.z);
                s.w += (a.w + b.w) + (c.w + d.w);
            }
#pragma unroll
        for (int k = 1; k < 33; k++)
            if (k > 4 * quads && k < ksplit)
            {
                s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w;
            }
        dst[e] = s;
        return;
    }
    float4 s = Y[e];
    int ks = 1;
    for (; ks + 3 < ksplit; ks += 4)
    {
        const float4 a = Y[(long long) ks * ks_stride4 + e];
        const float4 b = Y[(long long) (ks + 1) * ks_stride4 + e];
        const float4 c = Y[(long long) (ks + 2) * ks_stride4 + e];
        const float4 d = Y[(long long) (ks + 3) * ks_stride4 + e];
        s.x += (a.x + b.x) + (c.x + d.x);
        s.y += (a.y + b.y) + (c.y + d.y);
        s.z += (a.z + b.z) + (c.z + d.z);
        s.w += (a.w + b.w) + (c.w + d.w);
    }
    for (; ks < ksplit; ks++)
    {
        const float4 a = Y[(long long) ks * ks_stride4 + e];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    dst[e] = s;
}

// ------------------------------------------------------------------------------------------------
// K5: direct-form FIR head.  out[o][n] = sum_i sum_k taps[o][i][k] * x[i][n - k],  k < L (L padded to x4, <= 2048).
// Block = 128 threads = 512 output samples x OTD outputs; the input window and the taps are staged in LDS;
// each thread keeps a 4-sample x OTD-output register tile and walks the taps four at a time (aligned float4 LDS reads).
// OTD (1, 2 or 4) is chosen at launch so that small matrices still spread over the chip.
// ------------------------------------------------------------------------------------------------

constexpr int FIR_THREADS = 128;
constexpr int FIR_SPB = 4 * FIR_THREADS;      // samples per block

template <int FIR_OTD, bool CHECK>
__global__ __launch_bounds__(FIR_THREADS) void fir_head_kernel(const float *__restrict__ hist, long long hist_stride, long long hist_mask,
                                                       const float *__restrict__ taps, int Lpad, int tap_stride, int nin, int nin_alloc,
                                                       int nout, int diag, long long n0, int B, const long long *__restrict__ valid_from,
                                                       float *__restrict__ out, long long out_stride)
{
    extern __shared__ __attribute__((aligned(16))) float fir_lds[];
    float *xs = fir_lds;                                   // Lpad + FIR_SPB floats: x[n_blk - Lpad .. n_blk + SPB)
    float *hs = fir_lds + (Lpad + FIR_SPB);                // FIR_OTD * Lpad taps

    const int tid = threadIdx.x;
    const int nb = blockIdx.x * FIR_SPB;                   // first sample of this block inside the call
    const int o0 = blockIdx.y * FIR_OTD;
    const long long nabs = n0 + nb;                        // absolute sample index of the block start

    float acc[FIR_OTD][4];
#pragma unroll
    for (int j = 0; j < FIR_OTD; j++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[j][q] = 0.f;

    const int ni = diag ? 1 : nin;
    for (int ii = 0; ii < ni; ii++)
    {
        __syncthreads();
        if (!diag)
        {
            const float *row = hist + (long long) ii * hist_stride;
            for (int e = tid; e < Lpad + FIR_SPB; e += FIR_THREADS)
                xs[e] = row[(nabs - Lpad + e) & hist_mask];
        }
        for (int e = tid; e < FIR_OTD * Lpad; e += FIR_THREADS)
        {
            int j = e / Lpad, k = e - j * Lpad;
            int o = min(o0 + j, nout - 1);
            int i = diag ? 0 : ii;
            hs[e] = taps[((long long) o * nin_alloc + i) * tap_stride + k];
        }
        __syncthreads();

#pragma unroll
        for (int j = 0; j < FIR_OTD; j++)
        {
            int o = min(o0 + j, nout - 1);
            if (diag)
            {
                // each output has its own input row: restage x for this output
                __syncthreads();
                const float *row = hist + (long long) o * hist_stride;
                for (int e = tid; e < Lpad + FIR_SPB; e += FIR_THREADS)
                    xs[e] = row[(nabs - Lpad + e) & hist_mask];
                __syncthreads();
            }
            long long vf = 0;
            if (CHECK) vf = valid_from[(long long) o * nin_alloc + (diag ? 0 : ii)];
            // thread's four samples: n = nb + 4*tid + q ; x[n - k] lives at xs[Lpad + 4*tid + q - k]
            const float *hj = hs + j * Lpad;
            const int c = Lpad + 4 * tid;
            float4 hi4 = *reinterpret_cast<const float4 *>(xs + c);       // x[n0..n0+3] for k = 0..3 window top
            for (int k = 0; k < Lpad; k += 4)
            {
                float4 lo4 = *reinterpret_cast<const float4 *>(xs + c - k - 4);
                float4 h4 = *reinterpret_cast<const float4 *>(hj + k);
                float w[8] = { lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w };   // x at offsets -4..+3 relative to (c - k)
                if (CHECK)
                {
#pragma unroll
                    for (int e = 0; e < 8; e++)
                        if (nabs + 4 * tid - k - 4 + e < vf) w[e] = 0.f;
                }
                const float hk[4] = { h4.x, h4.y, h4.z, h4.w };
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int kk = 0; kk < 4; kk++)
                        acc[j][q] += hk[kk] * w[4 + q - kk];
                hi4 = lo4;
            }
        }
    }

#pragma unroll
    for (int j = 0; j < FIR_OTD; j++)
    {
        int o = o0 + j;
        if (o < nout)
        {
#pragma unroll
            for (int q = 0; q < 4; q++)
            {
                int n = nb + 4 * tid + q;
                if (n < B) out[(long long) o * out_stride + n] = acc[j][q];
            }
        }
    }
}

// The same sum for SMALL calls of a matrix with many inputs (the real-time sizes: 32 .. 256 samples per call, 64 x 64 channels).
// The kernel above gives a workgroup 512 output samples of one output and walks the inputs one after the other — two barriers and
// L taps of serial work per input: with 32 samples per call 8 of its 128 threads have anything to do and a 64-input row takes
// 222 us, three quarters of a 32-sample call of the 64 x 64 engine.  Here a workgroup owns 32 samples of one output and its 512
// threads split the TAPS sixteen ways (thread = sample n, tap slice ks); the windows and taps of a whole batch of inputs — all of them
// while they fit 96 KiB — are staged in LDS at once (every global load in flight before the one barrier), each thread runs
// nin x L / 16 multiply-adds out of LDS (x reads conflict-free, tap reads broadcast), and the slices are added up in LDS.
// TimeDomainConvolve.cpp:100-163.
constexpr int FIRS_THREADS = 512, FIRS_SAMPLES = 32, FIRS_SLICES = FIRS_THREADS / FIRS_SAMPLES;

// EMIT: the call completes no hop of any stage (three calls in four at 32 samples per call) — the head's samples are then the last
// thing the call computes, and this kernel delivers the block itself: out = head + what the stages' timelines hold for these samples
// (cleared as they are read, as emit_kernel does).  One launch and one cross-stream hand-over less in the chain of a plain real-time call.
template <bool CHECK, bool EMIT>
__global__ __launch_bounds__(FIRS_THREADS) void fir_head_small_kernel(const float *__restrict__ hist, long long hist_stride, long long hist_mask,
                                                                      const float *__restrict__ taps, int Lpad, int tap_stride, int nin, int nin_alloc,
                                                                      long long n0, int B, const long long *__restrict__ valid_from,
                                                                      float *__restrict__ out, long long out_stride, int ib,
                                                                      const float *__restrict__ din, long long in_stride, EmitSources src,
                                                                      float *__restrict__ ring)
{
    extern __shared__ __attribute__((aligned(16))) float firs_lds[];
    // `ring`: the kernel also FILES the call's samples in the history ring (scatter_input's work: the first output's workgroups, each its
    // own span of samples) — a plain real-time call is then this one launch.  Nobody reads the ring at these positions here: every
    // window takes the call's own samples from the caller's block.
    if (EMIT && ring && blockIdx.y == 0)
    {
        const int nb0 = blockIdx.x * FIRS_SAMPLES;
        for (int e = threadIdx.x; e < nin * FIRS_SAMPLES; e += FIRS_THREADS)
        {
            const int i = e / FIRS_SAMPLES, j = nb0 + (e & (FIRS_SAMPLES - 1));
            if (j < B) ring[(long long) i * hist_stride + ((n0 + j) & hist_mask)] = din[(long long) i * in_stride + j];
        }
    }
    const int W = FIRS_SAMPLES + Lpad;                     // x[nblk - Lpad .. nblk + 32) of one input
    float *xs = firs_lds;                                  // [ib][W]
    float *hs = firs_lds + (size_t) ib * W;                // [ib][Lpad]
    float *red = hs + (size_t) ib * Lpad;                  // [slices][32]

    const int tid = threadIdx.x, n = tid & (FIRS_SAMPLES - 1), ks = tid / FIRS_SAMPLES;
    const int nb = blockIdx.x * FIRS_SAMPLES, o = blockIdx.y;
    const long long nabs = n0 + nb;
    const int kper = Lpad / FIRS_SLICES, k0 = ks * kper;   // (Lpad is a multiple of 16)
    const int L4 = Lpad / 4;

    float acc = 0.f;
    for (int i0 = 0; i0 < nin; i0 += ib)
    {
        const int nbi = min(ib, nin - i0);
        __syncthreads();
        // the windows, four samples per load where the window starts on a multiple of four (every call of a host with a fixed block
        // size), and a thread's loads of a round all in flight before the first of them is stored: one load after the other — 20 round
        // trips to the L2 per thread — was most of the kernel's 22 us
        const bool quads = ((nabs - Lpad) & 3) == 0 && (hist_stride & 3) == 0;
        if (quads)
        {
            const int W4 = W / 4, total = nbi * W4;
            for (int base = 0; base < total; base += 4 * FIRS_THREADS)
            {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; u++)
                {
                    const int e = base + u * FIRS_THREADS + tid;
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (e < total)
                    {
                        const int i = e / W4, j4 = e - i * W4;
                        const long long pos = nabs - Lpad + 4 * j4;
                        if (din && pos >= n0)
                        {
                            // the call's own samples straight from the caller's block (rows of any alignment)
                            const float *src = din + (long long) (i0 + i) * in_stride + (pos - n0);
                            const long long left = (long long) B - (pos - n0);
                            v[u].x = left > 0 ? src[0] : 0.f;
                            v[u].y = left > 1 ? src[1] : 0.f;
                            v[u].z = left > 2 ? src[2] : 0.f;
                            v[u].w = left > 3 ? src[3] : 0.f;
                        }
                        else if (din && pos + 3 >= n0)
                        {
                            // (a group straddling the start of the call: n0 is not a multiple of four)
                            float t[4];
#pragma unroll
                            for (int c = 0; c < 4; c++)
                            {
                                const long long q = pos + c;
                                t[c] = q >= n0 ? (q - n0 < B ? din[(long long) (i0 + i) * in_stride + (q - n0)] : 0.f)
                                               : hist[(long long) (i0 + i) * hist_stride + (q & hist_mask)];
                            }
                            v[u] = make_float4(t[0], t[1], t[2], t[3]);
                        }
                        else
                            v[u] = *reinterpret_cast<const float4 *>(hist + (long long) (i0 + i) * hist_stride + (pos & hist_mask));
                        if (CHECK)
                        {
                            const long long vf = valid_from[(long long) o * nin_alloc + (i0 + i)];
                            if (pos < vf) v[u].x = 0.f;
                            if (pos + 1 < vf) v[u].y = 0.f;
                            if (pos + 2 < vf) v[u].z = 0.f;
                            if (pos + 3 < vf) v[u].w = 0.f;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                {
                    const int e = base + u * FIRS_THREADS + tid;
                    if (e < total) reinterpret_cast<float4 *>(xs)[e] = v[u];
                }
            }
        }
        else
            for (int e = tid; e < nbi * W; e += FIRS_THREADS)
            {
                const int i = e / W, j = e - i * W;
                const long long pos = nabs - Lpad + j;
                // the call's own samples straight from the caller's block (din != nullptr): the kernel then does not wait for the scatter
                // that files them in the ring, only for the ring's older contents
                float v;
                if (din && pos >= n0) v = pos - n0 < B ? din[(long long) (i0 + i) * in_stride + (pos - n0)] : 0.f;
                else v = hist[(long long) (i0 + i) * hist_stride + (pos & hist_mask)];
                if (CHECK)
                {
                    if (pos < valid_from[(long long) o * nin_alloc + (i0 + i)]) v = 0.f;
                }
                xs[e] = v;
            }
        for (int base = 0; base < nbi * L4; base += 4 * FIRS_THREADS)
        {
            float4 h4[4];
#pragma unroll
            for (int u = 0; u < 4; u++)
            {
                const int e = base + u * FIRS_THREADS + tid;
                if (e < nbi * L4)
                {
                    const int i = e / L4, k4 = e - i * L4;
                    h4[u] = *reinterpret_cast<const float4 *>(taps + ((long long) o * nin_alloc + (i0 + i)) * tap_stride + 4 * k4);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
            {
                const int e = base + u * FIRS_THREADS + tid;
                if (e < nbi * L4) reinterpret_cast<float4 *>(hs)[e] = h4[u];
            }
        }
        __syncthreads();
        for (int i = 0; i < nbi; i++)
        {
            const float *x = xs + i * W + Lpad + n - k0;   // x[n - k] = x[-(k - k0)]
            const float *h = hs + i * Lpad + k0;
#pragma unroll 8
            for (int k = 0; k < kper; k++) acc += h[k] * x[-k];
        }
    }
    red[ks * FIRS_SAMPLES + n] = acc;
    __syncthreads();
    if (ks == 0 && nb + n < B)
    {
        float s = acc;
#pragma unroll
        for (int q = 1; q < FIRS_SLICES; q++) s += red[q * FIRS_SAMPLES + n];
        if constexpr (EMIT)
        {
#pragma unroll
            for (int k = 0; k < kMaxStages; k++)
                if (k < src.count)
                {
                    float *p = src.timeline[k] + (long long) o * src.stride[k] + ((n0 + nb + n) & src.mask[k]);
                    s += *p;
                    *p = 0.f;
                }
        }
        out[(long long) o * out_stride + nb + n] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// K7: ring bookkeeping
// ------------------------------------------------------------------------------------------------

// hist[i][(n0 + j) & mask] = in[i][j]
__global__ void scatter_input_kernel(const float *__restrict__ in, long long in_stride, int B, float *__restrict__ hist,
                                     long long hist_stride, long long hist_mask, long long n0)
{
    int i = blockIdx.y;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < B; j += gridDim.x * blockDim.x)
        hist[(long long) i * hist_stride + ((n0 + j) & hist_mask)] = in[(long long) i * in_stride + j];
}

// out[o][j] = sum_s timeline_s[o][(n0 + j) & mask_s] + td[o][j]; the consumed timeline spans are zeroed for reuse.
// Every FFT stage owns a timeline ring (so the stages can run concurrently on their own streams).
__global__ void emit_kernel(EmitSources src, long long n0, int B, const float *__restrict__ td, long long td_stride, float *__restrict__ out,
                            long long out_stride)
{
    int o = blockIdx.y;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < B; j += gridDim.x * blockDim.x)
    {
        float v = 0.f;
#pragma unroll
        for (int s = 0; s < kMaxStages; s++)
        {
            if (s < src.count)
            {
                float *p = src.timeline[s] + (long long) o * src.stride[s] + ((n0 + j) & src.mask[s]);
                v += *p;
                *p = 0.f;
            }
        }
        if (td) v += td[(long long) o * td_stride + j];
        out[(long long) o * out_stride + j] = v;
    }
}

// the final per-output sum over the input groups of a sharded matrix (NToMonoConvolve.cpp:39-42 across devices):
// out[o][j] = sum_c part_c[o][j]; the partial blocks may live on peer devices (read over xGMI)
__global__ void sum_parts_kernel(PartSources src, long long part_stride, int B, float *__restrict__ out, long long out_stride)
{
    const int o = blockIdx.y;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < B; j += gridDim.x * blockDim.x)
    {
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < kMaxParts; c++)
            if (c < src.count) v += src.part[c][(long long) o * part_stride + j];
        out[(long long) o * out_stride + j] = v;
    }
}

__global__ void fill_i64_kernel(long long *p, long long n, long long v)
{
    for (long long j = blockIdx.x * (long long) blockDim.x + threadIdx.x; j < n; j += (long long) gridDim.x * blockDim.x) p[j] = v;
}

// copy spectra [pairs][Pold][M] -> [pairs][Pnew][M] (capacity growth of a stage), float4 granularity
__global__ void regrow_spectra_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, long long pairs, int Pold, int Pnew, int M2)
{
    long long per = (long long) Pold * M2;
    long long total = pairs * per;
    for (long long e = blockIdx.x * (long long) blockDim.x + threadIdx.x; e < total; e += (long long) gridDim.x * blockDim.x)
    {
        long long pair = e / per, r = e - pair * per;
        dst[pair * (long long) Pnew * M2 + r] = src[e];
    }
}

// move the live hops of an input-spectrum ring to a ring with a different slot count
__global__ void regrow_ring_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, int nin, int Rold, int Rnew, int M2,
                                   long long h_last, int live)
{
    // blockIdx.y = input, blockIdx.z = age (0 = newest hop h_last)
    int i = blockIdx.y, age = blockIdx.z;
    if (age >= live) return;
    long long h = h_last - age;
    if (h < 0) return;
    const float4 *s = src + ((long long) i * Rold + (h % Rold)) * M2;
    float4 *d = dst + ((long long) i * Rnew + (h % Rnew)) * M2;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < M2; e += gridDim.x * blockDim.x) d[e] = s[e];
}

// ------------------------------------------------------------------------------------------------
// One-shot spectral convolution / correlation (spectral_processor::convolve/correlate, real overloads —
// SpectralProcessor.hpp:617-674): bin-wise product of two packed half spectra and the output arrangement.
// ------------------------------------------------------------------------------------------------

// a = scale * (a op b): op = complex product (convolve) or product with the conjugate of b (correlate); bin 0 holds
// (DC, Nyquist) and takes two real products (real_operation, SpectralFunctions.hpp:63-83)
__global__ void spectral_pointwise_kernel(float2 *__restrict__ a, const float2 *__restrict__ b, int M, float scale, int correlate)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    const float2 x = a[k], y = b[k];
    if (k == 0)
        a[0] = make_float2(scale * (x.x * y.x), scale * (x.y * y.y));
    else if (!correlate)
        a[k] = make_float2(scale * (x.x * y.x - x.y * y.y), scale * (x.y * y.x + x.x * y.y));
    else
        a[k] = make_float2(scale * (x.x * y.x + x.y * y.y), scale * (x.y * y.x - x.x * y.y));
}

// out[o_off + i] (=, +=) t[off + i]  or  out[o_off + i] = 0 : the copy / wrap / zero steps of arrange_convolve and
// arrange_correlate (SpectralProcessor.hpp:448-545); op 0 copy, 1 add, 2 zero
__global__ void segment_op_kernel(float *__restrict__ out, const float *__restrict__ t, long long o_off, long long off, long long n, int op)
{
    for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x)
    {
        if (op == 0) out[o_off + i] = t[off + i];
        else if (op == 1) out[o_off + i] += t[off + i];
        else out[o_off + i] = 0.f;
    }
}

// copy_fold of the fold edge modes (SpectralProcessor.hpp:361-377): dst = [mirrored head | in | mirrored tail], the mirrors
// `fold` samples long, including the edge sample itself when off == 0 (FoldRepeat) and excluding it when off == 1 (Fold)
__global__ void fold_copy_kernel(float *__restrict__ dst, const float *__restrict__ in, long long n, long long fold, int off)
{
    const long long total = n + 2 * fold;
    for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < total; i += (long long) gridDim.x * blockDim.x)
    {
        long long src;
        if (i < fold) src = off + fold - 1 - i;
        else if (i < fold + n) src = i - fold;
        else src = n - off - 1 - (i - fold - n);
        dst[i] = in[src];
    }
}

// ================================================================================================
// launchers
// ================================================================================================

#define HCV_FFT_DISPATCH(LOG2M_EXPR, CALL)                                                                             \
    switch (LOG2M_EXPR)                                                                                                \
    {                                                                                                                  \
        case 4: { constexpr int L = 4; CALL; } break;                                                                  \
        case 5: { constexpr int L = 5; CALL; } break;                                                                  \
        case 6: { constexpr int L = 6; CALL; } break;                                                                  \
        case 7: { constexpr int L = 7; CALL; } break;                                                                  \
        case 8: { constexpr int L = 8; CALL; } break;                                                                  \
        case 9: { constexpr int L = 9; CALL; } break;                                                                  \
        case 10: { constexpr int L = 10; CALL; } break;                                                                \
        case 11: { constexpr int L = 11; CALL; } break;                                                                \
        case 12: { constexpr int L = 12; CALL; } break;                                                                \
        case 13: { constexpr int L = 13; CALL; } break;                                                                \
        case 14: { constexpr int L = 14; CALL; } break;                                                                \
        default: return hipErrorInvalidValue;                                                                          \
    }

static thread_local bool tlsPinHint = false;
static thread_local int tlsPinXcd = 0;
void xcd_pin_hint(bool on, int xcd)
{
    tlsPinHint = on;
    tlsPinXcd = xcd & 7;
}

int xcd_pin_for(long long workgroups)
{
    // the launches of blocks the engine marked (xcd_pin_hint), up to one XCD's worth of workgroups
    if (!tlsPinHint || workgroups <= 0 || workgroups > 32) return -1;
    return tlsPinXcd;
}

template <int L> static inline size_t fft_lds_bytes() { return sizeof(float2) * lds_padded(FFTGeom<L>::M) * FFTGeom<L>::G; }

template <typename K> static hipError_t allow_lds(K kernel, size_t bytes)
{
    if (bytes > 48 * 1024) return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes);
    return hipSuccess;
}

hipError_t launch_rfft_frames(int log2n, const float *hist, long long hist_stride, long long hist_mask, long long h_first, int T, int nin,
                              float2 *X, int R, const float2 *tw, const BigFFTWork *big, hipStream_t st)
{
    if (T <= 0 || nin <= 0) return hipSuccess;
    {
        const long long M_ = 1ll << (log2n - 1);
        ORD_ACCESS(st, hist, (h_first - 1) * M_, (h_first + T) * M_, hist_mask + 1, false, "history ring (frames of a forward transform)");
        ORD_ACCESS(st, X, h_first, h_first + T, (long long) R, true, "input-spectrum ring slots (forward transform)");
    }
    if (is_big_fft(log2n)) return big ? big_rfft_frames(log2n, hist, hist_stride, hist_mask, h_first, T, nin, X, R, tw, *big, st) : hipErrorInvalidValue;
    HCV_FFT_DISPATCH(log2n - 1, {
        using Gm = FFTGeom<L>;
        size_t lds = fft_lds_bytes<L>();
        hipError_t e = allow_lds(rfft_frames_kernel<L>, lds);
        if (e != hipSuccess) return e;
        int grid = (T * nin + Gm::G - 1) / Gm::G;
        hipLaunchKernelGGL(rfft_frames_kernel<L>, dim3(grid), dim3(Gm::THREADS), lds, st, hist, hist_stride, hist_mask, h_first, T, nin, X, R, tw);
    });
    return hipGetLastError();
}

hipError_t launch_rfft_frames_direct(int log2n, float *hist, long long hist_stride, long long hist_mask, const float *in, long long in_stride, long long n0,
                                     long long h_first, int T, int nin, float2 *X, int R, const float2 *tw, hipStream_t st)
{
    if (T <= 0 || nin <= 0) return hipSuccess;
    if (is_big_fft(log2n)) return hipErrorInvalidValue;
    {
        const long long M_ = 1ll << (log2n - 1);
        ORD_ACCESS(st, hist, (h_first - 1) * M_, h_first * M_, hist_mask + 1, false, "history ring (the hop before the caller's block)");
        ORD_ACCESS(st, hist, n0, n0 + (long long) T * M_, hist_mask + 1, true, "history ring (the caller's block filed by the forward transform)");
        ORD_ACCESS(st, X, h_first, h_first + T, (long long) R, true, "input-spectrum ring slots (forward transform)");
    }
    if (fft_split_applies(log2n, T * nin)) return launch_rfft_frames_direct_split(log2n, hist, hist_stride, hist_mask, in, in_stride, n0, h_first, T, nin, X, R, tw, st);
    HCV_FFT_DISPATCH(log2n - 1, {
        using Gm = FFTGeom<L>;
        size_t lds = fft_lds_bytes<L>();
        hipError_t e = allow_lds(rfft_frames_direct_kernel<L>, lds);
        if (e != hipSuccess) return e;
        int grid = (T * nin + Gm::G - 1) / Gm::G;
        hipLaunchKernelGGL(rfft_frames_direct_kernel<L>, dim3(grid), dim3(Gm::THREADS), lds, st, hist, hist_stride, hist_mask, in, in_stride, n0, h_first, T,
                           nin, X, R, tw);
    });
    return hipGetLastError();
}

hipError_t launch_rfft_ir(int log2n, const float *src, long long count, int P, float2 *dst, const float2 *tw, const BigFFTWork *big, hipStream_t st)
{
    if (P <= 0) return hipSuccess;
    if (is_big_fft(log2n)) return big ? big_rfft_ir(log2n, src, count, P, dst, tw, *big, st) : hipErrorInvalidValue;
    HCV_FFT_DISPATCH(log2n - 1, {
        using Gm = FFTGeom<L>;
        size_t lds = fft_lds_bytes<L>();
        hipError_t e = allow_lds(rfft_ir_kernel<L>, lds);
        if (e != hipSuccess) return e;
        int grid = (P + Gm::G - 1) / Gm::G;
        hipLaunchKernelGGL(rfft_ir_kernel<L>, dim3(grid), dim3(Gm::THREADS), lds, st, src, count, P, dst, tw);
    });
    return hipGetLastError();
}

hipError_t launch_rfft_rows(int log2n, const float *src, long long src_stride, long long in_len, int batch, float2 *dst, const float2 *tw,
                            const BigFFTWork *big, hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    if (is_big_fft(log2n)) return big ? big_rfft_rows(log2n, src, src_stride, in_len, batch, dst, tw, *big, st) : hipErrorInvalidValue;
    HCV_FFT_DISPATCH(log2n - 1, {
        using Gm = FFTGeom<L>;
        size_t lds = fft_lds_bytes<L>();
        hipError_t e = allow_lds(rfft_rows_kernel<L>, lds);
        if (e != hipSuccess) return e;
        int grid = (batch + Gm::G - 1) / Gm::G;
        hipLaunchKernelGGL(rfft_rows_kernel<L>, dim3(grid), dim3(Gm::THREADS), lds, st, src, src_stride, in_len, batch, dst, tw);
    });
    return hipGetLastError();
}

hipError_t launch_rifft_rows(int log2n, const float2 *src, int batch, float *dst, const float2 *tw, const BigFFTWork *big, hipStream_t st)
{
    if (batch <= 0) return hipSuccess;
    if (is_big_fft(log2n)) return big ? big_rifft_rows(log2n, src, batch, dst, tw, *big, st) : hipErrorInvalidValue;
    HCV_FFT_DISPATCH(log2n - 1, {
        using Gm = FFTGeom<L>;
        size_t lds = fft_lds_bytes<L>();
        hipError_t e = allow_lds(rifft_rows_kernel<L>, lds);
        if (e != hipSuccess) return e;
        int grid = (batch + Gm::G - 1) / Gm::G;
        hipLaunchKernelGGL(rifft_rows_kernel<L>, dim3(grid), dim3(Gm::THREADS), lds, st, src, batch, dst, tw);
    });
    return hipGetLastError();
}

hipError_t launch_rifft_overlap_add(int log2n, const float2 *Y, int ksplit, long long ks_stride, long long h_first, int T, int nout,
                                    float *timeline, long long tl_stride, long long tl_mask, const float2 *tw, const BigFFTWork *big, hipStream_t st)
{
    if (T <= 0 || nout <= 0) return hipSuccess;
    {
        const long long M_ = 1ll << (log2n - 1);
        ORD_ACCESS(st, Y, 0, 1, 0, false, "partial spectra (inverse transform)");
        ORD_ACCESS(st, timeline, (h_first + 1) * M_, (h_first + 1 + T) * M_, tl_mask + 1, true, "stage timeline (overlap-add)");
    }
    if (is_big_fft(log2n))      // the caller has already reduced the split-K partials (ksplit == 1)
        return (big && ksplit == 1) ? big_rifft_overlap_add(log2n, Y, h_first, T, nout, timeline, tl_stride, tl_mask, tw, *big, st) : hipErrorInvalidValue;
    HCV_FFT_DISPATCH(log2n - 1, {
        using Gm = FFTGeom<L>;
        size_t lds = fft_lds_bytes<L>();
        hipError_t e = allow_lds(rifft_overlap_add_kernel<L>, lds);
        if (e != hipSuccess) return e;
        int grid = (T * nout + Gm::G - 1) / Gm::G;
        hipLaunchKernelGGL(rifft_overlap_add_kernel<L>, dim3(grid), dim3(Gm::THREADS), lds, st, Y, ksplit, ks_stride, h_first, T, nout, timeline,
                           tl_stride, tl_mask, tw);
    });
    return hipGetLastError();
}

hipError_t launch_rifft_emit(int log2n, const float2 *Y, int ksplit, long long ks_stride, int T, int nout, float *out, long long out_stride,
                             const float2 *tw, hipStream_t st)
{
    if (T <= 0 || nout <= 0) return hipSuccess;
    if (is_big_fft(log2n)) return hipErrorInvalidValue;
    ORD_ACCESS(st, Y, 0, 1, 0, false, "partial spectra (inverse transform)");
    if (fft_split_applies(log2n, T * nout)) return launch_rifft_emit_split(log2n, Y, ksplit, ks_stride, T, nout, out, out_stride, tw, st);
    HCV_FFT_DISPATCH(log2n - 1, {
        using Gm = FFTGeom<L>;
        size_t lds = fft_lds_bytes<L>();
        hipError_t e = allow_lds(rifft_emit_kernel<L>, lds);
        if (e != hipSuccess) return e;
        int grid = (T * nout + Gm::G - 1) / Gm::G;
        hipLaunchKernelGGL(rifft_emit_kernel<L>, dim3(grid), dim3(Gm::THREADS), lds, st, Y, ksplit, ks_stride, T, nout, out, out_stride, tw);
    });
    return hipGetLastError();
}

hipError_t launch_reduce_partials(float2 *Y, int ksplit, long long ks_stride, long long elems, hipStream_t st, float2 *dst)
{
    if (elems <= 0 || ksplit < 1) return hipSuccess;
    if (!dst) dst = Y;
    if (ksplit == 1 && dst == Y) return hipSuccess;            // (one slice elsewhere: the launch is the copy)
    ORD_ACCESS(st, Y, 0, 1, 0, dst == Y, "partial spectra (reduction)");
    if (dst != Y) ORD_ACCESS(st, dst, 0, 1, 0, true, "partial spectra (reduction's sum)");
    const long long n4 = elems / 2;
    const int grid = (int) ((n4 + 255) / 256);
    const int pin = xcd_pin_for(grid);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(grid * (pin >= 0 ? 8 : 1)), dim3(256), 0, st, reinterpret_cast<float4 *>(Y), ksplit, ks_stride / 2, n4, pin,
                       reinterpret_cast<float4 *>(dst));
    return hipGetLastError();
}

template <int OTD>
static hipError_t launch_fir_otd(const float *hist, long long hist_stride, long long hist_mask, const float *taps, int Lpad, int tap_stride, int nin,
                                 int nin_alloc, int nout, int diag, long long n0, int B, const long long *valid_from, bool check, float *out,
                                 long long out_stride, hipStream_t st)
{
    dim3 grid((B + FIR_SPB - 1) / FIR_SPB, (nout + OTD - 1) / OTD);
    size_t lds = sizeof(float) * ((size_t) Lpad + FIR_SPB + (size_t) OTD * Lpad);
    if (check)
        hipLaunchKernelGGL((fir_head_kernel<OTD, true>), grid, dim3(FIR_THREADS), lds, st, hist, hist_stride, hist_mask, taps, Lpad, tap_stride, nin,
                           nin_alloc, nout, diag, n0, B, valid_from, out, out_stride);
    else
        hipLaunchKernelGGL((fir_head_kernel<OTD, false>), grid, dim3(FIR_THREADS), lds, st, hist, hist_stride, hist_mask, taps, Lpad, tap_stride, nin,
                           nin_alloc, nout, diag, n0, B, valid_from, out, out_stride);
    return hipGetLastError();
}

bool fir_head_is_small(int B, int nin, int Lpad, int diag)
{
    return !diag && B > 0 && B <= 256 && nin >= 1 && Lpad >= 16 && Lpad <= 1024;
}

hipError_t launch_fir_head(const float *hist, long long hist_stride, long long hist_mask, const float *taps, int Lpad, int tap_stride, int nin,
                           int nin_alloc, int nout, int diag, long long n0, int B, const long long *valid_from, bool check, float *out,
                           long long out_stride, hipStream_t st, const float *din, long long in_stride, const EmitSources *emit, float *ring)
{
    if (B <= 0 || nout <= 0) return hipSuccess;
    if (ring && !(emit && din)) return hipErrorInvalidValue;
    ORD_ACCESS(st, hist, n0 - Lpad, din ? n0 : n0 + B, hist_mask + 1, false, "history ring (time-domain head)");
    if (ring) ORD_ACCESS(st, ring, n0, n0 + B, hist_mask + 1, true, "history ring (a plain small call files its samples)");
    if (emit)
        for (int k_ = 0; k_ < emit->count; k_++) ORD_ACCESS(st, emit->timeline[k_], n0, n0 + B, emit->mask[k_] + 1, true, "stage timeline (read and cleared by a plain small call)");
    if (emit && !fir_head_is_small(B, nin, Lpad, diag)) return hipErrorInvalidValue;
    // small calls of matrices with several inputs: taps split over the threads, a batch of inputs staged at once (HCV_FIR_SMALL = 0:
    // the general kernel everywhere)
    if (fir_head_is_small(B, nin, Lpad, diag))
    {
        const int per_input = FIRS_SAMPLES + 2 * Lpad;                             // floats of LDS per staged input
        const int ibmax = std::max(1, (96 * 1024 / 4 - FIRS_THREADS) / per_input);
        const int batches = (nin + ibmax - 1) / ibmax;
        const int ib = (nin + batches - 1) / batches;
        const size_t lds = sizeof(float) * ((size_t) ib * per_input + FIRS_THREADS);
        const dim3 grid((B + FIRS_SAMPLES - 1) / FIRS_SAMPLES, nout);
        EmitSources none;
        none.count = 0;
#define HCV_FIRS(CHK, EM)                                                                                                                              \
        {                                                                                                                                              \
            const hipError_t ea = allow_lds(fir_head_small_kernel<CHK, EM>, lds);                                                                      \
            if (ea != hipSuccess) return ea;                                                                                                           \
            hipLaunchKernelGGL((fir_head_small_kernel<CHK, EM>), grid, dim3(FIRS_THREADS), lds, st, hist, hist_stride, hist_mask, taps, Lpad, tap_stride, \
                               nin, nin_alloc, n0, B, valid_from, out, out_stride, ib, din, in_stride, emit ? *emit : none, ring);                     \
        }
        if (check && emit) HCV_FIRS(true, true)
        else if (check) HCV_FIRS(true, false)
        else if (emit) HCV_FIRS(false, true)
        else HCV_FIRS(false, false)
#undef HCV_FIRS
        return hipGetLastError();
    }
    // widest output tile that still leaves >= 2 workgroups per CU
    const long long sblocks = (B + FIR_SPB - 1) / FIR_SPB;
    int otd = 4;
    while (otd > 1 && sblocks * ((nout + otd - 1) / otd) < 512) otd >>= 1;
    if (diag) otd = 1;
    switch (otd)
    {
        case 4: return launch_fir_otd<4>(hist, hist_stride, hist_mask, taps, Lpad, tap_stride, nin, nin_alloc, nout, diag, n0, B, valid_from, check, out, out_stride, st);
        case 2: return launch_fir_otd<2>(hist, hist_stride, hist_mask, taps, Lpad, tap_stride, nin, nin_alloc, nout, diag, n0, B, valid_from, check, out, out_stride, st);
        default: return launch_fir_otd<1>(hist, hist_stride, hist_mask, taps, Lpad, tap_stride, nin, nin_alloc, nout, diag, n0, B, valid_from, check, out, out_stride, st);
    }
}

hipError_t launch_scatter_input(const float *in, long long in_stride, int B, int nin, float *hist, long long hist_stride, long long hist_mask,
                                long long n0, hipStream_t st)
{
    if (B <= 0 || nin <= 0) return hipSuccess;
    ORD_ACCESS(st, hist, n0, n0 + B, hist_mask + 1, true, "history ring (scatter)");
    dim3 grid(std::min((B + 255) / 256, 64), nin);
    hipLaunchKernelGGL(scatter_input_kernel, grid, dim3(256), 0, st, in, in_stride, B, hist, hist_stride, hist_mask, n0);
    return hipGetLastError();
}

hipError_t launch_emit(const EmitSources &src, long long n0, int B, int nout, const float *td, long long td_stride, float *out,
                       long long out_stride, hipStream_t st)
{
    if (B <= 0 || nout <= 0) return hipSuccess;
    for (int k_ = 0; k_ < src.count; k_++) ORD_ACCESS(st, src.timeline[k_], n0, n0 + B, src.mask[k_] + 1, true, "stage timeline (read and cleared by emit)");
    dim3 grid(std::min((B + 255) / 256, 64), nout);
    hipLaunchKernelGGL(emit_kernel, grid, dim3(256), 0, st, src, n0, B, td, td_stride, out, out_stride);
    return hipGetLastError();
}

hipError_t launch_sum_parts(const PartSources &src, long long part_stride, int B, int nout, float *out, long long out_stride, hipStream_t st)
{
    if (B <= 0 || nout <= 0 || src.count <= 0) return hipSuccess;
    dim3 grid(std::min((B + 255) / 256, 64), nout);
    hipLaunchKernelGGL(sum_parts_kernel, grid, dim3(256), 0, st, src, part_stride, B, out, out_stride);
    return hipGetLastError();
}

hipError_t launch_spectral_pointwise(float2 *a, const float2 *b, int M, float scale, int correlate, hipStream_t st)
{
    hipLaunchKernelGGL(spectral_pointwise_kernel, dim3((M + 255) / 256), dim3(256), 0, st, a, b, M, scale, correlate);
    return hipGetLastError();
}

hipError_t launch_segment_op(float *out, const float *t, long long o_off, long long off, long long n, int op, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    const int grid = (int) std::min<long long>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(segment_op_kernel, dim3(grid), dim3(256), 0, st, out, t, o_off, off, n, op);
    return hipGetLastError();
}

hipError_t launch_fold_copy(float *dst, const float *in, long long n, long long fold, int off, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    const int grid = (int) std::min<long long>((n + 2 * fold + 255) / 256, 2048);
    hipLaunchKernelGGL(fold_copy_kernel, dim3(grid), dim3(256), 0, st, dst, in, n, fold, off);
    return hipGetLastError();
}

// ---- a control call's swap section as ONE launch: up to kSwapSegs segments, each `copy` bytes from src to dst followed by `zero`
//      bytes of zeros (16-byte granularity), and up to kSwapFills 64-bit values.  The section runs on the audio thread while a stream is
//      running (hcv_engine.h: the mailbox): a dozen hipMemcpyAsync / hipMemsetAsync calls were most of what it cost there.
__global__ __launch_bounds__(256) void swap_in_kernel(SwapPlan pl)
{
    const int seg = blockIdx.y;
    if (seg < pl.nseg)
    {
        const SwapSeg sg = pl.seg[seg];
        const long long nc = sg.copy >> 4, nz = sg.zero >> 4;
        const float4 *src = static_cast<const float4 *>(sg.src);
        float4 *dst = static_cast<float4 *>(sg.dst);
        for (long long e = blockIdx.x * (long long) blockDim.x + threadIdx.x; e < nc + nz; e += (long long) gridDim.x * blockDim.x)
            dst[e] = e < nc ? src[e] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && (int) threadIdx.x < pl.nfill) *pl.fill[threadIdx.x].p = pl.fill[threadIdx.x].v;
}

hipError_t launch_swap_in(const SwapPlan &pl, hipStream_t st)
{
    if (pl.nseg <= 0 && pl.nfill <= 0) return hipSuccess;
    long long most = 0;
    for (int k = 0; k < pl.nseg; k++) most = std::max(most, (pl.seg[k].copy + pl.seg[k].zero) >> 4);
    const int gx = (int) std::max<long long>(1, std::min<long long>((most + 1023) / 1024, 512));
    hipLaunchKernelGGL(swap_in_kernel, dim3(gx, std::max(1, pl.nseg)), dim3(256), 0, st, pl);
    return hipGetLastError();
}

hipError_t launch_fill_i64(long long *p, long long n, long long v, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    int grid = (int) std::min<long long>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(fill_i64_kernel, dim3(grid), dim3(256), 0, st, p, n, v);
    return hipGetLastError();
}

hipError_t launch_regrow_spectra(const float2 *src, float2 *dst, long long pairs, int Pold, int Pnew, int M, hipStream_t st)
{
    if (pairs <= 0 || Pold <= 0) return hipSuccess;
    hipLaunchKernelGGL(regrow_spectra_kernel, dim3(2048), dim3(256), 0, st, reinterpret_cast<const float4 *>(src), reinterpret_cast<float4 *>(dst), pairs,
                       Pold, Pnew, M / 2);
    return hipGetLastError();
}

hipError_t launch_regrow_ring(const float2 *src, float2 *dst, int nin, int Rold, int Rnew, int M, long long h_last, int live, hipStream_t st)
{
    if (nin <= 0 || live <= 0) return hipSuccess;
    int M2 = M / 2;
    dim3 grid(std::min((M2 + 255) / 256, 16), nin, live);
    hipLaunchKernelGGL(regrow_ring_kernel, grid, dim3(256), 0, st, reinterpret_cast<const float4 *>(src), reinterpret_cast<float4 *>(dst), nin, Rold, Rnew,
                       M2, h_last, live);
    return hipGetLastError();
}

// (Engine::init, once per device: HIP loads a translation unit's code object at the first launch of one of its kernels — 0.3 - 0.8 ms on the
// calling thread, which for the kernels of a control section or a restart is the audio thread in mid-stream; asking for a kernel's attributes loads it now)
void preload_kernels()
{
    hipFuncAttributes fa;
    (void) hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(swap_in_kernel));
    (void) hipGetLastError();
}

} // namespace hcv
