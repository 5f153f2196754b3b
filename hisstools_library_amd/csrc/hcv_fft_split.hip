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
#include "hcv_fft_split_device.h"
#include "hcv_order_check.h"

#include <atomic>
#include <cstdlib>
#include <string>

namespace hcv
{

// One workgroup = (transform q = (t, i), residue r).
template <int LOG2N, int LOG2R, bool DIRECT>
__global__ __launch_bounds__(256) void rfft_split_kernel(float *__restrict__ hist, long long hist_stride, long long hist_mask, const float *__restrict__ in,
                                                          long long in_stride, long long n0, long long h_first, int nin, float2 *__restrict__ X, int Rring,
                                                          const float2 *__restrict__ tw, const float2 *__restrict__ tws, int pin)
{
    constexpr int N = 1 << LOG2N, M = N / 2, R = 1 << LOG2R, LOG2S = LOG2N - LOG2R, S = 1 << LOG2S, NW = R / 2 + 1;
    __shared__ __attribute__((aligned(16))) float2 lds[lds_padded(S)];
    __shared__ __attribute__((aligned(16))) float2 tl[S];
    __shared__ float2 wr[R];
    int bx = blockIdx.x;
    if (pin >= 0)
    {
        if ((bx & 7) != pin) return;
        bx >>= 3;
    }
    const int r = bx % NW, q = bx / NW;
    const int t = q / nin, i = q % nin;
    const long long h = h_first + t;
    rfft_split_body<LOG2N, LOG2R, DIRECT, false>(lds, tl, wr, threadIdx.x, r, hist + (long long) i * hist_stride, in + (long long) i * in_stride,
                                                 (h - 1) * (long long) M, n0, hist_mask, X + ((long long) i * Rring + (int) (h % Rring)) * M, tw, tws);
}

// One workgroup = (transform q = (t, o), sample classes n2 = 2 j, 2 j + 1).  Y: [ksplit][T][nout][M] partial sums.
// (TG = 1024 for launches that add up several slices: sixteen loads per thread cover them all at once)
template <int LOG2N, int LOG2R, int TG = 256>
__global__ __launch_bounds__(TG) void rifft_split_emit_kernel(const float2 *__restrict__ Y, int ksplit, long long ks_stride, int nout, float *__restrict__ out,
                                                               long long out_stride, const float2 *__restrict__ tw, const float2 *__restrict__ tws, int pin,
                                                               unsigned long long *started, int started_marks, unsigned long long started_seq)
{
    constexpr int N = 1 << LOG2N, M = N / 2, R = 1 << LOG2R, NW = R / 2;
    extern __shared__ __attribute__((aligned(16))) float2 dyn[];
    int bx = blockIdx.x;
    // ("this launch has started": marks 128 bytes apart that the n x m block's NEXT forward launch waits for — hcv_fused_nxm.hip)
    if (started && (int) blockIdx.x < started_marks && threadIdx.x == 0)
        __hip_atomic_store(started + (size_t) blockIdx.x * 16, started_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (pin >= 0)
    {
        if ((bx & 7) != pin) return;
        bx >>= 3;
    }
    const int j = bx % NW, q = bx / NW;
    const int t = q / nout, o = q % nout;
    rifft_split_body<LOG2N, LOG2R, false, TG>(dyn, threadIdx.x, j, Y + ((long long) t * nout + o) * M, ksplit, ks_stride,
                                              out + (long long) o * out_stride + (long long) t * M - M, tw, tws);
}

// ------------------------------------------------------------------------------------------------ the fused 1 x 1 block
//
// One whole-hop block of a 1 x 1 engine — forward transform, multiply-accumulate over the P live partitions, inverse transform,
// PartitionedConvolve::process for one hop (PartitionedConvolve.cpp:243-385) — as ONE launch with in-launch hand-overs in place of
// two kernel boundaries.  What crosses workgroups inside the launch (X[h], then Y) is written and read with agent-scope 8-byte
// atomics; every thread drains its stores before its workgroup is counted in, and the consumers poll the counter from one lane
// with relaxed agent loads (`bar[k]` counts arrivals over all launches and the host keeps the running totals: no reset).
//
// Forward progress BY CONSTRUCTION: no workgroup ever waits without bound.  A workgroup that spins on a counter holds its CU, so a
// consumer placed while one of its producers is not yet resident could keep that producer off the chip for good (other engines'
// launches, a CU mask or a 32-CU partition take the rest; the dispatcher's block order is per XCD and promises nothing across
// them).  Two things keep that from ever mattering:
//   * producers have the LOW block indices (forward transforms, then multiply-accumulate, whose first workgroups go on to the
//     inverse), so in the ordinary case everything a workgroup waits for was dispatched before it;
//   * every wait is BOUNDED (`spin` polls, ~1 us each), and a workgroup whose wait runs out does the missing work ITSELF: every
//     task of the launch (residue class r of a forward transform, a multiply-accumulate bin range) is a pure function of data
//     that is complete before the launch, or of tasks it can in turn complete itself, and writes the same values whoever runs
//     it and however often — so a waiter walks the unfinished tasks (a per-task flag holds the sequence number of the launch that
//     last completed it), runs them, and goes on.  Nobody then depends on a workgroup that is not running: the launch
//     completes on one CU as on 256, under any mask, beside any number of other engines (work stealing, in effect — a consumer
//     that got onto the chip early does the producers' work instead of idling).  The task's own workgroup, placed late, repeats
//     it (same values) and is the only one counted in `bar`, so the counters' running totals stay exact.
// HCV_COOP_SPIN = polls before helping (default 2048; 0 = help at once: the tests run the whole parity suite that way).
struct FusedBlockParams
{
    float *hist;
    const float *in;
    float *out;                 // the hop's output samples
    float2 *X;                  // [Rring][M] the input's spectrum ring
    const float2 *H;            // [P][M] the pair's partition spectra (lead slot first where the stage has one)
    float2 *Y;                  // [M] scratch
    const float2 *tw, *tws;
    FusedSync sy;
    long long hist_mask, n0, h;
    int Rring, P, hmac_mod;     // hmac_mod = (hop the MAC's partition 0 reads) mod Rring
    int pin;
};

// Workgroups 0 .. R/2: the forward transform's residue classes; the next R: multiply-accumulate (512 bins each), of which the
// first R/2 go on to the inverse.  A lone stage (`lone`: its partitions read X[h-1] and older, PartitionedConvolve's one hop of
// latency) needs no hand-over from the forward transform at all — it runs BESIDE the multiply-accumulate and the inverse instead
// of in front of them; a lead-slot stage (partition 0 reads X[h]) makes the multiply-accumulate wait for the nine residue classes.
template <int LOG2N, int LOG2R> struct Fused1x1Bodies
{
    static constexpr int N = 1 << LOG2N, M = N / 2, R = 1 << LOG2R, S = N >> LOG2R;
    const FusedBlockParams &a;
    float2 *dyn;
    int tid, slot;

    __device__ __forceinline__ void forward(int r) const      // residue class r of the new frame's spectrum X[h]
    {
        rfft_split_body<LOG2N, LOG2R, true, true>(dyn, dyn + lds_padded(S), dyn + lds_padded(S) + S, tid, r, a.hist, a.in, (a.h - 1) * (long long) M, a.n0,
                                                  a.hist_mask, a.X + (long long) slot * M, a.tw, a.tws);
    }
    __device__ __forceinline__ void mac_old(int) const {}
    // Y[b] = sum_p X[hmac - p][b] H[p][b] for the 512 bins of range m; bin 0 = (DC, Nyquist): two real products
    __device__ __forceinline__ void mac_new(int m) const
    {
        const int b4 = m * 256 + tid;
        const float4 *X4 = reinterpret_cast<const float4 *>(a.X), *H4 = reinterpret_cast<const float4 *>(a.H);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float ny = 0.f;
        for (int p0 = 0; p0 < a.P; p0 += 8)
        {
            float4 x[8], hh[8];
#pragma unroll
            for (int k = 0; k < 8; k++)
            {
                const int p = min(p0 + k, a.P - 1);
                int sl = a.hmac_mod - p;
                if (sl < 0) sl += a.Rring;
                const float4 *xp = X4 + (long long) sl * (M / 2) + b4;
                if (sl == slot)
                {
                    // written a moment ago by other workgroups of this launch
                    const float2 lo = get2<true>(reinterpret_cast<const float2 *>(xp)), hi = get2<true>(reinterpret_cast<const float2 *>(xp) + 1);
                    x[k] = make_float4(lo.x, lo.y, hi.x, hi.y);
                }
                else
                    x[k] = *xp;
                hh[k] = H4[(long long) p * (M / 2) + b4];
            }
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (p0 + k < a.P)
                {
                    acc.x += x[k].x * hh[k].x - x[k].y * hh[k].y;
                    acc.y += x[k].x * hh[k].y + x[k].y * hh[k].x;
                    acc.z += x[k].z * hh[k].z - x[k].w * hh[k].w;
                    acc.w += x[k].z * hh[k].w + x[k].w * hh[k].z;
                    ny += x[k].y * hh[k].y;
                }
        }
        if (b4 == 0)
        {
            acc.x += ny;
            acc.y = ny;
        }
        float2 *y = a.Y + 2 * (long long) b4;
        put2<true>(y, make_float2(acc.x, acc.y));
        put2<true>(y + 1, make_float2(acc.z, acc.w));
    }
    __device__ __forceinline__ void inverse(int j) const { rifft_split_body<LOG2N, LOG2R, true>(dyn, tid, j, a.Y, 1, 0, a.out - M, a.tw, a.tws); }      // the hop's samples
};

// (`ka` = the launch's parameters in the kernel-argument segment: a reference to the kernel's own by-value copy would force that copy into scratch)
template <int LOG2N, int LOG2R> __device__ __noinline__ void fused_1x1_slow(const FusedBlockParams *ka, float2 *dyn, int tid, int m, bool from_mac_wait, int *slot_b)
{
    constexpr int R = 1 << LOG2R;
    Fused1x1Bodies<LOG2N, LOG2R> b = { *ka, dyn, tid, (int) (ka->h % ka->Rring) };
    fused_slow_path(b, ka->sy, m, R / 2 + 1, R, R / 2, from_mac_wait, slot_b);
}

template <int LOG2N, int LOG2R>
__global__ __launch_bounds__(256) void fused_block_1x1_kernel(FusedBlockParams a)
{
    constexpr int N = 1 << LOG2N, M = N / 2, R = 1 << LOG2R, NF = R / 2 + 1;
    static_assert(M / 2 == R * 256, "one float4 (two bins) per thread in the multiply-accumulate phase");
    extern __shared__ __attribute__((aligned(16))) float2 dyn[];
    __shared__ int slot_b[2];
    int w = blockIdx.x;
    if (a.pin >= 0)
    {
        if ((w & 7) != a.pin) return;
        w >>= 3;
    }
    const int slot = (int) (a.h % a.Rring);
    const bool lone = a.hmac_mod != slot;
    Fused1x1Bodies<LOG2N, LOG2R> b = { a, dyn, (int) threadIdx.x, slot };
    fused_roles(b, a.sy, w, NF, R / 2, !lone, slot_b, [&](int m, bool from_mac_wait)
                { fused_1x1_slow<LOG2N, LOG2R>((const FusedBlockParams *) __builtin_amdgcn_kernarg_segment_ptr(), dyn, b.tid, m, from_mac_wait, slot_b); });
}

// ------------------------------------------------------------------------------------------------ launchers

const float2 *fft_split_sub_table(int log2s)
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

// log2 of the split radix per transform size (16384 points: 16 residue classes over 9 forward / 8 inverse workgroups; 4096 points: 8
// over 5 / 4) — radices 8 and 32, 4 and 16 were built and measured beside them in round 3 and are gone
static int split_radix_log2(int log2n) { return log2n == 14 ? 4 : 3; }

void fft_split_prepare(int log2n)
{
    if (log2n == 14 || log2n == 12) (void) fft_split_sub_table(log2n - split_radix_log2(log2n));
}

template <int LOG2N, int LOG2R>
static hipError_t launch_rfft_split_t(float *hist, long long hist_stride, long long hist_mask, const float *in, long long in_stride, long long n0,
                                      long long h_first, int T, int nin, float2 *X, int R, const float2 *tw, hipStream_t st)
{
    const float2 *tws = fft_split_sub_table(LOG2N - LOG2R);
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
    HCV_SPLIT_F(14, 4); HCV_SPLIT_F(12, 3);
#undef HCV_SPLIT_F
    return hipErrorInvalidValue;
}

template <int LOG2N, int LOG2R>
static hipError_t launch_rifft_split_t(const float2 *Y, int ksplit, long long ks_stride, int T, int nout, float *out, long long out_stride, const float2 *tw,
                                       hipStream_t st, unsigned long long *started, int started_marks, unsigned long long started_seq)
{
    const float2 *tws = fft_split_sub_table(LOG2N - LOG2R);
    if (!tws) return hipErrorInvalidValue;
    constexpr int M = 1 << (LOG2N - 1), S = 1 << (LOG2N - LOG2R), R = 1 << LOG2R, NW = R / 2;
    constexpr size_t lds = sizeof(float2) * (size_t) (M + lds_padded(S) + S + R);
    if (lds > 48 * 1024)
    {
        // (more than the default dynamic LDS must be asked for, once per device and instantiation)
        static std::atomic<bool> allowed[64];                 // (several engines' host threads come through here: found by ThreadSanitizer)
        int dev = 0;
        (void) hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !allowed[dev].load(std::memory_order_acquire))
        {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rifft_split_emit_kernel<LOG2N, LOG2R>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int) lds);
            if (e == hipSuccess && LOG2N == 14)
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(rifft_split_emit_kernel<LOG2N, LOG2R, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int) lds);
            if (e != hipSuccess) return e;
            if (dev >= 0 && dev < 64) allowed[dev].store(true, std::memory_order_release);
        }
    }
    const int pin = xcd_pin_for((long long) NW * T * nout);
    if (LOG2N == 14 && ksplit > 1)
        hipLaunchKernelGGL((rifft_split_emit_kernel<LOG2N, LOG2R, 1024>), dim3(NW * T * nout * (pin >= 0 ? 8 : 1)), dim3(1024), lds, st, Y, ksplit, ks_stride, nout, out,
                           out_stride, tw, tws, pin, started, started_marks, started_seq);
    else
        hipLaunchKernelGGL((rifft_split_emit_kernel<LOG2N, LOG2R>), dim3(NW * T * nout * (pin >= 0 ? 8 : 1)), dim3(256), lds, st, Y, ksplit, ks_stride, nout, out,
                           out_stride, tw, tws, pin, started, started_marks, started_seq);
    return hipGetLastError();
}

hipError_t launch_rifft_emit_split(int log2n, const float2 *Y, int ksplit, long long ks_stride, int T, int nout, float *out, long long out_stride,
                                   const float2 *tw, hipStream_t st, unsigned long long *started, int started_marks, unsigned long long started_seq)
{
    const int lr = split_radix_log2(log2n);
    ORD_ACCESS(st, Y, 0, 1, 0, false, "partial spectra (inverse transform)");
#define HCV_SPLIT_I(LN, LR) if (log2n == LN && lr == LR) return launch_rifft_split_t<LN, LR>(Y, ksplit, ks_stride, T, nout, out, out_stride, tw, st, started, started_marks, started_seq)
    HCV_SPLIT_I(14, 4); HCV_SPLIT_I(12, 3);
#undef HCV_SPLIT_I
    return hipErrorInvalidValue;
}

// The same for nin inputs and ONE output (NToMonoConvolve::process, NToMonoConvolve.cpp:35-43, for one hop) and for 1 x 1 engines
// with long impulse responses: the reduction over (input, partition) — K = nin P terms — is split over the 16 waves of 1024-thread
// workgroups (64 lanes = 128 bins each) and added up in LDS, as spectral_mac_kernel's INWG form does.  Workgroups
// 0 .. nin (R/2 + 1) - 1: residue class r of input i's forward transform; the next M/128: multiply-accumulate, of which the first
// R/2 go on to the inverse.  Bounded waits and helping as in the 1 x 1 kernel.
struct FusedNx1Params
{
    float *hist;
    const float *in;
    float *out;
    float2 *X;                  // [nin][Rring][M]
    const float2 *H;            // [nin rows of hstride float2][P][M]: output 0's pairs
    float2 *Y;
    const float2 *tw, *tws;
    FusedSync sy;
    long long hist_stride, in_stride, hist_mask, n0, h, hstride;
    int Rring, P, hmac_mod, nin;
};

template <int LOG2N, int LOG2R> struct FusedNx1Bodies
{
    static constexpr int N = 1 << LOG2N, M = N / 2, M2 = M / 2, R = 1 << LOG2R, S = N >> LOG2R, NF = R / 2 + 1, TG = 1024;
    const FusedNx1Params &a;
    float2 *dyn;
    int tid, slot;
    bool lone;
    float4 acc;
    float ny;

    __device__ __forceinline__ void forward(int task) const
    {
        const int i = task / NF, r = task % NF;
        rfft_split_body<LOG2N, LOG2R, true, true, TG>(dyn, dyn + lds_padded(S), dyn + lds_padded(S) + S, tid, r, a.hist + (long long) i * a.hist_stride,
                                                      a.in + (long long) i * a.in_stride, (a.h - 1) * (long long) M, a.n0, a.hist_mask,
                                                      a.X + ((long long) i * a.Rring + slot) * M, a.tw, a.tws);
    }
    // with a lead slot (partition 0 reads the NEW spectra X[h]) the terms of partitions >= 1 — all but nin of the nin P — are
    // accumulated first, beside the forward transforms; the nin lead terms follow once those have arrived
    __device__ __forceinline__ void mac_old(int m)
    {
        const int lane = tid & 63, ks = tid >> 6;
        const int b4 = m * 64 + lane;
        const float4 *X4 = reinterpret_cast<const float4 *>(a.X), *H4 = reinterpret_cast<const float4 *>(a.H);
        const int P1 = lone ? a.P : a.P - 1, pofs = lone ? 0 : 1;
        const int K = a.nin * P1;
        const int kper = (K + 15) / 16;
        const int k0 = ks * kper, k1 = min(K, k0 + kper);
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        ny = 0.f;
        for (int kb = k0; kb < k1; kb += 8)
        {
            float4 x[8], hh[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
            {
                const int k = min(kb + u, k1 - 1);
                const int i = k / P1, p = k - i * P1 + pofs;
                int sl = a.hmac_mod - p;
                if (sl < 0) sl += a.Rring;
                x[u] = X4[((long long) i * a.Rring + sl) * M2 + b4];
                hh[u] = H4[(long long) i * (a.hstride / 2) + (long long) p * M2 + b4];
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (kb + u < k1)
                {
                    acc.x += x[u].x * hh[u].x - x[u].y * hh[u].y;
                    acc.y += x[u].x * hh[u].y + x[u].y * hh[u].x;
                    acc.z += x[u].z * hh[u].z - x[u].w * hh[u].w;
                    acc.w += x[u].z * hh[u].w + x[u].w * hh[u].z;
                    ny += x[u].y * hh[u].y;
                }
        }
    }
    __device__ __forceinline__ void mac_new(int m)
    {
        const int lane = tid & 63, ks = tid >> 6;
        const int b4 = m * 64 + lane;
        const float4 *X4 = reinterpret_cast<const float4 *>(a.X), *H4 = reinterpret_cast<const float4 *>(a.H);
        if (!lone)
        {
            // the lead terms: input i by wave i, i + 16, ... (X[h] was written a moment ago by other workgroups of this launch)
            for (int i = ks; i < a.nin; i += 16)
            {
                const float2 *xp = reinterpret_cast<const float2 *>(X4 + ((long long) i * a.Rring + slot) * M2 + b4);
                const float2 lo = get2<true>(xp), hi = get2<true>(xp + 1);
                const float4 hv = H4[(long long) i * (a.hstride / 2) + b4];
                acc.x += lo.x * hv.x - lo.y * hv.y;
                acc.y += lo.x * hv.y + lo.y * hv.x;
                acc.z += hi.x * hv.z - hi.y * hv.w;
                acc.w += hi.x * hv.w + hi.y * hv.z;
                ny += lo.y * hv.y;
            }
        }
        if (b4 == 0)
        {
            acc.x += ny;                                    // (DC and Nyquist products are sums too: repaired per slice, added up below)
            acc.y = ny;
        }
        float4 *red = reinterpret_cast<float4 *>(dyn);      // [16][64] (free: whatever used the LDS before ended in a barrier)
        red[ks * 64 + lane] = acc;
        __syncthreads();
        if (ks == 0)
        {
            float4 sum = acc;
#pragma unroll
            for (int k = 1; k < 16; k++)
            {
                const float4 p = red[k * 64 + lane];
                sum.x += p.x; sum.y += p.y; sum.z += p.z; sum.w += p.w;
            }
            float2 *y = a.Y + 2 * (long long) b4;
            put2<true>(y, make_float2(sum.x, sum.y));
            put2<true>(y + 1, make_float2(sum.z, sum.w));
        }
    }
    __device__ __forceinline__ void inverse(int j) const { rifft_split_body<LOG2N, LOG2R, true, TG>(dyn, tid, j, a.Y, 1, 0, a.out - M, a.tw, a.tws); }
};

template <int LOG2N, int LOG2R> __device__ __noinline__ void fused_nx1_slow(const FusedNx1Params *ka, float2 *dyn, int tid, int m, bool from_mac_wait, int *slot_b)
{
    constexpr int R = 1 << LOG2R, MACW = (1 << LOG2N) / 4 / 64;
    const int slot = (int) (ka->h % ka->Rring);
    FusedNx1Bodies<LOG2N, LOG2R> b = { *ka, dyn, tid, slot, ka->hmac_mod != slot };
    fused_slow_path(b, ka->sy, m, ka->nin * (R / 2 + 1), MACW, R / 2, from_mac_wait, slot_b);
}

template <int LOG2N, int LOG2R>
__global__ __launch_bounds__(1024) void fused_block_nx1_kernel(FusedNx1Params a)
{
    constexpr int N = 1 << LOG2N, R = 1 << LOG2R, NF = R / 2 + 1;
    extern __shared__ __attribute__((aligned(16))) float2 dyn[];
    __shared__ int slot_b[2];
    const int slot = (int) (a.h % a.Rring);
    const bool lone = a.hmac_mod != slot;
    FusedNx1Bodies<LOG2N, LOG2R> b = { a, dyn, (int) threadIdx.x, slot, lone };
    fused_roles(b, a.sy, (int) blockIdx.x, a.nin * NF, R / 2, !lone, slot_b, [&](int m, bool from_mac_wait)
                { fused_nx1_slow<LOG2N, LOG2R>((const FusedNx1Params *) __builtin_amdgcn_kernarg_segment_ptr(), dyn, b.tid, m, from_mac_wait, slot_b); });
    (void) N;
}


// The same for a 1 x 1 engine whose block is SEVERAL hops of a short stage (PartitionedConvolve with 4096-point partitions called
// with 8192 samples: four hops per block, BASELINE config 2) — until now four launches: transforms, hop-tiled multiply-accumulate,
// reduction, inverse.  Hop t of the block needs the spectra of hops h + t - (1 - lead) - p: all but a handful of its terms (those
// with p < T) read spectra of EARLIER blocks, so the 16 waves of every multiply-accumulate workgroup first sum the partitions
// p >= T over their slices — each IR spectrum loaded once for all T hops, the window of input spectra sliding as in the hop-tiled
// kernel (hcv_mac_tiled.hip) — beside the T x (R/2 + 1) workgroups of the forward transforms, then wait for those and add the
// T partitions p < T (one per wave).  Workgroups 0 .. T (R/2 + 1) - 1: residue class r of hop t's forward transform; the next
// M/128: multiply-accumulate (128 bins each), of which the first T R/2 go on to the inverse of hop t = m / (R/2).  Bounded waits
// and helping as in the 1 x 1 kernel.
struct FusedHopsParams
{
    float *hist;
    const float *in;
    float *out;                 // the block's output samples (T hops)
    float2 *X;                  // [Rring][M]
    const float2 *H;            // [P][M]
    float2 *Y;                  // [T][M] scratch
    const float2 *tw, *tws;
    FusedSync sy;
    long long hist_mask, n0, h; // h = the block's first hop
    int Rring, P, T, hmac_mod;  // hmac_mod = (hop that partition 0 of hop h reads: h with a lead slot, h - 1 for a lone stage) mod Rring
};

template <int LOG2N, int LOG2R, int TMAX> struct FusedHopsBodies
{
    static constexpr int N = 1 << LOG2N, M = N / 2, M2 = M / 2, R = 1 << LOG2R, S = N >> LOG2R, NF = R / 2 + 1, TG = 1024;
    const FusedHopsParams &a;
    float2 *dyn;
    int tid;
    float4 acc[TMAX];
    float ny[TMAX];

    __device__ __forceinline__ void forward(int task) const
    {
        const int t = task / NF, r = task % NF;
        const long long h = a.h + t;
        rfft_split_body<LOG2N, LOG2R, true, true, TG>(dyn, dyn + lds_padded(S), dyn + lds_padded(S) + S, tid, r, a.hist, a.in, (h - 1) * (long long) M, a.n0,
                                                      a.hist_mask, a.X + (long long) (h % a.Rring) * M, a.tw, a.tws);
    }
    __device__ __forceinline__ int slot_of(int d) const      // ring slot of hop (hmac + d), d in (-Rring, TMAX)
    {
        int sl = a.hmac_mod + d;
        if (sl < 0) sl += a.Rring;
        if (sl >= a.Rring) sl -= a.Rring;
        return sl;
    }
    static __device__ __forceinline__ void cmac(float4 &c, float &n, const float4 &x, const float4 &hv)
    {
        c.x += x.x * hv.x - x.y * hv.y;
        c.y += x.x * hv.y + x.y * hv.x;
        c.z += x.z * hv.z - x.w * hv.w;
        c.w += x.z * hv.w + x.w * hv.z;
        n += x.y * hv.y;
    }
    // ---- partitions p >= T: every spectrum they read is older than this block
    __device__ __forceinline__ void mac_old(int m)
    {
        const int lane = tid & 63, ks = tid >> 6, T = a.T;
        const int b4 = m * 64 + lane;
        const float4 *X4 = reinterpret_cast<const float4 *>(a.X), *H4 = reinterpret_cast<const float4 *>(a.H);
#pragma unroll
        for (int t = 0; t < TMAX; t++)
        {
            acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            ny[t] = 0.f;
        }
        const int K = a.P - T;
        if (K > 0)
        {
            const int kper = (K + 15) / 16;
            const int p0 = T + ks * kper, p1 = min(a.P, p0 + kper);
            if (p0 < p1)
            {
                float4 win[TMAX];                        // win[t] = X[hmac + t - p]
#pragma unroll
                for (int t = 0; t < TMAX; t++) win[t] = X4[(long long) slot_of(min(t, T - 1) - p0) * M2 + b4];
                for (int pb = p0; pb < p1; pb += 8)
                {
                    float4 hh[8], xn[8];
#pragma unroll
                    for (int u = 0; u < 8; u++)
                    {
                        const int p = min(pb + u, p1 - 1);
                        hh[u] = H4[(long long) p * M2 + b4];
                        xn[u] = X4[(long long) slot_of(-p - 1) * M2 + b4];      // the hop the window gains at p + 1
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (pb + u < p1)
                        {
#pragma unroll
                            for (int t = 0; t < TMAX; t++) cmac(acc[t], ny[t], win[t], hh[u]);
#pragma unroll
                            for (int t = TMAX - 1; t > 0; t--) win[t] = win[t - 1];
                            win[0] = xn[u];
                        }
                }
            }
        }
    }
    // ---- partitions p < T, one per wave: hop t's term reads hop hmac + t - p, which this launch itself may be writing (t - p >= h - hmac):
    //      all of them after the hand-over, through agent-scope loads
    __device__ __forceinline__ void mac_new(int m)
    {
        const int lane = tid & 63, ks = tid >> 6, T = a.T;
        const int b4 = m * 64 + lane;
        const float4 *X4 = reinterpret_cast<const float4 *>(a.X), *H4 = reinterpret_cast<const float4 *>(a.H);
        if (ks < T && ks < a.P)
        {
            const int p = ks;
            const float4 hv = H4[(long long) p * M2 + b4];
#pragma unroll
            for (int t = 0; t < TMAX; t++)
                if (t < T)
                {
                    const float2 *xp = reinterpret_cast<const float2 *>(X4 + (long long) slot_of(t - p) * M2 + b4);
                    const float2 lo = get2<true>(xp), hi = get2<true>(xp + 1);
                    cmac(acc[t], ny[t], make_float4(lo.x, lo.y, hi.x, hi.y), hv);
                }
        }
        float4 *red = reinterpret_cast<float4 *>(dyn);      // [16][TMAX][64] (free: whatever used the LDS before ended in a barrier)
#pragma unroll
        for (int t = 0; t < TMAX; t++)
        {
            if (b4 == 0)
            {
                acc[t].x += ny[t];
                acc[t].y = ny[t];
            }
            red[(ks * TMAX + t) * 64 + lane] = acc[t];
        }
        __syncthreads();
        // hop t's sums by wave t (the slices in wave order, as the one-hop kernel adds them)
        if (ks < T)
        {
            float4 sum = red[ks * 64 + lane];
#pragma unroll
            for (int k = 1; k < 16; k++)
            {
                const float4 q = red[(k * TMAX + ks) * 64 + lane];
                sum.x += q.x; sum.y += q.y; sum.z += q.z; sum.w += q.w;
            }
            float2 *y = a.Y + (long long) ks * M + 2 * (long long) b4;
            put2<true>(y, make_float2(sum.x, sum.y));
            put2<true>(y + 1, make_float2(sum.z, sum.w));
        }
    }
    __device__ __forceinline__ void inverse(int m) const
    {
        const int t = m / (R / 2), j = m % (R / 2);
        rifft_split_body<LOG2N, LOG2R, true, TG>(dyn, tid, j, a.Y + (long long) t * M, 1, 0, a.out + (long long) t * M - M, a.tw, a.tws);
    }
};

template <int LOG2N, int LOG2R, int TMAX> __device__ __noinline__ void fused_hops_slow(const FusedHopsParams *ka, float2 *dyn, int tid, int m, bool from_mac_wait, int *slot_b)
{
    constexpr int R = 1 << LOG2R, MACW = (1 << LOG2N) / 4 / 64;
    FusedHopsBodies<LOG2N, LOG2R, TMAX> b = { *ka, dyn, tid };
    fused_slow_path(b, ka->sy, m, ka->T * (R / 2 + 1), MACW, ka->T * (R / 2), from_mac_wait, slot_b);
}

template <int LOG2N, int LOG2R, int TMAX>
__global__ __launch_bounds__(1024) void fused_block_hops_kernel(FusedHopsParams a)
{
    constexpr int N = 1 << LOG2N, R = 1 << LOG2R, NF = R / 2 + 1, MACW = N / 4 / 64;
    static_assert(TMAX * (R / 2) <= MACW, "the inverse's workgroups are the first of the multiply-accumulate's");
    extern __shared__ __attribute__((aligned(16))) float2 dyn[];
    __shared__ int slot_b[2];
    FusedHopsBodies<LOG2N, LOG2R, TMAX> b = { a, dyn, (int) threadIdx.x };
    fused_roles(b, a.sy, (int) blockIdx.x, a.T * NF, a.T * (R / 2), true, slot_b, [&](int m, bool from_mac_wait)
                { fused_hops_slow<LOG2N, LOG2R, TMAX>((const FusedHopsParams *) __builtin_amdgcn_kernarg_segment_ptr(), dyn, b.tid, m, from_mac_wait, slot_b); });
}

bool fused_block_1x1_applies(int log2n)
{
    static const bool on = !(std::getenv("HCV_COOP") && std::atoi(std::getenv("HCV_COOP")) == 0);
    return on && log2n == 14;
}

hipError_t launch_fused_block_nx1(int log2n, float *hist, long long hist_stride, long long hist_mask, const float *in, long long in_stride, long long n0, long long h,
                                  int nin, float2 *X, int Rring, const float2 *H, long long hstride, int P, long long h_mac, float2 *Y, float *out, const float2 *tw,
                                  unsigned *bar, unsigned long long *flags, unsigned *arrived, unsigned long long *seq, hipStream_t st)
{
    constexpr int LOG2N = 14, LOG2R = 4, M = 1 << (LOG2N - 1), S = 1 << (LOG2N - LOG2R), R = 1 << LOG2R;
    if (log2n != 14 || nin < 1 || nin * (R / 2 + 1) > kFusedFwdTasks) return hipErrorInvalidValue;
    const float2 *tws = fft_split_sub_table(LOG2N - LOG2R);
    if (!tws) return hipErrorInvalidValue;
    constexpr size_t lds = sizeof(float2) * (size_t) (M + lds_padded(S) + S + R);
    static std::atomic<bool> allowed[64];                 // (several engines' host threads come through here: found by ThreadSanitizer)
    int dev = 0;
    (void) hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !allowed[dev].load(std::memory_order_acquire))
    {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fused_block_nx1_kernel<LOG2N, LOG2R>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int) lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) allowed[dev].store(true, std::memory_order_release);
    }
    FusedNx1Params a;
    a.hist = hist; a.in = in; a.out = out; a.X = X; a.H = H; a.Y = Y; a.tw = tw; a.tws = tws;
    a.hist_stride = hist_stride; a.in_stride = in_stride; a.hist_mask = hist_mask; a.n0 = n0; a.h = h; a.hstride = hstride;
    a.Rring = Rring; a.P = P; a.nin = nin;
    a.hmac_mod = (int) (h_mac % Rring);
    constexpr unsigned NM = M / 128;
    static_assert(NM <= kFusedMacTasks, "");
    const unsigned nfwd = (unsigned) (nin * (R / 2 + 1));
    a.sy = fused_sync(bar, flags, arrived, *seq, nfwd, NM);
    hipLaunchKernelGGL((fused_block_nx1_kernel<LOG2N, LOG2R>), dim3(nfwd + NM), dim3(1024), lds, st, a);
    return fused_launched(arrived, seq, nfwd, NM);
}

hipError_t launch_fused_block_1x1(int log2n, float *hist, long long hist_mask, const float *in, long long n0, long long h, float2 *X, int Rring, const float2 *H,
                                  int P, long long h_mac, float2 *Y, float *out, const float2 *tw, unsigned *bar, unsigned long long *flags, unsigned *arrived,
                                  unsigned long long *seq, hipStream_t st)
{
    if (log2n != 14) return hipErrorInvalidValue;
    constexpr int LOG2N = 14, LOG2R = 4, M = 1 << (LOG2N - 1), S = 1 << (LOG2N - LOG2R), R = 1 << LOG2R;
    const float2 *tws = fft_split_sub_table(LOG2N - LOG2R);
    if (!tws) return hipErrorInvalidValue;
    constexpr size_t lds = sizeof(float2) * (size_t) (M + lds_padded(S) + S + R);
    static std::atomic<bool> allowed[64];                 // (several engines' host threads come through here: found by ThreadSanitizer)
    int dev = 0;
    (void) hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !allowed[dev].load(std::memory_order_acquire))
    {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fused_block_1x1_kernel<LOG2N, LOG2R>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int) lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) allowed[dev].store(true, std::memory_order_release);
    }
    FusedBlockParams a;
    a.hist = hist; a.in = in; a.out = out; a.X = X; a.H = H; a.Y = Y; a.tw = tw; a.tws = tws;
    a.hist_mask = hist_mask; a.n0 = n0; a.h = h; a.Rring = Rring; a.P = P;
    a.hmac_mod = (int) (h_mac % Rring);
    constexpr unsigned NFW = R / 2 + 1, NM = R, G = NFW + NM;
    a.sy = fused_sync(bar, flags, arrived, *seq, NFW, NM);
    a.pin = xcd_pin_for(G);
    hipLaunchKernelGGL((fused_block_1x1_kernel<LOG2N, LOG2R>), dim3(G * (a.pin >= 0 ? 8 : 1)), dim3(256), lds, st, a);
    return fused_launched(arrived, seq, NFW, NM);
}


bool fused_block_hops_applies(int log2n, int T)
{
    static const bool on = !(std::getenv("HCV_COOP") && std::atoi(std::getenv("HCV_COOP")) == 0);
    return on && log2n == 12 && T >= 2 && T <= 4;
}

hipError_t launch_fused_block_hops(int log2n, float *hist, long long hist_mask, const float *in, long long n0, long long h, int T, float2 *X, int Rring,
                                   const float2 *H, int P, long long h_mac, float2 *Y, float *out, const float2 *tw, unsigned *bar, unsigned long long *flags,
                                   unsigned *arrived, unsigned long long *seq, hipStream_t st)
{
    if (log2n != 12 || T < 2 || T > 4 || P < 1) return hipErrorInvalidValue;
    constexpr int LOG2N = 12, LOG2R = 3, TMAX = 4, M = 1 << (LOG2N - 1), S = 1 << (LOG2N - LOG2R), R = 1 << LOG2R, MACW = M / 128;
    const float2 *tws = fft_split_sub_table(LOG2N - LOG2R);
    if (!tws) return hipErrorInvalidValue;
    constexpr size_t lds_fft = sizeof(float2) * (size_t) (M + lds_padded(S) + S + R), lds_red = sizeof(float4) * 16 * TMAX * 64;
    constexpr size_t lds = lds_fft > lds_red ? lds_fft : lds_red;
    static std::atomic<bool> allowed[64];                 // (several engines' host threads come through here: found by ThreadSanitizer)
    int dev = 0;
    (void) hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !allowed[dev].load(std::memory_order_acquire))
    {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fused_block_hops_kernel<LOG2N, LOG2R, TMAX>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) allowed[dev].store(true, std::memory_order_release);
    }
    FusedHopsParams a;
    a.hist = hist; a.in = in; a.out = out; a.X = X; a.H = H; a.Y = Y; a.tw = tw; a.tws = tws;
    a.hist_mask = hist_mask; a.n0 = n0; a.h = h; a.Rring = Rring; a.P = P; a.T = T;
    a.hmac_mod = (int) (h_mac % Rring);
    const unsigned nfwd = (unsigned) (T * (R / 2 + 1));
    a.sy = fused_sync(bar, flags, arrived, *seq, nfwd, MACW);
    hipLaunchKernelGGL((fused_block_hops_kernel<LOG2N, LOG2R, TMAX>), dim3(nfwd + MACW), dim3(1024), lds, st, a);
    return fused_launched(arrived, seq, nfwd, MACW);
}

// (Engine::init, once per device: HIP loads a translation unit's code object at the first launch of one of its kernels — 0.3 - 0.8 ms on the
// calling thread, which for the kernels of a control section or a restart is the audio thread in mid-stream; asking for a kernel's attributes loads it now)
void preload_fft_split()
{
    hipFuncAttributes fa;
    (void) hipFuncGetAttributes(&fa, reinterpret_cast<const void *>((rifft_split_emit_kernel<14, 4, 256>)));
    (void) hipGetLastError();
}

} // namespace hcv
