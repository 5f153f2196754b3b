// The general hisstools_* FFT surface on gfx950 (see hcv_fftx.h).  Reference behaviour restated:
//   hisstools_fft / ifft      HISSTools_FFT_Core.h:1325-1346   in-place complex transform of split data; the inverse is the
//                                                              forward transform with the real/imaginary pointers exchanged
//   hisstools_rfft / rifft    HISSTools_FFT_Core.h:1350-1374   complex transform of N/2 points + the real pass (:934-988):
//                                                              forward spectrum doubled, DC/Nyquist packed in bin 0
//   small cases               HISSTools_FFT_Core.h:994-1150    complex 2 points and real 2 / 4 points
//   unzip / unzip_zero / zip  HISSTools_FFT_Core.h:1199-1287   even/odd de/interleave, zero padding, odd trailing sample
//
// Device plan per transform of M complex points (M = N/2 for the real transforms):
//   M <= 2                    one thread per transform                                           (fx_tiny_kernel)
//   M <= 16384 (f32) / 8192 (f64)   one LDS-resident Stockham transform per thread group: HBM is read once and
//                             written once, the real pre/post pass and the zip/unzip are fused into the load/store
//   larger, up to 2^22        four-step through HBM: column transforms (+ twiddle) into scratch, row transforms out; the
//                             loads are fused into the column pass and the stores into the row pass (the forward real
//                             post pass needs bins k and M-k together, so it runs as a third pass)

#include "hcv_fftx.h"
#include "hcv_engine.h"
#include "hcv_fft_device.h"
#include "hcv_fused_sync.h"

#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

namespace hcv
{

namespace
{
    template <class T> struct Cx;
    template <> struct Cx<float> { typedef float2 type; };
    template <> struct Cx<double> { typedef double2 type; };

    enum { L_SPLIT = 0, L_ZIP = 1, L_PRE = 2 };
    constexpr int kFxStreamRing = 512;      // ring slots of the one-launch four-step batch (fx_stream_kernel) at most
    enum { S_SPLIT = 0, S_ZIP = 1, S_POST = 2 };

    template <class T> struct FxK
    {
        const void *sa;                 // split real part, or the interleaved samples (float when src_f32)
        const T *sb;                    // split imaginary part
        T *da, *db;                     // split destination, or da = interleaved samples
        long long sstride, dstride, in_len;
        int load, store, swap_out, src_f32;
        long long batch;
    };

    // -------------------------------------------------------------------------------------------- element access

    // the same descriptor with its pointers advanced to transform q (cheap scalar arithmetic when q is wave-uniform)
    template <class T> __device__ __forceinline__ FxK<T> fx_at(FxK<T> a, long long q)
    {
        const long long so = q * a.sstride, dof = q * a.dstride;
        a.sa = a.src_f32 ? static_cast<const void *>(static_cast<const float *>(a.sa) + so) : static_cast<const void *>(static_cast<const T *>(a.sa) + so);
        if (a.sb) a.sb += so;
        a.da += dof;
        if (a.db) a.db += dof;
        return a;
    }

    // element n of the M-point complex input of the transform that starts `off` elements after the descriptor's pointers
    // (off = transform-within-workgroup * stride: 32-bit lane arithmetic on top of scalar base pointers)
    template <class T, class C>
    __device__ __forceinline__ C fx_load(const FxK<T> &a, int off, int n, int M, const C *__restrict__ twN)
    {
        if (a.load == L_SPLIT)
        {
            const T *re = static_cast<const T *>(a.sa), *im = a.sb;
            return C(re[off + n], im[off + n]);
        }
        if (a.load == L_ZIP)
        {
            // (the pair of samples as ONE load of twice the width where it lies inside the input and is aligned)
            const int i0 = 2 * n;
            if (a.src_f32)
            {
                const float *x = static_cast<const float *>(a.sa) + off + i0;
                if (i0 + 1 < a.in_len && (reinterpret_cast<uintptr_t>(x) & 7) == 0)
                {
                    const float2 v = *reinterpret_cast<const float2 *>(x);
                    return C((T) v.x, (T) v.y);
                }
                return C(i0 < a.in_len ? (T) x[0] : (T) 0, i0 + 1 < a.in_len ? (T) x[1] : (T) 0);
            }
            const T *x = static_cast<const T *>(a.sa) + off + i0;
            if (i0 + 1 < a.in_len && (reinterpret_cast<uintptr_t>(x) & (2 * sizeof(T) - 1)) == 0) return *reinterpret_cast<const C *>(x);
            return C(i0 < a.in_len ? x[0] : (T) 0, i0 + 1 < a.in_len ? x[1] : (T) 0);
        }
        // L_PRE: pass_real_trig_table<true> (Core.h:934-988), delivered with re/im exchanged so that the forward
        // transform that follows acts as the inverse (Core.h:1341-1346)
        const T *re = static_cast<const T *>(a.sa) + off;
        const T *im = a.sb + off;
        if (n == 0)
        {
            const T r = re[0], i = im[0];
            return C(r - i, r + i);
        }
        const bool lo = n <= M / 2;
        const int k = lo ? n : M - n, m = M - k;
        const C w = twN[k];
        const T c = -w.x, sn = w.y;
        const T r1 = re[k], i1 = im[k], r2 = re[m], i2 = im[m];
        const T r3 = r1 + r2, i3 = i1 + i2, r4 = r1 - r2, i4 = i1 - i2;
        const T u1 = (c * i3) + (sn * r4);
        const T u2 = (sn * i3) - (c * r4);
        return lo ? C(u2 + i4, r3 + u1) : C(u2 - i4, r3 - u1);
    }

    template <class T, class C>
    __device__ __forceinline__ void fx_store(const FxK<T> &a, int off, int k, C v)
    {
        if (a.swap_out) v = C(v.y, v.x);
        if (a.store == S_SPLIT)
        {
            a.da[off + k] = v.x;
            a.db[off + k] = v.y;
        }
        else
        {
            a.da[off + 2 * k] = v.x;
            a.da[off + 2 * k + 1] = v.y;
        }
    }

    // pass_real_trig_table<false> for the bin pair (k, M-k), k in [0, M/2]
    template <class T, class C>
    __device__ __forceinline__ void fx_post(const FxK<T> &a, int off, int k, int M, C z1, C z2, const C *__restrict__ twN)
    {
        T *re = a.da + off, *im = a.db + off;
        if (k == 0)
        {
            const T t1 = z1.x + z1.y, t2 = z1.x - z1.y;
            re[0] = t1 + t1;
            im[0] = t2 + t2;
            return;
        }
        const int m = M - k;
        const C w = twN[k];
        const T r3 = z1.x + z2.x, i3 = z1.y + z2.y, r4 = z1.x - z2.x, i4 = z1.y - z2.y;
        const T u1 = (w.x * i3) + (w.y * r4);
        const T u2 = (w.y * i3) - (w.x * r4);
        re[k] = r3 + u1;
        im[k] = u2 + i4;
        re[m] = r3 - u1;
        im[m] = u2 - i4;
    }

    // first-pass source / last-pass sink of the LDS-resident kernel: HBM <-> butterfly registers
    template <class T, int M> struct FxLoad
    {
        typedef typename Cx<T>::type C;
        static constexpr bool is_lds = false;
        const FxK<T> &a;
        int off;
        const C *__restrict__ tw;
        bool live;
        __device__ __forceinline__ C operator()(int n) const { return live ? fx_load<T, C>(a, off, n, M, tw) : C(0, 0); }
    };

    template <class T> struct FxStore
    {
        typedef typename Cx<T>::type C;
        static constexpr bool is_lds = false;
        const FxK<T> &a;
        int off;
        bool live;
        __device__ __forceinline__ void operator()(int k, C v) const
        {
            if (live) fx_store<T, C>(a, off, k, v);
        }
    };

    // 16-byte vectors of the split arrays: 4 floats / 2 doubles of adjacent columns (or bins) per lane.  The tiles' rows are runs
    // of 128 bytes per split array, so one-element-per-lane access issued four (two) times the memory instructions for the same
    // bytes; the four-step passes are bound by exactly that issue rate, not by HBM (their scratch stays in the Infinity Cache).
    template <class T> struct FxVec;
    template <> struct FxVec<float> { typedef float4 type; static constexpr int V = 4; };
    template <> struct FxVec<double> { typedef double2 type; static constexpr int V = 2; };
    __device__ __forceinline__ float fx_get(const float4 &v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }
    __device__ __forceinline__ double fx_get(const double2 &v, int j) { return j == 0 ? v.x : v.y; }
    __device__ __forceinline__ void fx_put(float4 &v, int j, float x) { if (j == 0) v.x = x; else if (j == 1) v.y = x; else if (j == 2) v.z = x; else v.w = x; }
    __device__ __forceinline__ void fx_put(double2 &v, int j, double x) { if (j == 0) v.x = x; else v.y = x; }
    template <class T> __device__ __forceinline__ bool fx_aligned16(const T *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

    // pass_real_trig_table<true> for the bin pair (k, M - k), k in [0, M/2], on a split spectrum staged in LDS, in place: the
    // arithmetic of fx_load's L_PRE branch, operand for operand (the pair is read and written by this one thread)
    template <class T, class C>
    __device__ __forceinline__ void fx_pre_inplace(LdsBuf<C> d, int k, int M, const C *__restrict__ tw)
    {
        if (k == 0)
        {
            const C z = d[0];
            d[0] = C(z.x - z.y, z.x + z.y);
            return;
        }
        const int m = M - k;
        const C w = tw[k];
        const T c = -w.x, sn = w.y;
        const C z1 = d[k], z2 = d[m];
        const T r3 = z1.x + z2.x, i3 = z1.y + z2.y, r4 = z1.x - z2.x, i4 = z1.y - z2.y;
        const T u1 = (c * i3) + (sn * r4);
        const T u2 = (sn * i3) - (c * r4);
        d[k] = C(u2 + i4, r3 + u1);
        if (m != k) d[m] = C(u2 - i4, r3 - u1);
    }

    // pass_real_trig_table<false> for the bin pair (k, M - k), k in [0, M/2], in place on a transform's result in LDS (real part
    // -> .x, imaginary -> .y of the split output): fx_post's arithmetic, operand for operand; bin M/2 keeps the value fx_post's
    // second pair of stores leaves there
    template <class T, class C>
    __device__ __forceinline__ void fx_post_inplace(LdsBuf<C> d, int k, int M, const C *__restrict__ tw)
    {
        if (k == 0)
        {
            const C z = d[0];
            const T t1 = z.x + z.y, t2 = z.x - z.y;
            d[0] = C(t1 + t1, t2 + t2);
            return;
        }
        const int m = M - k;
        const C w = tw[k];
        const C z1 = d[k], z2 = d[m];
        const T r3 = z1.x + z2.x, i3 = z1.y + z2.y, r4 = z1.x - z2.x, i4 = z1.y - z2.y;
        const T u1 = (w.x * i3) + (w.y * r4);
        const T u2 = (w.y * i3) - (w.x * r4);
        if (m != k) d[k] = C(r3 + u1, u2 + i4);
        d[m] = C(r3 - u1, u2 - i4);
    }

    // -------------------------------------------------------------------------------------------- LDS-resident transforms

#ifndef HCV_FX_STAGE_TG
#define HCV_FX_STAGE_TG 32        // thread groups up to this size may move their transforms through LDS (below)
#endif


    // second launch bound = waves per SIMD the register budget must allow (HIP semantics): 4 -> 128 VGPRs in float
    // STAGED: thread groups up to HCV_FX_STAGE_TG threads pass the workgroup's transforms through LDS on their way in and out
    // instead of loading them into / storing them from the butterflies' registers; larger groups stage the real inverse's
    // pre-pass (the forward transform's post-pass the same way measured slower there: 2^11 4.4 -> 4.1 TB/s, 2^15 2.5 -> 2.2).
    // Chosen per launch (launch_lds).
    template <class T, int LOG2M, bool STAGED>
    __global__ __launch_bounds__((FFTGeom<LOG2M>::THREADS), (sizeof(T) == 4 ? 4 : 2)) void fx_lds_kernel(FxK<T> a0, const typename Cx<T>::type *__restrict__ tw)
    {
        typedef typename Cx<T>::type C;
        typedef FFTGeom<LOG2M> Gm;
        constexpr int M = Gm::M, TG = Gm::TG, G = Gm::G;
        extern __shared__ __attribute__((aligned(16))) unsigned char fx_raw[];
        C *lds = reinterpret_cast<C *>(fx_raw);

        // groups of whole waves are wave-uniform: keeps the per-transform pointers scalar and the lane addresses 32-bit
        const int g = (TG % 64 == 0) ? __builtin_amdgcn_readfirstlane((int) (threadIdx.x / TG)) : (int) (threadIdx.x / TG), t = threadIdx.x % TG;
        const long long q0 = (long long) blockIdx.x * G;
        const bool live = q0 + g < a0.batch;
        // scalar pointers: to this thread group's transform when the group is wave-uniform, else to the workgroup's first
        // transform with 32-bit lane offsets on top
        constexpr bool UNI = TG % 64 == 0;
        const FxK<T> a = fx_at(a0, UNI ? q0 + g : q0);
        const int soff = UNI ? 0 : g * (int) a.sstride, doff = UNI ? 0 : g * (int) a.dstride;
        const LdsBuf<C> s = { lds + g * lds_padded(M) };
        if constexpr (STAGED && TG <= HCV_FX_STAGE_TG)
        {
            // Sixteen or more transforms per wave (M <= 64; measured: 2^4 1.4 -> 3.8 TB/s, 2^6 +13 %): with one butterfly's
            // operands per lane a load instruction would touch 16 cache lines for 256 useful bytes.  The workgroup instead
            // moves its G transforms through LDS with consecutive lanes on consecutive elements (one contiguous run when the
            // batch is dense), in both directions — as 16-byte vectors of the split arrays where strides and alignment allow
            // (a quarter of the memory instructions for the same bytes).
            typedef typename FxVec<T>::type VT;
            constexpr int V = FxVec<T>::V;
            const long long left = a0.batch - q0;
            const int groups = left < G ? (int) left : G;
            // (the real inverse's pre-pass, L_PRE, reads the same split arrays: staged raw, combined in LDS below)
            const bool pre = a.load == L_PRE;
            if ((a.load == L_SPLIT || pre) && M % V == 0 && a.sstride % V == 0 && fx_aligned16(static_cast<const T *>(a.sa)) && fx_aligned16(a.sb))   // (wave-uniform)
            {
                const T *re = static_cast<const T *>(a.sa), *im = a.sb;
                for (int e = threadIdx.x; e < groups * (M / V); e += Gm::THREADS)
                {
                    const int gg = e / (M / V), n = (e % (M / V)) * V;
                    const long long idx = (long long) gg * a.sstride + n;
                    const VT vr = *reinterpret_cast<const VT *>(re + idx), vi = *reinterpret_cast<const VT *>(im + idx);
                    const LdsBuf<C> d = { lds + gg * lds_padded(M) };
#pragma unroll
                    for (int j = 0; j < V; j++) d[n + j] = C(fx_get(vr, j), fx_get(vi, j));
                }
            }
            else
            {
                const T *re = static_cast<const T *>(a.sa), *im = a.sb;
                for (int e = threadIdx.x; e < groups * M; e += Gm::THREADS)
                {
                    const int gg = e / M, n = e % M;
                    const int off = gg * (int) a.sstride;
                    LdsBuf<C>{ lds + gg * lds_padded(M) }[n] = pre ? C(re[off + n], im[off + n]) : fx_load<T, C>(a, off, n, M, tw);
                }
            }
            __syncthreads();
            if (pre)
            {
                // pass_real_trig_table<true> on the staged spectrum, in place: the thread that owns the bin pair (k, M - k)
                // reads both and writes both (the arithmetic of fx_load's L_PRE branch, operand for operand)
                for (int e = threadIdx.x; e < groups * (M / 2 + 1); e += Gm::THREADS)
                {
                    const int gg = e / (M / 2 + 1), k = e % (M / 2 + 1);
                    fx_pre_inplace<T, C>(LdsBuf<C>{ lds + gg * lds_padded(M) }, k, M, tw);
                }
                __syncthreads();
            }
            LdsFFT<LOG2M, TG, C>::run(s, t, tw);
            const bool vec_out = M % V == 0 && a.dstride % V == 0 && fx_aligned16(a.da) && fx_aligned16(a.db);                                   // (wave-uniform)
            if (a.store == S_POST && vec_out)
            {
                // pass_real_trig_table<false> in place (fx_post's arithmetic, operand for operand; bin M/2 keeps the value
                // fx_post's second pair of stores leaves there), then out as vectors below
                for (int e = threadIdx.x; e < groups * (M / 2 + 1); e += Gm::THREADS)
                {
                    const int gg = e / (M / 2 + 1), k = e % (M / 2 + 1);
                    fx_post_inplace<T, C>(LdsBuf<C>{ lds + gg * lds_padded(M) }, k, M, tw);
                }
                __syncthreads();
            }
            if (a.store == S_POST && !vec_out)
            {
                if (live)
                    for (int k = t; k <= M / 2; k += TG) fx_post<T, C>(a, doff, k, M, s[k], s[(M - k) & (M - 1)], tw);
            }
            else if ((a.store == S_SPLIT || a.store == S_POST) && vec_out)
            {
                const bool swap = a.store == S_SPLIT && a.swap_out;
                for (int e = threadIdx.x; e < groups * (M / V); e += Gm::THREADS)
                {
                    const int gg = e / (M / V), k = (e % (M / V)) * V;
                    const long long idx = (long long) gg * a.dstride + k;
                    const LdsBuf<C> b = { lds + gg * lds_padded(M) };
                    VT vr, vi;
#pragma unroll
                    for (int j = 0; j < V; j++)
                    {
                        const C v = b[k + j];
                        fx_put(vr, j, swap ? v.y : v.x);
                        fx_put(vi, j, swap ? v.x : v.y);
                    }
                    *reinterpret_cast<VT *>(a.da + idx) = vr;
                    *reinterpret_cast<VT *>(a.db + idx) = vi;
                }
            }
            else
            {
                for (int e = threadIdx.x; e < groups * M; e += Gm::THREADS)
                {
                    const int gg = e / M, k = e % M;
                    fx_store<T, C>(a, gg * (int) a.dstride, k, LdsBuf<C>{ lds + gg * lds_padded(M) }[k]);
                }
            }
            return;
        }
        if constexpr (UNI && STAGED)
        {
            // The real inverse's pre-pass straight into the butterflies costs four scattered loads and a twiddle per element
            // (bins k and M - k of both arrays).  The group instead stages its split spectrum in LDS as 16-byte vectors,
            // combines the bin pairs there, and runs the transform from LDS (2^12: 3.0 -> see DESIGN section 7).
            typedef typename FxVec<T>::type VT;
            constexpr int V = FxVec<T>::V;
            if (a0.load == L_PRE && a0.store != S_POST && M % V == 0 && a0.sstride % V == 0 && fx_aligned16(static_cast<const T *>(a0.sa)) && fx_aligned16(a0.sb))   // (workgroup-uniform)
            {
                if (live)
                {
                    const T *re = static_cast<const T *>(a.sa), *im = a.sb;
                    for (int e = t; e < M / V; e += TG)
                    {
                        const VT vr = reinterpret_cast<const VT *>(re)[e], vi = reinterpret_cast<const VT *>(im)[e];
#pragma unroll
                        for (int j = 0; j < V; j++) s[e * V + j] = C(fx_get(vr, j), fx_get(vi, j));
                    }
                }
                __syncthreads();
                if (live)
                    for (int k = t; k <= M / 2; k += TG) fx_pre_inplace<T, C>(s, k, M, tw);
                __syncthreads();
                LdsFFT<LOG2M, TG, C>::run(LdsIO<C>{ s }, FxStore<T>{ a, doff, live }, s, t, tw);
                return;
            }
        }
        const FxLoad<T, M> ld = { a, soff, tw, live };
        if (a.store == S_POST)
        {
            // the real post pass pairs bin k with bin M-k: finish in LDS, then combine
            LdsFFT<LOG2M, TG, C>::run(ld, LdsIO<C>{ s }, s, t, tw);
            if (live)
                for (int k = t; k <= M / 2; k += TG) fx_post<T, C>(a, doff, k, M, s[k], s[(M - k) & (M - 1)], tw);
        }
        else
            LdsFFT<LOG2M, TG, C>::run(ld, FxStore<T>{ a, doff, live }, s, t, tw);
    }

    // -------------------------------------------------------------------------------------------- tiny transforms (M <= 2)

    // kind: 0 complex, 1 real forward, 2 real inverse; one thread per transform
    template <class T>
    __global__ void fx_tiny_kernel(FxK<T> a, int kind, int log2n)
    {
        typedef typename Cx<T>::type C;
        const long long q = (long long) blockIdx.x * blockDim.x + threadIdx.x;
        if (q >= a.batch) return;
        const int M = kind == 0 ? (1 << log2n) : ((1 << log2n) >> 1);
        if (M == 0) return;
        // the pre pass does not apply at these sizes: read the packed spectrum as plain split data
        FxK<T> ld = fx_at(a, q);
        if (ld.load == L_PRE) ld.load = L_SPLIT;
        T re[2], im[2];
        for (int n = 0; n < M; n++)
        {
            const C v = fx_load<T, C>(ld, 0, n, M, nullptr);
            re[n] = v.x;
            im[n] = v.y;
        }
        if (kind == 0)
        {
            if (M == 2)                                             // small_fft, Core.h:1000-1011
            {
                const T r1 = re[0], r2 = re[1], i1 = im[0], i2 = im[1];
                re[0] = r1 + r2; re[1] = r1 - r2; im[0] = i1 + i2; im[1] = i1 - i2;
            }
        }
        else if (kind == 1)
        {
            if (M == 1)                                             // small_real_fft<false>, Core.h:1101-1108
            {
                const T r1 = re[0] + re[0], r2 = im[0] + im[0];
                re[0] = r1 + r2; im[0] = r1 - r2;
            }
            else                                                    // Core.h:1110-1128
            {
                const T r1 = re[0] + re[1], r2 = re[0] - re[1], i1 = im[0] + im[1], i2 = im[1] - im[0];
                const T r3 = r1 + i1, i3 = r1 - i1;
                re[0] = r3 + r3; re[1] = r2 + r2; im[0] = i3 + i3; im[1] = i2 + i2;
            }
        }
        else
        {
            if (M == 1)                                             // small_real_fft<true>
            {
                const T r1 = re[0], r2 = im[0];
                re[0] = r1 + r2; im[0] = r1 - r2;
            }
            else                                                    // Core.h:1130-1148
            {
                const T i1 = re[0], r2 = re[1] + re[1], i2 = im[0], r4 = im[1] + im[1];
                const T r1 = i1 + i2, r3 = i1 - i2;
                re[0] = r1 + r2; re[1] = r1 - r2; im[0] = r3 - r4; im[1] = r3 + r4;
            }
        }
        FxK<T> sv = fx_at(a, q);
        sv.swap_out = 0;                                            // the explicit formulas above already give (re, im)
        if (sv.store == S_POST) sv.store = S_SPLIT;
        for (int k = 0; k < M; k++) fx_store<T, C>(sv, 0, k, C(re[k], im[k]));
    }

    // -------------------------------------------------------------------------------------------- four-step passes

    template <int P, int ELEM_BYTES> using FxTile = FourStepTile<P, ELEM_BYTES>;

    template <class C>
    __device__ __forceinline__ C fx_root_rt(const C *__restrict__ tw, int idx, int M)
    {
        const C w = tw[idx & (M - 1)];
        return (idx & M) ? C(-w.x, -w.y) : w;
    }

    // The four-step twiddle W_M^(n2 k1) = exp(-2 pi i idx / 2M), idx = 2 n2 k1 (an index into the 2M-th roots).  Every element of
    // a column tile needs its own, and as a table look-up that is one scattered 8-byte load per element from a table far larger
    // than the L1 (512 KiB at 2^16 points): the column pass was bound by those gathers.  In float the value is computed instead
    // (sincospi of the exactly reduced fraction: <= 2 ulp, inside the table's own rounding for the stated tolerance); double keeps
    // the table (a double-precision sincospi costs more than the load).
    __device__ __forceinline__ float2 fx_twiddle(const float2 *__restrict__, int idx, int M)
    {
        float sn, cs;
        sincospif(-(float) (idx & (2 * M - 1)) * (1.0f / (float) M), &sn, &cs);      // idx / 2M turns = idx / M half-turns; M is a power of two: reciprocal and product exact (a float division is ten instructions)
        return make_float2(cs, sn);
    }
    __device__ __forceinline__ double2 fx_twiddle(const double2 *__restrict__, int idx, int M)
    {
        double sn, cs;
        sincospi(-(double) (idx & (2 * M - 1)) / (double) M, &sn, &cs);
        return make_double2(cs, sn);
    }


#ifdef HCV_FX_PHASE_TIMING
    // diagnostic builds (tools/micro/fx_phases.py): time from a workgroup's start to the end of its load / transform / store phase,
    // summed over workgroups, in 10 ns ticks of the constant-rate counter; [0..3] column pass, [4..7] row pass, last = workgroups
    __device__ unsigned long long g_fx_phase[8];
#define FX_T0 const unsigned long long fx_t0 = wall_clock64();
#define FX_MARK(i) do { __syncthreads(); if (threadIdx.x == 0) atomicAdd(&g_fx_phase[i], wall_clock64() - fx_t0); } while (0)
#define FX_COUNT(i) do { if (threadIdx.x == 0) atomicAdd(&g_fx_phase[i], 1ull); } while (0)
#else
#define FX_T0
#define FX_MARK(i)
#define FX_COUNT(i)
#endif

    // M = M1 * M2, n = M2*n1 + n2, k = k1 + M1*k2.   cols: for every n2 an M1-point transform over n1, times W_M^(n2 k1)
    template <class T, int L1>
    __global__ __launch_bounds__((FxTile<(1 << L1), (int) sizeof(typename Cx<T>::type)>::THREADS)) void fx_cols_kernel(FxK<T> a0, typename Cx<T>::type *__restrict__ work, int M2, int M, long long q0,
                                                          const typename Cx<T>::type *__restrict__ tw1, const typename Cx<T>::type *__restrict__ twN)
    {
        typedef typename Cx<T>::type C;
        constexpr int M1 = 1 << L1;
        typedef FxTile<M1, (int) sizeof(C)> Tile;
        constexpr int TG = Tile::TG, G = Tile::G, COLS = Tile::TILE, NT = Tile::THREADS;
        extern __shared__ __attribute__((aligned(16))) unsigned char fx_raw[];
        C *lds = reinterpret_cast<C *>(fx_raw);                                // [COLS][M1]

        FX_T0
        const int col0 = fourstep_tile_of(blockIdx.x, gridDim.x) * COLS;
        const FxK<T> a = fx_at(a0, q0 + blockIdx.y);
        typedef typename FxVec<T>::type VT;
        constexpr int V = FxVec<T>::V;
        if (a.load == L_SPLIT && COLS % V == 0 && fx_aligned16(static_cast<const T *>(a.sa)) && fx_aligned16(a.sb))      // (wave-uniform)
        {
            const T *re = static_cast<const T *>(a.sa), *im = a.sb;
            for (int e = threadIdx.x; e < (COLS / V) * M1; e += NT)
            {
                const int cv = (e % (COLS / V)) * V, n1 = e / (COLS / V);
                const long long idx = (long long) n1 * M2 + col0 + cv;
                const VT vr = *reinterpret_cast<const VT *>(re + idx), vi = *reinterpret_cast<const VT *>(im + idx);
#pragma unroll
                for (int j = 0; j < V; j++) LdsBuf<C>{ lds + (cv + j) * fourstep_pitch(M1) }[n1] = C(fx_get(vr, j), fx_get(vi, j));
            }
        }
        else
        {
            for (int e = threadIdx.x; e < COLS * M1; e += NT)
            {
                const int c = e % COLS, n1 = e / COLS;
                LdsBuf<C>{ lds + c * fourstep_pitch(M1) }[n1] = fx_load<T, C>(a, 0, n1 * M2 + col0 + c, M, twN);
            }
        }
        __syncthreads();
        FX_MARK(0);
        const int g = threadIdx.x / TG, t = threadIdx.x % TG;
        for (int c0 = 0; c0 < COLS; c0 += G) LdsFFT<L1, TG, C>::run(LdsBuf<C>{ lds + (c0 + g) * fourstep_pitch(M1) }, t, tw1);
        FX_MARK(1);
        C *out = work + (long long) blockIdx.y * M;
        constexpr int CV = 16 / (int) sizeof(C) > 0 ? 16 / (int) sizeof(C) : 1;       // complex values per 16-byte store: 2 (float), 1 (double)
        constexpr int NI = ((COLS / CV) * M1) / NT, DK = NT / (COLS / CV);      // a thread's elements: its column pair at k1 = k10 + i DK
        if constexpr (((COLS / CV) * M1) % NT == 0 && (NI & (NI - 1)) == 0 && NI >= 2 && NI <= 16)
        {
            // The pass is bound by its instruction stream, and one sincospi per element (sixteen per thread) was 40 % of it.
            // A thread's twiddles per column are W^(n2 k10) * S^i, i < NI, with S = W^(n2 DK): the first and S, S^2, S^4, ...
            // each from sincospi of an exactly reduced argument (1 + log2 NI calls instead of NI), the rest as products
            // (term i = term (i - b) * S^b for its top bit b): at most log2 NI factors, each within 2 ulp.
            const int c = ((int) threadIdx.x % (COLS / CV)) * CV, k10 = (int) threadIdx.x / (COLS / CV);
            const int n2 = col0 + c;
            C w[CV][NI];
#pragma unroll
            for (int j = 0; j < CV; j++)
            {
                w[j][0] = fx_twiddle(twN, 2 * (n2 + j) * k10, M);
#pragma unroll
                for (int bit = 1; bit < NI; bit *= 2)
                {
                    // (S^bit from its own exactly reduced argument: squaring S instead would double its error every time)
                    const C sb = fx_twiddle(twN, 2 * (n2 + j) * DK * bit, M);
#pragma unroll
                    for (int i = 0; i < bit; i++) w[j][bit + i] = cmul(w[j][i], sb);
                }
            }
#pragma unroll
            for (int i = 0; i < NI; i++)
            {
                const int k1 = k10 + i * DK;
                C v[CV];
#pragma unroll
                for (int j = 0; j < CV; j++) v[j] = cmul(LdsBuf<C>{ lds + (c + j) * fourstep_pitch(M1) }[k1], w[j][i]);
                C *d = out + (long long) k1 * M2 + n2;
                if (CV == 2) *reinterpret_cast<float4 *>(d) = make_float4((float) v[0].x, (float) v[0].y, (float) v[CV - 1].x, (float) v[CV - 1].y);
                else d[0] = v[0];
            }
        }
        else
        {
            for (int e = threadIdx.x; e < (COLS / CV) * M1; e += NT)
            {
                const int c = (e % (COLS / CV)) * CV, k1 = e / (COLS / CV);
                const int n2 = col0 + c;
                C v[CV];
#pragma unroll
                for (int j = 0; j < CV; j++) v[j] = cmul(LdsBuf<C>{ lds + (c + j) * fourstep_pitch(M1) }[k1], fx_twiddle(twN, 2 * (n2 + j) * k1, M));
                C *d = out + (long long) k1 * M2 + n2;
                if (CV == 2) *reinterpret_cast<float4 *>(d) = make_float4((float) v[0].x, (float) v[0].y, (float) v[CV - 1].x, (float) v[CV - 1].y);
                else d[0] = v[0];
            }
        }
        FX_MARK(2);
        FX_COUNT(3);
    }

    // rows: for every k1 an M2-point transform over n2; element k2 of row k1 is bin k1 + M1*k2
    template <class T, int L2>
    __global__ __launch_bounds__((FxTile<(1 << L2), (int) sizeof(typename Cx<T>::type)>::THREADS)) void fx_rows_kernel(const typename Cx<T>::type *__restrict__ work, FxK<T> a0, typename Cx<T>::type *__restrict__ post,
                                                          int M1, int M, long long q0, const typename Cx<T>::type *__restrict__ tw2)
    {
        typedef typename Cx<T>::type C;
        constexpr int M2 = 1 << L2;
        typedef FxTile<M2, (int) sizeof(C)> Tile;
        constexpr int TG = Tile::TG, G = Tile::G, ROWS = Tile::TILE, NT = Tile::THREADS;
        extern __shared__ __attribute__((aligned(16))) unsigned char fx_raw[];
        C *lds = reinterpret_cast<C *>(fx_raw);                                // [ROWS][M2]

        FX_T0
        const int row0 = fourstep_tile_of(blockIdx.x, gridDim.x) * ROWS;
        const C *in = work + (long long) blockIdx.y * M + (long long) row0 * M2;
        constexpr int CV = 16 / (int) sizeof(C) > 0 ? 16 / (int) sizeof(C) : 1;
        if (CV == 2)
        {
            // (the tile's rows are contiguous in the scratch: two complex values per 16-byte load)
            for (int e = threadIdx.x; e < ROWS * M2 / 2; e += NT)
            {
                const float4 v = reinterpret_cast<const float4 *>(in)[e];
                const int row = (2 * e) / M2, n2 = (2 * e) % M2;
                LdsBuf<C> b = { lds + row * fourstep_pitch(M2) };
                b[n2] = C(v.x, v.y);
                b[n2 + 1] = C(v.z, v.w);
            }
        }
        else
            for (int e = threadIdx.x; e < ROWS * M2; e += NT) LdsBuf<C>{ lds + (e / M2) * fourstep_pitch(M2) }[e % M2] = in[e];
        __syncthreads();
        FX_MARK(4);
        const int g = threadIdx.x / TG, t = threadIdx.x % TG;
        for (int r0 = 0; r0 < ROWS; r0 += G) LdsFFT<L2, TG, C>::run(LdsBuf<C>{ lds + (r0 + g) * fourstep_pitch(M2) }, t, tw2);
        FX_MARK(5);
        const FxK<T> a = fx_at(a0, q0 + blockIdx.y);
        typedef typename FxVec<T>::type VT;
        constexpr int V = FxVec<T>::V;
        if (a0.store == S_SPLIT && ROWS % V == 0 && fx_aligned16(a.da) && fx_aligned16(a.db))                              // (wave-uniform)
        {
            for (int e = threadIdx.x; e < (ROWS / V) * M2; e += NT)
            {
                const int r = (e % (ROWS / V)) * V, k2 = e / (ROWS / V);
                const long long k = row0 + r + (long long) M1 * k2;
                VT vr, vi;
#pragma unroll
                for (int j = 0; j < V; j++)
                {
                    const C v = LdsBuf<C>{ lds + (r + j) * fourstep_pitch(M2) }[k2];
                    fx_put(vr, j, a.swap_out ? v.y : v.x);
                    fx_put(vi, j, a.swap_out ? v.x : v.y);
                }
                *reinterpret_cast<VT *>(a.da + k) = vr;
                *reinterpret_cast<VT *>(a.db + k) = vi;
            }
        }
        else
        {
            for (int e = threadIdx.x; e < ROWS * M2; e += NT)
            {
                const int r = e % ROWS, k2 = e / ROWS;
                const int k = row0 + r + M1 * k2;
                const C v = LdsBuf<C>{ lds + r * fourstep_pitch(M2) }[k2];
                if (a0.store == S_POST) post[(long long) blockIdx.y * M + k] = v;
                else fx_store<T, C>(a, 0, k, v);
            }
        }
        FX_MARK(6);
        FX_COUNT(7);
    }

    // -------------------------------------------------------------------------------------------- four-step passes, register tiles
    //
    // The tile of a pass — 16384 complex values: LN lines of P = 2^L points — held in the REGISTERS of a 1024-thread workgroup
    // (sixteen values per thread) from its global loads to its global stores; LDS is only the medium of the one or two exchanges
    // between the radix-16 / radix-16 / radix-(P / 256) passes.  The workgroup is persistent and software-pipelined: the NEXT
    // tile's sixteen loads per thread are issued into a second register set before the current tile is transformed, and the
    // current tile's stores drain behind it — what the LDS-staged passes above do strictly in sequence per tile (load 8 us,
    // transform 8 us, store 4.5 us for 128 KiB) here overlaps within the one workgroup a CU holds.
    //
    // A thread's elements: first pass n = t + TG r (r < 16, TG = P / 16 threads per line), last pass k = t' + TG m.  The two thread
    // maps — line-fast (line = tid % LN: adjacent lanes on adjacent lines, for the strided side of a pass) and line-slow
    // (t = tid % TG: adjacent lanes on adjacent elements of a line, for the contiguous side) — are chosen per pass; the exchange
    // through LDS makes the change of map free.
    template <int L> struct RegTile
    {
        static constexpr int P = 1 << L, TG = P / 16, LN = 1024 / TG;
        // two sixteen-element "virtual threads" (vtid = tid + NT h) per thread: 512 threads with 256 registers each hold the two
        // tiles (64 + 64 registers) and leave the butterflies their working set; 1024 threads with 128 spilled 18 - 62 of them
        static constexpr int VT = 2, NT = 1024 / VT;
        static constexpr int R3 = P / 256;                              // radix of the third pass (1: none)
        // line pitch in LDS (complex elements): the padded line plus what makes lanes on adjacent lines with 64 / LN consecutive
        // elements each land on 64 different 8-byte slots (two wavefront halves per bank pair: the floor of an 8-byte access)
        static constexpr int PITCH = P + P / 16 + (TG / 16 > 1 ? TG / 16 : 1);
        static constexpr size_t LDS_BYTES = sizeof(float2) * ((size_t) LN * PITCH + P);      // the tile + the P-th roots
        static_assert(L >= 8 && L <= 10, "register tiles: 256-, 512- and 1024-point lines");
    };

    __device__ __forceinline__ int rt_pad(int i) { return i + (i >> 4); }

    // first pass: sixteen-point transforms of u[r] = x[t + TG r], bins to positions 16 t + q of the line
    template <int L> __device__ __forceinline__ void rt_pass_a(float2 *u, float2 *line, int t)
    {
        dft16<true>(u, float2(), float2(), float2());
        float2 *bp = line + 17 * t;                                      // rt_pad(16 t + q) = 17 t + q
#pragma unroll
        for (int q2 = 0; q2 < 4; q2++)
#pragma unroll
            for (int q1 = 0; q1 < 4; q1++) bp[q1 + 4 * q2] = u[4 * q1 + q2];
    }

    template <int L> __device__ __forceinline__ void rt_read_b(float2 *u, const float2 *line, int t)
    {
        typedef RegTile<L> G;
        const float2 *bp = line + rt_pad(t);
#pragma unroll
        for (int r = 0; r < 16; r++) u[r] = bp[r * (G::TG + G::TG / 16)];
    }

    // second pass (p = 16): twiddles exp(-2 pi i k r / 256), k = t & 15, from the P-th roots in LDS.  LAST: the results stay in
    // u, u[m] = bin t + TG m (P = 256); else they go to positions j + 16 q, j = (t >> 4) * 256 + k
    template <int L, bool LAST> __device__ __forceinline__ void rt_pass_b(float2 *u, float2 *line, const float2 *rootP, int t)
    {
        typedef RegTile<L> G;
        const int k = t & 15, step = k * (G::P / 256);
        const float2 w1 = rootP[step], w2 = rootP[2 * step], w3 = rootP[3 * step], w4 = rootP[4 * step], w8 = rootP[8 * step], w12 = rootP[12 * step];
#pragma unroll
        for (int c = 0; c < 4; c++)
        {
            u[4 + c] = cmul(u[4 + c], w4);
            u[8 + c] = cmul(u[8 + c], w8);
            u[12 + c] = cmul(u[12 + c], w12);
        }
        dft16<false>(u, w1, w2, w3);
        if constexpr (LAST)
        {
            float2 v[16];
#pragma unroll
            for (int q2 = 0; q2 < 4; q2++)
#pragma unroll
                for (int q1 = 0; q1 < 4; q1++) v[q1 + 4 * q2] = u[4 * q1 + q2];
#pragma unroll
            for (int m = 0; m < 16; m++) u[m] = v[m];
        }
        else
        {
            const int j = ((t - k) << 4) + k;
            float2 *bp = line + rt_pad(j);                                  // rt_pad(j + 16 q) = rt_pad(j) + 17 q
#pragma unroll
            for (int q2 = 0; q2 < 4; q2++)
#pragma unroll
                for (int q1 = 0; q1 < 4; q1++) bp[17 * (q1 + 4 * q2)] = u[4 * q1 + q2];
        }
    }

    // third pass (p = 256, radix R3 = P / 256): butterflies i = t + TG b, b < 16 / R3, inputs at i + 256 r, twiddle exp(-2 pi i i r / P);
    // on return u[m] = bin t + TG m
    template <int L> __device__ __forceinline__ void rt_pass_c(float2 *u, const float2 *line, const float2 *rootP, int t)
    {
        typedef RegTile<L> G;
        constexpr int R3 = G::R3 > 1 ? G::R3 : 2, NBT = 16 / R3;
        float2 x[NBT][R3];
        const float2 *bp = line + rt_pad(t);
#pragma unroll
        for (int b = 0; b < NBT; b++)
#pragma unroll
            for (int r = 0; r < R3; r++) x[b][r] = bp[b * (G::TG + G::TG / 16) + r * 272];
#pragma unroll
        for (int b = 0; b < NBT; b++)
        {
            const int i = t + b * G::TG;
            if constexpr (R3 == 4)
            {
                x[b][1] = cmul(x[b][1], rootP[i]);
                x[b][2] = cmul(x[b][2], rootP[2 * i]);
                x[b][3] = cmul(x[b][3], rootP[3 * i]);
                radix4(x[b][0], x[b][1], x[b][2], x[b][3]);
            }
            else
            {
                const float2 o = cmul(x[b][1], rootP[i]), e = x[b][0];
                x[b][0] = make_float2(e.x + o.x, e.y + o.y);
                x[b][1] = make_float2(e.x - o.x, e.y - o.y);
            }
#pragma unroll
            for (int r = 0; r < R3; r++) u[b + NBT * r] = x[b][r];          // bin i + 256 r = t + TG (b + (256 / TG) r)
        }
    }

    // The thread maps of a pass, for virtual thread h of thread tid: line-fast (FAST: adjacent lanes on adjacent PAIRS of lines, the
    // thread's two virtual threads on the two lines of a pair — the strided side of a pass: one 8- or 16-byte access covers both)
    // or line-slow (adjacent lanes on adjacent PAIRS of elements of a line, the two virtual threads on the two elements — the
    // contiguous side: one 16-byte access covers both)
    template <int L, bool FAST> struct RtMap
    {
        typedef RegTile<L> G;
        static_assert(G::VT == 2, "pairs");
        int line0, t0;
        __device__ __forceinline__ explicit RtMap(int tid) : line0(FAST ? 2 * (tid % (G::LN / 2)) : tid / (G::TG / 2)), t0(FAST ? tid / (G::LN / 2) : 2 * (tid % (G::TG / 2))) {}
        __device__ __forceinline__ int line(int h) const { return FAST ? line0 + h : line0; }
        __device__ __forceinline__ int t(int h) const { return FAST ? t0 : t0 + h; }
    };

    // the transform of the tile in u (first-pass map A) to u (last-pass map B)
    template <int L, class MapA, class MapB> __device__ __forceinline__ void rt_transform(float2 (*u)[16], float2 *lds, const float2 *rootP, const MapA &ma, const MapB &mb)
    {
        typedef RegTile<L> G;
#pragma unroll
        for (int h = 0; h < G::VT; h++) rt_pass_a<L>(u[h], lds + ma.line(h) * G::PITCH, ma.t(h));
        __syncthreads();
        if constexpr (G::R3 == 1)
        {
#pragma unroll
            for (int h = 0; h < G::VT; h++) rt_read_b<L>(u[h], lds + mb.line(h) * G::PITCH, mb.t(h));
            __syncthreads();                                                // (the next tile's first pass writes the lines again)
#pragma unroll
            for (int h = 0; h < G::VT; h++) rt_pass_b<L, true>(u[h], nullptr, rootP, mb.t(h));
        }
        else
        {
#pragma unroll
            for (int h = 0; h < G::VT; h++) rt_read_b<L>(u[h], lds + ma.line(h) * G::PITCH, ma.t(h));
            __syncthreads();
#pragma unroll
            for (int h = 0; h < G::VT; h++) rt_pass_b<L, false>(u[h], lds + ma.line(h) * G::PITCH, rootP, ma.t(h));
            __syncthreads();
#pragma unroll
            for (int h = 0; h < G::VT; h++) rt_pass_c<L>(u[h], lds + mb.line(h) * G::PITCH, rootP, mb.t(h));
            __syncthreads();
        }
    }

    // the P-th roots exp(-2 pi i m / P), m < P, from the table of the 2P-th roots' first half
    template <int L> __device__ __forceinline__ void rt_roots(float2 *rootP, const float2 *__restrict__ tw2P)
    {
        typedef RegTile<L> G;
        for (int m = threadIdx.x; m < G::P; m += G::NT)
        {
            const float2 w = tw2P[(2 * m) & (G::P - 1)];
            rootP[m] = m < G::P / 2 ? w : make_float2(-w.x, -w.y);
        }
    }

    // tile `it` of persistent workgroup w of `wgs`: the workgroups an XCD runs at one time (w % 8 equal) take NEIGHBOURING tiles,
    // so that the two halves of the 128-byte lines their strided runs share meet in that XCD's L2
    __device__ __forceinline__ int rt_tile_of(int w, int wgs, int it)
    {
        if ((wgs & 7) == 0) return (it * 8 + (w & 7)) * (wgs >> 3) + (w >> 3);
        return it * wgs + w;
    }

    // The loop of both passes is ONE basic block from the next tile's loads to the hand-over copy behind the current tile's
    // stores (the load and store kinds are template parameters, the last iteration loads a single cache line instead of
    // branching): the compiler's wait counts are then exact — the copy waits for the loads and leaves the stores in flight, the
    // first pass waits for nothing.  With a branch in between, its counts at the join are the merge of both ways in, and the
    // first butterflies waited for the loads just issued.

    // cols: for every n2 a P-point transform over n1 (P = M1), times W_M^(n2 k1), into the scratch at k1 M2 + n2
    // SPLIT_IN: split arrays whose transforms start on 8-byte boundaries (the two lines of a thread as one load)
    template <int L1, bool SPLIT_IN>
    __global__ __launch_bounds__(RegTile<L1>::NT) void fx_cols_tile_kernel(FxK<float> a0, float2 *__restrict__ work, int M2, int M, long long q0, int nb,
                                                                           const float2 *__restrict__ tw1, const float2 *__restrict__ twN)
    {
        typedef RegTile<L1> G;
        constexpr int TG = G::TG, LN = G::LN, VT = G::VT;
        extern __shared__ __attribute__((aligned(16))) unsigned char fx_raw[];
        float2 *lds = reinterpret_cast<float2 *>(fx_raw), *rootP = lds + LN * G::PITCH;
        rt_roots<L1>(rootP, tw1);
        const RtMap<L1, true> map((int) threadIdx.x);
        const int line = map.line(0), t = map.t(0);
        const int per = M2 / LN;                                            // tiles per transform
        const int total = per * nb, sh = __builtin_ctz(per);                 // (tiles per transform: a power of two; divisions would put branches into the loop)
        const unsigned lane_io = (unsigned) (t * M2 + line);
        // the second line's four-step twiddles from the first line's: W^((n2 + 1) k1) = W^(n2 k1) W^k1, k1 = t + TG m
        const float2 wt = fx_twiddle(twN, 2 * t, M);
        float2 cur[VT][16], nxt[VT][16];
        // (`lane`: lane_io, or 0 for the load that only keeps the last iteration's code straight)
        auto load = [&](float2 (*u)[16], int tile, unsigned lane)
        {
            const FxK<float> a = fx_at(a0, q0 + (tile >> sh));
            const int c0 = (tile & (per - 1)) * LN;
            if constexpr (SPLIT_IN)
            {
                // (a uniform pointer per element index and ONE 32-bit lane offset: sixteen 64-bit lane addresses per virtual
                // thread, loop invariants all, would be kept in registers beside the two tiles)
                const float *re = static_cast<const float *>(a.sa) + c0, *im = a.sb + c0;
#pragma unroll
                for (int r = 0; r < 16; r++)
                {
                    const long long idx = (long long) (TG * r) * M2;          // (uniform)
                    const float2 vr = *reinterpret_cast<const float2 *>(re + idx + lane), vi = *reinterpret_cast<const float2 *>(im + idx + lane);
                    u[0][r] = make_float2(vr.x, vi.x);
                    u[1][r] = make_float2(vr.y, vi.y);
                }
            }
            else
            {
#pragma unroll
                for (int h = 0; h < VT; h++)
#pragma unroll
                    for (int r = 0; r < 16; r++) u[h][r] = fx_load<float, float2>(a, 0, (TG * r) * M2 + c0 + (int) lane + h, M, twN);
            }
        };
        int tile = rt_tile_of(blockIdx.x, gridDim.x, 0);
        if (tile >= total) return;
        load(nxt, tile, lane_io);
#pragma unroll
        for (int h = 0; h < VT; h++)
#pragma unroll
            for (int r = 0; r < 16; r++) cur[h][r] = nxt[h][r];
        __builtin_amdgcn_s_waitcnt(0);                                      // (nothing in flight on the way into the loop: see above)
        __syncthreads();                                                    // (the roots)
        for (int it = 0; tile < total; it++)
        {
            const int next = rt_tile_of(blockIdx.x, gridDim.x, it + 1);
            const bool more = next < total;
            load(nxt, more ? next : tile, more ? lane_io : 0u);
            rt_transform<L1>(cur, lds, rootP, map, map);
            // the four-step twiddles W^(n2 (t + TG m)) = W^(n2 t) S^m, S = W^(n2 TG): the first and S, S^2, S^4, S^8 each from
            // sincospi of an exactly reduced argument, applied to the values factor by factor (a table of the sixteen products
            // would be thirty-two registers beside the two tiles): at most six factors per value, each within 2 ulp
            const int c0 = (tile & (per - 1)) * LN, n2 = c0 + line;
            {
                const float2 w0 = fx_twiddle(twN, 2 * n2 * t, M), w1 = cmul(w0, wt);
#pragma unroll
                for (int m = 0; m < 16; m++)
                {
                    cur[0][m] = cmul(cur[0][m], w0);
                    cur[1][m] = cmul(cur[1][m], w1);
                }
            }
#pragma unroll
            for (int bit = 1; bit < 16; bit *= 2)
            {
                const float2 s0 = fx_twiddle(twN, 2 * n2 * TG * bit, M), s1 = cmul(s0, fx_twiddle(twN, 2 * TG * bit, M));
#pragma unroll
                for (int m = 0; m < 16; m++)
                    if (m & bit)
                    {
                        cur[0][m] = cmul(cur[0][m], s0);
                        cur[1][m] = cmul(cur[1][m], s1);
                    }
            }
            float2 *out = work + (tile >> sh) * (long long) M + c0;
#pragma unroll
            for (int m = 0; m < 16; m++)
                *reinterpret_cast<float4 *>(out + (long long) (TG * m) * M2 + lane_io) = make_float4(cur[0][m].x, cur[0][m].y, cur[1][m].x, cur[1][m].y);
#pragma unroll
            for (int h = 0; h < VT; h++)
#pragma unroll
                for (int r = 0; r < 16; r++) cur[h][r] = nxt[h][r];
            tile = next;
        }
    }

    // rows: for every k1 a P-point transform over n2 (P = M2); element k2 of row k1 is bin k1 + M1 k2
    // STORE: S_SPLIT = split arrays whose transforms start on 8-byte boundaries, S_POST = the scratch of the real post pass, else any
    template <int L2, int STORE>
    __global__ __launch_bounds__(RegTile<L2>::NT) void fx_rows_tile_kernel(const float2 *__restrict__ work, FxK<float> a0, float2 *__restrict__ post, int M1, int M,
                                                                           long long q0, int nb, const float2 *__restrict__ tw2)
    {
        typedef RegTile<L2> G;
        constexpr int TG = G::TG, LN = G::LN, M2 = G::P, VT = G::VT;
        extern __shared__ __attribute__((aligned(16))) unsigned char fx_raw[];
        float2 *lds = reinterpret_cast<float2 *>(fx_raw), *rootP = lds + LN * G::PITCH;
        rt_roots<L2>(rootP, tw2);
        const RtMap<L2, false> ma((int) threadIdx.x);                        // loads: adjacent lanes on adjacent elements of a row
        const RtMap<L2, true> mb((int) threadIdx.x);                         // stores: adjacent lanes on adjacent rows (adjacent bins)
        const int per = M1 / LN;
        const int total = per * nb, sh = __builtin_ctz(per);
        const unsigned lane_in = (unsigned) (ma.line(0) * M2 + ma.t(0)), lane_out = (unsigned) (mb.line(0) + M1 * mb.t(0));
        float2 cur[VT][16], nxt[VT][16];
        auto load = [&](float2 (*u)[16], int tile, unsigned lane)
        {
            const float2 *in = work + (tile >> sh) * (long long) M + ((tile & (per - 1)) * LN) * (long long) M2;
#pragma unroll
            for (int r = 0; r < 16; r++)
            {
                const float4 v = *reinterpret_cast<const float4 *>(in + TG * r + lane);
                u[0][r] = make_float2(v.x, v.y);
                u[1][r] = make_float2(v.z, v.w);
            }
        };
        int tile = rt_tile_of(blockIdx.x, gridDim.x, 0);
        if (tile >= total) return;
        load(nxt, tile, lane_in);
#pragma unroll
        for (int h = 0; h < VT; h++)
#pragma unroll
            for (int r = 0; r < 16; r++) cur[h][r] = nxt[h][r];
        __builtin_amdgcn_s_waitcnt(0);                                      // (nothing in flight on the way into the loop: see above)
        __syncthreads();
        for (int it = 0; tile < total; it++)
        {
            const int next = rt_tile_of(blockIdx.x, gridDim.x, it + 1);
            const bool more = next < total;
            load(nxt, more ? next : tile, more ? lane_in : 0u);
            rt_transform<L2>(cur, lds, rootP, ma, mb);
            const FxK<float> a = fx_at(a0, q0 + (tile >> sh));
            const int k0 = (tile & (per - 1)) * LN;                         // (uniform)
            if constexpr (STORE == S_SPLIT)
            {
                float *re = (a.swap_out ? a.db : a.da) + k0, *im = (a.swap_out ? a.da : a.db) + k0;
#pragma unroll
                for (int m = 0; m < 16; m++)
                {
                    const long long k = (long long) M1 * (TG * m);          // (uniform)
                    *reinterpret_cast<float2 *>(re + k + lane_out) = make_float2(cur[0][m].x, cur[1][m].x);
                    *reinterpret_cast<float2 *>(im + k + lane_out) = make_float2(cur[0][m].y, cur[1][m].y);
                }
            }
            else if constexpr (STORE == S_POST)
            {
                float2 *out = post + (tile >> sh) * (long long) M + k0;
#pragma unroll
                for (int m = 0; m < 16; m++)
                    *reinterpret_cast<float4 *>(out + (long long) M1 * (TG * m) + lane_out) = make_float4(cur[0][m].x, cur[0][m].y, cur[1][m].x, cur[1][m].y);
            }
            else
            {
#pragma unroll
                for (int h = 0; h < VT; h++)
#pragma unroll
                    for (int m = 0; m < 16; m++) fx_store<float, float2>(a, 0, k0 + mb.line(h) + M1 * (mb.t(h) + TG * m), cur[h][m]);
            }
#pragma unroll
            for (int h = 0; h < VT; h++)
#pragma unroll
                for (int r = 0; r < 16; r++) cur[h][r] = nxt[h][r];
            tile = next;
        }
    }

    // -------------------------------------------------------------------------------------------- four-step, both passes in one launch
    //
    // A batch of four-step transforms as ONE launch of persistent workgroups (one per CU), each taking tickets from a device counter.
    // Ticket t = the column tile t of the batch AND the row tile t - D: the row pass of a transform follows its column pass at a
    // distance of D tiles, so the scratch between them is a RING of a few transforms that never leaves the Infinity Cache, neither
    // pass has launch ramps of its own (two launches per 64 MiB chunk spent a third of their time filling and draining 256 CUs with
    // two tiles each), and every workgroup alternates column tile, row tile, ... in one straight loop, the next tile's loads in
    // flight while the current one is transformed (register tiles above).
    //
    // Hand-overs inside the launch: the scratch is written and read with agent-scope 8-byte accesses; per ring slot one counter of
    // finished column tiles and one of finished row tiles.  A row tile waits until its transform's column tiles are all counted;
    // a column tile waits, before it stores, until the row tiles of the transform that used its slot RING transforms earlier are.
    // Both are counted only when the workgroup's stores have completed (a tile later: the wait is free by then), both are polled a
    // tile ahead by one lane, and D and RING are chosen so that in the ordinary case nothing ever waits.
    // Forward progress by construction: tickets are handed out in order to workgroups that are RUNNING, and a tile depends only on
    // tiles with smaller tickets — whoever holds those is on the chip and waits, in turn, only for still smaller ones.
    struct FxStreamCtl
    {
        unsigned ticket, exited, pad[2];
        unsigned cols_done[kFxStreamRing], rows_done[kFxStreamRing];
    };

    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    constexpr int kWaitVm32 = 0x8F70, kWaitVm16 = 0x4F70;      // s_waitcnt vmcnt(32) / vmcnt(16) alone (gfx9 encoding: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14)
    __device__ __forceinline__ unsigned fx_poll(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

    template <int L1, int L2>
    __global__ __launch_bounds__(RegTile<L1>::NT) void fx_stream_kernel(FxK<float> a0, float2 *__restrict__ ring, FxStreamCtl *__restrict__ ctl, int M, int nb, int D, int RING,
                                                                        const float2 *__restrict__ tw1, const float2 *__restrict__ tw2, const float2 *__restrict__ twN)
    {
        typedef RegTile<L1> G1;
        typedef RegTile<L2> G2;
        constexpr int VT = G1::VT, NT = G1::NT, M1 = G1::P, M2 = G2::P;
        constexpr int TILE_ELEMS = (G1::LN * G1::PITCH > G2::LN * G2::PITCH) ? G1::LN * G1::PITCH : G2::LN * G2::PITCH;
        extern __shared__ __attribute__((aligned(16))) unsigned char fx_raw[];
        float2 *lds = reinterpret_cast<float2 *>(fx_raw), *root1 = lds + TILE_ELEMS, *root2 = (L1 == L2) ? root1 : root1 + M1;
        int *slots = reinterpret_cast<int *>(root2 + M2);                    // [0] ticket after next, [1] early polls, [2] spin result
        rt_roots<L1>(root1, tw1);
        if (L1 != L2) rt_roots<L2>(root2, tw2);
        const int tid = (int) threadIdx.x;
        const int per = M >> 14, sh = __builtin_ctz(per);                    // tiles per transform and pass (16384 elements each)
        const int T = nb * per, END = T + D;
        const __amdgpu_buffer_rsrc_t ring_rsrc = __builtin_amdgcn_make_buffer_rsrc(ring, 0, (int) ((long long) (RING + 1) * M * sizeof(float2)), 0x00020000);
        float2 *dump = ring + (long long) RING * M;                          // where the halves of tickets without a tile store
        float2 cur[VT][16], nxt[VT][16];

        auto load_cols = [&](float2 (*u)[16], int ct, bool valid, unsigned lane_c)
        {
            const int q = valid ? ct >> sh : 0, c0 = valid ? (ct & (per - 1)) * G1::LN : 0;
            const unsigned lane = valid ? lane_c : 0u;
            const FxK<float> a = fx_at(a0, q);
            const float *re = static_cast<const float *>(a.sa) + c0, *im = a.sb + c0;
#pragma unroll
            for (int r = 0; r < 16; r++)
            {
                const long long idx = (long long) (G1::TG * r) * M2;         // (uniform)
                // (streamed once: not kept in the caches the ring lives in)
                const v2f vr = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(re + idx + lane)), vi = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(im + idx + lane));
                u[0][r] = make_float2(vr.x, vi.x);
                u[1][r] = make_float2(vr.y, vi.y);
            }
        };
        auto load_rows = [&](float2 (*u)[16], int rt, bool valid, unsigned lane_ri)
        {
            const int q = valid ? rt >> sh : 0, r0 = valid ? (rt & (per - 1)) * G2::LN : 0;
            const unsigned lane = valid ? lane_ri : 0u;
            const unsigned base = (unsigned) (((long long) (q % RING) * M + (long long) r0 * M2) * sizeof(float2));
#pragma unroll
            for (int r = 0; r < 16; r++)
            {
                const v4u v = __builtin_amdgcn_raw_buffer_load_b128(ring_rsrc, lane * (unsigned) sizeof(float2), base + (unsigned) (G2::TG * r * sizeof(float2)), /* sc1 */ 16);
                u[0][r] = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
                u[1][r] = make_float2(__uint_as_float(v.z), __uint_as_float(v.w));
            }
        };
        // one lane's view of the two conditions of ticket t (bit 0: its row tile may be loaded, bit 1: its column tile may be stored),
        // in two halves: the counters are read now, looked at later (a tile later for the early poll: nothing waits for the reads)
        struct Polled { unsigned cols, rows; };
        auto poll_issue = [&](int t) -> Polled
        {
            Polled p = { 0u, 0u };
            const int rt = t - D;
            if (rt >= 0 && rt < T) p.cols = fx_poll(&ctl->cols_done[(rt >> sh) % RING]);
            if (t < T) p.rows = fx_poll(&ctl->rows_done[(t >> sh) % RING]);
            return p;
        };
        auto poll_eval = [&](int t, Polled p) -> int
        {
            int ok = 0;
            const int rt = t - D;
            if (rt < 0 || rt >= T || (int) (p.cols - (unsigned) (((rt >> sh) / RING + 1) * per)) >= 0) ok |= 1;
            if (t >= T || (int) (p.rows - (unsigned) (((t >> sh) / RING) * per)) >= 0) ok |= 2;
            return ok;
        };
        auto poll = [&](int t) -> int { return poll_eval(t, poll_issue(t)); };
        // Before a workgroup starts to wait it counts in the row tile it has finished and not yet counted (normally counted a tile
        // later, when its stores have completed for free): a tile's count must never depend on its workgroup getting past a wait of
        // a LATER ticket — that later ticket may itself be waiting for the workgroup that needs this count.  With that, a waiting
        // workgroup has published everything below its ticket, and the lowest waiting ticket depends only on running workgroups.
        // (A wait that outlasts a million polls — seconds — is a fault of the scheme, never of the load: trap rather than hang.)
        auto wait_for = [&](int t, int have, int bit, int &unpublished_rows_slot)
        {
            if (have & bit) return;
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (tid == 0 && unpublished_rows_slot >= 0) __hip_atomic_fetch_add(&ctl->rows_done[unpublished_rows_slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unpublished_rows_slot = -1;
            for (int spins = 0; !(have & bit); spins++)
            {
                if (spins >= (1 << 20)) __builtin_trap();
                if (tid == 0)
                {
                    if (spins) __builtin_amdgcn_s_sleep(8);
                    slots[2] = poll(t);
                }
                __syncthreads();
                have = __builtin_amdgcn_readfirstlane(slots[2]);
                __syncthreads();
            }
        };

        if (tid == 0)
        {
            slots[0] = (int) __hip_atomic_fetch_add(&ctl->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            slots[1] = (int) __hip_atomic_fetch_add(&ctl->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();                                                    // (and the roots)
        // (what the workgroup reads back from LDS is uniform, and the compiler has to be told: every tile address hangs off these)
        int tk = __builtin_amdgcn_readfirstlane(slots[0]), tk1 = __builtin_amdgcn_readfirstlane(slots[1]), flags = 0, prev_rows_slot = -1;
        __syncthreads();
        if (tk < END)
        {
            const RtMap<L1, true> mc0(tid);
            load_cols(nxt, tk, tk < T, (unsigned) (mc0.t(0) * M2 + mc0.line(0)));
#pragma unroll
            for (int h = 0; h < VT; h++)
#pragma unroll
                for (int r = 0; r < 16; r++) cur[h][r] = nxt[h][r];
            __builtin_amdgcn_s_waitcnt(0);                                  // (nothing in flight on the way into the loop: see the passes above)
        }
        while (tk < END)
        {
            // (the thread maps and everything derived from them — sixteen LDS and lane addresses — are worked out again every tile
            // from a copy of the thread index the compiler cannot see through: hoisted out of the loop they were the registers
            // that spilled, and a spilled register's reload waits for every load in flight)
            int tid_now = tid;
            asm volatile("" : "+v"(tid_now));
            const RtMap<L1, true> mc(tid_now);                               // column tiles: adjacent lanes on adjacent columns, in and out
            const RtMap<L2, false> ma(tid_now);                              // row tiles in: adjacent lanes on adjacent elements of a row
            const RtMap<L2, true> mb(tid_now);                               // row tiles out: adjacent lanes on adjacent rows (bins)
            const unsigned lane_c = (unsigned) (mc.t(0) * M2 + mc.line(0));
            const unsigned lane_ri = (unsigned) (ma.line(0) * M2 + ma.t(0)), lane_ro = (unsigned) (mb.line(0) + M1 * mb.t(0));
            // ---- the column tile of this ticket; its row tile's loads go first
            // (the ticket after next, used a tile from now.  As an asm statement: the compiler counts a returning atomic of its own as
            // in flight at the loop header — whatever waits the loop holds — and drains every store before the register is written
            // again; the explicit vmcnt(32) in front of the last barrier below is this statement's wait)
            unsigned tk2 = 0;
            if (tid == 0) asm volatile("global_atomic_add %0, %1, %2, %3 sc0" : "=v"(tk2) : "v"(0u), "v"(1u), "s"(&ctl->ticket) : "memory");
            const int rt = tk - D;
            const bool cols_valid = tk < T, rows_valid = rt >= 0 && rt < T;
            wait_for(tk, flags, 1, prev_rows_slot);
            load_rows(nxt, rt, rows_valid, lane_ri);
            rt_transform<L1>(cur, lds, root1, mc, mc);
            {
                // the four-step twiddles W^(n2 (t + TG m)), as in the column pass above
                const int c0 = (tk & (per - 1)) * G1::LN, n2 = c0 + mc.line(0);
                {
                    // (the second line's factors from sincospi as well: kept as products of the first line's, their constants were ten
                    // registers held across the loop — beside the two tiles that is what spilled)
                    const float2 w0 = fx_twiddle(twN, 2 * n2 * mc.t(0), M), w1 = fx_twiddle(twN, 2 * (n2 + 1) * mc.t(0), M);
#pragma unroll
                    for (int m = 0; m < 16; m++)
                    {
                        cur[0][m] = cmul(cur[0][m], w0);
                        cur[1][m] = cmul(cur[1][m], w1);
                    }
                }
#pragma unroll
                for (int bit = 1; bit < 16; bit *= 2)
                {
                    const float2 s0 = fx_twiddle(twN, 2 * n2 * G1::TG * bit, M), s1 = fx_twiddle(twN, 2 * (n2 + 1) * G1::TG * bit, M);
#pragma unroll
                    for (int m = 0; m < 16; m++)
                        if (m & bit)
                        {
                            cur[0][m] = cmul(cur[0][m], s0);
                            cur[1][m] = cmul(cur[1][m], s1);
                        }
                }
                wait_for(tk, flags, 2, prev_rows_slot);
                // (16-byte write-through stores through a buffer descriptor — an agent-scope atomic store is 8 bytes at most, and
                // 8-byte write-through stores are one fabric write each: 2.7 times the time per byte)
                const int q = tk >> sh;
                const unsigned base = cols_valid ? (unsigned) (((long long) (q % RING) * M + c0) * sizeof(float2)) : (unsigned) ((long long) RING * M * sizeof(float2));
#pragma unroll
                for (int m = 0; m < 16; m++)
                {
                    const v4u v = { __float_as_uint(cur[0][m].x), __float_as_uint(cur[0][m].y), __float_as_uint(cur[1][m].x), __float_as_uint(cur[1][m].y) };
                    __builtin_amdgcn_raw_buffer_store_b128(v, ring_rsrc, lane_c * (unsigned) sizeof(float2), base + (unsigned) (G1::TG * m * M2 * sizeof(float2)), /* sc1 */ 16);
                }
            }
#pragma unroll
            for (int h = 0; h < VT; h++)
#pragma unroll
                for (int r = 0; r < 16; r++) cur[h][r] = nxt[h][r];
            // ---- the row tile; the next ticket's column tile's loads go first.  The previous row tile's stores have completed once
            // the loads issued behind them have (the wait is spelled out: the compiler may fold the copy above into the first
            // butterflies and wait only there; the 16 column stores issued since may stay in flight): count it
            __builtin_amdgcn_s_waitcnt(kWaitVm16);
            __syncthreads();
            Polled early = { 0u, 0u };
            if (tid == 0)
            {
                if (prev_rows_slot >= 0) __hip_atomic_fetch_add(&ctl->rows_done[prev_rows_slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (tk1 < END) early = poll_issue(tk1);                        // (looked at a tile from now)
            }
            load_cols(nxt, tk1, tk1 < T, lane_c);
            rt_transform<L2>(cur, lds, root2, ma, mb);
            {
                const int q = rows_valid ? rt >> sh : 0, k0 = rows_valid ? (rt & (per - 1)) * G2::LN : 0;
                const FxK<float> a = fx_at(a0, q);
                float *re = rows_valid ? (a.swap_out ? a.db : a.da) + k0 : reinterpret_cast<float *>(dump);
                float *im = rows_valid ? (a.swap_out ? a.da : a.db) + k0 : reinterpret_cast<float *>(dump) + M;
#pragma unroll
                for (int m = 0; m < 16; m++)
                {
                    const long long k = (long long) M1 * (G2::TG * m);      // (uniform)
                    __builtin_nontemporal_store(v2f{ cur[0][m].x, cur[1][m].x }, reinterpret_cast<v2f *>(re + k + lane_ro));
                    __builtin_nontemporal_store(v2f{ cur[0][m].y, cur[1][m].y }, reinterpret_cast<v2f *>(im + k + lane_ro));
                }
            }
#pragma unroll
            for (int h = 0; h < VT; h++)
#pragma unroll
                for (int r = 0; r < 16; r++) cur[h][r] = nxt[h][r];
            // ---- the column tile's stores have completed (issued before the loads this copy waits for; the 32 row stores since
            // may stay in flight): count it, and hand the ticket after next and the early polls to the workgroup
            __builtin_amdgcn_s_waitcnt(kWaitVm32);
            __syncthreads();
            // (the results of the lane's ticket and polls are taken up by EVERY lane, outside the branch: a result waited for only
            // inside a divergent branch still counts as in flight at the loop header, and the next tile's first write to its
            // register then waits for every store in flight)
            const int tk2_u = __builtin_amdgcn_readfirstlane((int) tk2);
            const Polled early_u = { (unsigned) __builtin_amdgcn_readfirstlane((int) early.cols), (unsigned) __builtin_amdgcn_readfirstlane((int) early.rows) };
            if (tid == 0)
            {
                if (cols_valid) __hip_atomic_fetch_add(&ctl->cols_done[(tk >> sh) % RING], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                slots[0] = tk2_u;
                slots[1] = tk1 < END ? poll_eval(tk1, early_u) : 3;
            }
            __syncthreads();
            prev_rows_slot = rows_valid ? (rt >> sh) % RING : -1;
            tk = tk1;
            tk1 = __builtin_amdgcn_readfirstlane(slots[0]);
            flags = __builtin_amdgcn_readfirstlane(slots[1]);
        }
        // the last row tile's stores, then out; the last workgroup out leaves the counters as it found them
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0)
        {
            if (prev_rows_slot >= 0) __hip_atomic_fetch_add(&ctl->rows_done[prev_rows_slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(&ctl->exited, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1)
            {
                for (int i = 0; i < kFxStreamRing; i++)
                {
                    __hip_atomic_store(&ctl->cols_done[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&ctl->rows_done[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __hip_atomic_store(&ctl->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ctl->exited, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }

    template <class T>
    __global__ void fx_post_kernel(const typename Cx<T>::type *__restrict__ Z, FxK<T> a, int M, long long q0, const typename Cx<T>::type *__restrict__ twN)
    {
        typedef typename Cx<T>::type C;
        const int k = blockIdx.x * blockDim.x + threadIdx.x;
        if (k > M / 2) return;
        const C *z = Z + (long long) blockIdx.y * M;
        fx_post<T, C>(fx_at(a, q0 + blockIdx.y), 0, k, M, z[k], z[(M - k) & (M - 1)], twN);
    }

    // -------------------------------------------------------------------------------------------- zip / unzip

    template <class T, class U>
    __global__ void fx_unzip_kernel(const U *__restrict__ in, T *__restrict__ re, T *__restrict__ im, long long half, long long in_len,
                                    long long sstride, long long dstride, long long batch)
    {
        const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
        if (k >= half) return;
        for (long long q = blockIdx.y; q < batch; q += gridDim.y)
        {
            const U *x = in + q * sstride;
            re[q * dstride + k] = (2 * k < in_len) ? (T) x[2 * k] : (T) 0;
            im[q * dstride + k] = (2 * k + 1 < in_len) ? (T) x[2 * k + 1] : (T) 0;
        }
    }

    template <class T>
    __global__ void fx_zip_kernel(const T *__restrict__ re, const T *__restrict__ im, T *__restrict__ out, long long half, long long sstride,
                                  long long dstride, long long batch)
    {
        const long long k = (long long) blockIdx.x * blockDim.x + threadIdx.x;
        if (k >= half) return;
        for (long long q = blockIdx.y; q < batch; q += gridDim.y)
        {
            T *o = out + q * dstride + 2 * k;
            o[0] = re[q * sstride + k];
            o[1] = im[q * sstride + k];
        }
    }

    // -------------------------------------------------------------------------------------------- host: tables and scratch

    std::mutex gFxMutex;
    std::map<std::pair<int, int>, double2 *> gTwF64;

    struct Scratch
    {
        void *a = nullptr, *b = nullptr;
        size_t bytes = 0;
    };
    std::map<int, Scratch> gScratch;

    const double2 *twiddles_f64(int device, int log2n, std::string *err)
    {
        std::lock_guard<std::mutex> g(gFxMutex);
        auto key = std::make_pair(device, log2n);
        auto it = gTwF64.find(key);
        if (it != gTwF64.end()) return it->second;
        const size_t half = size_t(1) << (log2n - 1);
        std::vector<double2> host(half);
        const double pi = 3.14159265358979323846264338327950288;
        for (size_t m = 0; m < half; m++)
        {
            const double angle = -(double) m * pi / (double) half;
            host[m] = make_double2(std::cos(angle), std::sin(angle));
        }
        double2 *dev = nullptr;
        hipError_t e = hipMalloc(&dev, half * sizeof(double2));
        if (e == hipSuccess) e = hipMemcpy(dev, host.data(), half * sizeof(double2), hipMemcpyHostToDevice);
        if (e != hipSuccess)
        {
            if (err) *err = std::string("twiddle table upload failed: ") + hipGetErrorString(e);
            if (dev) (void) hipFree(dev);
            return nullptr;
        }
        gTwF64[key] = dev;
        return dev;
    }

    template <class T> const typename Cx<T>::type *fx_twiddles(int device, int log2n, std::string *err);
    template <> const float2 *fx_twiddles<float>(int device, int log2n, std::string *err) { return twiddles(device, log2n, err); }
    template <> const double2 *fx_twiddles<double>(int device, int log2n, std::string *err) { return twiddles_f64(device, log2n, err); }

    // grow-only per-device scratch for the four-step path (two buffers of `bytes`)
    hipError_t scratch(int device, size_t bytes, Scratch &out)
    {
        std::lock_guard<std::mutex> g(gFxMutex);
        Scratch &s = gScratch[device];
        if (s.bytes < bytes)
        {
            if (s.a) (void) hipFree(s.a);                           // hipFree waits for work that still uses the old buffers
            if (s.b) (void) hipFree(s.b);
            s = Scratch();
            hipError_t e = hipMalloc(&s.a, bytes);
            if (e == hipSuccess) e = hipMalloc(&s.b, bytes);
            if (e != hipSuccess)
            {
                if (s.a) (void) hipFree(s.a);
                s = Scratch();
                return e;
            }
            s.bytes = bytes;
        }
        out = s;
        return hipSuccess;
    }

    template <class K> hipError_t allow_big_lds(K kernel, size_t bytes)
    {
        if (bytes <= 64 * 1024) return hipSuccess;
        return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes);
    }

    template <class T> constexpr int max_lds_log2m() { return sizeof(T) == 4 ? 14 : 13; }

    // -------------------------------------------------------------------------------------------- host: launches

    template <class T, int L> hipError_t launch_lds(const FxK<T> &k, const typename Cx<T>::type *tw, hipStream_t st)
    {
        typedef typename Cx<T>::type C;
        typedef FFTGeom<L> Gm;
        const size_t lds = sizeof(C) * lds_padded(Gm::M) * Gm::G;
        // thread groups of up to 32 threads (2^9 points) are staged: 2^7 ... 2^9 3.2 - 4.0 -> 4.8 - 5.5 TB/s
        // (larger groups: the instantiation that stages the real inverse's pre-pass, for those launches only — it costs the
        // plain path registers)
        void (*kernel)(FxK<T>, const C *) = fx_lds_kernel<T, L, false>;
        // (not the 1024-thread groups, one 2^14-point transform per CU: 2.27 -> 1.9 - 2.3 TB/s with the two extra barriers)
        if (Gm::TG <= HCV_FX_STAGE_TG || (k.load == L_PRE && k.store != S_POST && Gm::TG < 1024)) kernel = fx_lds_kernel<T, L, true>;
        hipError_t e = allow_big_lds(kernel, lds);
        if (e != hipSuccess) return e;
        // lane addressing is 32-bit relative to the workgroup's first transform: absurd strides go one transform per launch
        const long long reach = (long long) (Gm::G - 1) * std::max(k.sstride, k.dstride) + 4LL * Gm::M;
        if (Gm::G > 1 && reach >= (1LL << 31))
        {
            for (long long q = 0; q < k.batch; q++)
            {
                FxK<T> one = k;
                const long long so = q * k.sstride, dof = q * k.dstride;
                one.sa = k.src_f32 ? static_cast<const void *>(static_cast<const float *>(k.sa) + so) : static_cast<const void *>(static_cast<const T *>(k.sa) + so);
                if (one.sb) one.sb += so;
                one.da += dof;
                if (one.db) one.db += dof;
                one.batch = 1;
                hipLaunchKernelGGL(kernel, dim3(1), dim3(Gm::THREADS), lds, st, one, tw);
            }
            return hipGetLastError();
        }
        const long long grid = (k.batch + Gm::G - 1) / Gm::G;
        hipLaunchKernelGGL(kernel, dim3((unsigned) grid), dim3(Gm::THREADS), lds, st, k, tw);
        return hipGetLastError();
    }

    template <class T> hipError_t dispatch_lds(int lm, const FxK<T> &k, const typename Cx<T>::type *tw, hipStream_t st)
    {
        switch (lm)
        {
#define FX_CASE(L) case L: return launch_lds<T, L>(k, tw, st);
            FX_CASE(2) FX_CASE(3) FX_CASE(4) FX_CASE(5) FX_CASE(6) FX_CASE(7) FX_CASE(8) FX_CASE(9) FX_CASE(10) FX_CASE(11) FX_CASE(12) FX_CASE(13)
#undef FX_CASE
            case 14:
                if constexpr (sizeof(T) == 4) return launch_lds<T, 14>(k, tw, st);
                return hipErrorInvalidValue;
            default: return hipErrorInvalidValue;
        }
    }

    template <class T, int L1> hipError_t launch_cols(const FxK<T> &k, typename Cx<T>::type *work, int M2, int M, long long q0, int nb,
                                                      const typename Cx<T>::type *tw1, const typename Cx<T>::type *twN, hipStream_t st)
    {
        typedef typename Cx<T>::type C;
        typedef FxTile<(1 << L1), (int) sizeof(C)> Tile;
        const size_t lds = sizeof(C) * Tile::TILE * (size_t) fourstep_pitch(1 << L1);
        hipError_t e = allow_big_lds(fx_cols_kernel<T, L1>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((fx_cols_kernel<T, L1>), dim3(M2 / Tile::TILE, nb), dim3(Tile::THREADS), lds, st, k, work, M2, M, q0, tw1, twN);
        return hipGetLastError();
    }

    template <class T, int L2> hipError_t launch_rows(const typename Cx<T>::type *work, const FxK<T> &k, typename Cx<T>::type *post, int M1, int M,
                                                      long long q0, int nb, const typename Cx<T>::type *tw2, hipStream_t st)
    {
        typedef typename Cx<T>::type C;
        typedef FxTile<(1 << L2), (int) sizeof(C)> Tile;
        const size_t lds = sizeof(C) * Tile::TILE * (size_t) fourstep_pitch(1 << L2);
        hipError_t e = allow_big_lds(fx_rows_kernel<T, L2>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((fx_rows_kernel<T, L2>), dim3(M1 / Tile::TILE, nb), dim3(Tile::THREADS), lds, st, work, k, post, M1, M, q0, tw2);
        return hipGetLastError();
    }

    // the register-tile passes (float, 256- to 1024-point lines): persistent workgroups, one per CU
    inline int fx_tile_wgs(int device, long long tiles)
    {
        static std::mutex m;
        static std::map<int, int> cus;
        std::lock_guard<std::mutex> g(m);
        auto it = cus.find(device);
        if (it == cus.end())
        {
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n <= 0) n = 256;
            it = cus.emplace(device, n).first;
        }
        return (int) std::min<long long>(tiles, it->second);
    }
    inline bool fx_tile_on()
    {
        static const bool on = !(std::getenv("HCV_FX_TILE") && std::atoi(std::getenv("HCV_FX_TILE")) == 0);
        return on;
    }

    template <int L1> hipError_t launch_cols_tile(int device, const FxK<float> &k, float2 *work, int M2, int M, long long q0, int nb, const float2 *tw1,
                                                  const float2 *twN, hipStream_t st)
    {
        typedef RegTile<L1> G;
        void (*kernel)(FxK<float>, float2 *, int, int, long long, int, const float2 *, const float2 *) =
            (k.load == L_SPLIT && (reinterpret_cast<uintptr_t>(k.sa) & 7) == 0 && (reinterpret_cast<uintptr_t>(k.sb) & 7) == 0 && k.sstride % 2 == 0)
                ? fx_cols_tile_kernel<L1, true> : fx_cols_tile_kernel<L1, false>;
        hipError_t e = allow_big_lds(kernel, G::LDS_BYTES);
        if (e != hipSuccess) return e;
        const long long tiles = (long long) (M2 / G::LN) * nb;
        hipLaunchKernelGGL(kernel, dim3(fx_tile_wgs(device, tiles)), dim3(G::NT), G::LDS_BYTES, st, k, work, M2, M, q0, nb, tw1, twN);
        return hipGetLastError();
    }
    template <int L2> hipError_t launch_rows_tile(int device, const float2 *work, const FxK<float> &k, float2 *post, int M1, int M, long long q0, int nb,
                                                  const float2 *tw2, hipStream_t st)
    {
        typedef RegTile<L2> G;
        void (*kernel)(const float2 *, FxK<float>, float2 *, int, int, long long, int, const float2 *) =
            (k.store == S_SPLIT && (reinterpret_cast<uintptr_t>(k.da) & 7) == 0 && (reinterpret_cast<uintptr_t>(k.db) & 7) == 0 && k.dstride % 2 == 0)
                ? fx_rows_tile_kernel<L2, S_SPLIT> : k.store == S_POST ? fx_rows_tile_kernel<L2, S_POST> : fx_rows_tile_kernel<L2, S_ZIP>;
        hipError_t e = allow_big_lds(kernel, G::LDS_BYTES);
        if (e != hipSuccess) return e;
        const long long tiles = (long long) (M1 / G::LN) * nb;
        hipLaunchKernelGGL(kernel, dim3(fx_tile_wgs(device, tiles)), dim3(G::NT), G::LDS_BYTES, st, work, k, post, M1, M, q0, nb, tw2);
        return hipGetLastError();
    }

    // the one-launch batch (fx_stream_kernel): complex float transforms of 2^16 ... 2^20 points between split arrays on 8-byte
    // boundaries, batches of at least four tiles per CU; the ring (and the dump slot behind it) and the counters live with the
    // device's scratch
    struct StreamBuf
    {
        void *ring = nullptr, *ctl = nullptr;
        size_t bytes = 0;
    };
    std::map<int, StreamBuf> gStream;

    inline bool fx_stream_on()
    {
        static const bool on = !(std::getenv("HCV_FX_STREAM") && std::atoi(std::getenv("HCV_FX_STREAM")) == 0);
        return on;
    }

    template <int L1, int L2> hipError_t launch_stream(int device, const FxK<float> &k, int lm, const float2 *tw1, const float2 *tw2, const float2 *twN, hipStream_t st)
    {
        typedef RegTile<L1> G1;
        typedef RegTile<L2> G2;
        const int M = 1 << lm, per = M >> 14, wgs = fx_tile_wgs(device, 1 << 30);
        // the row tile of a ticket lags its column tile by a transform plus what the chip holds in flight (LAG quarters of a
        // tile per workgroup); a slot is written again when the rows that read it are twice that far behind
        static const int lag4 = std::getenv("HCV_FX_LAG") ? std::atoi(std::getenv("HCV_FX_LAG")) : 4;       // (experiment)
        const int D = per + lag4 * wgs / 4, RING = 2 + (D + 2 * wgs + per - 1) / per;
        if (RING > kFxStreamRing) return hipErrorInvalidValue;
        const size_t bytes = sizeof(float2) * (size_t) M * (size_t) (RING + 1);
        StreamBuf b;
        {
            std::lock_guard<std::mutex> g(gFxMutex);
            StreamBuf &sb = gStream[device];
            if (!sb.ctl)
            {
                hipError_t e = hipMalloc(&sb.ctl, sizeof(FxStreamCtl));
                if (e == hipSuccess) e = hipMemset(sb.ctl, 0, sizeof(FxStreamCtl));
                if (e != hipSuccess) { sb.ctl = nullptr; return e; }
            }
            if (sb.bytes < bytes)
            {
                if (sb.ring) (void) hipFree(sb.ring);                       // (hipFree waits for work that still uses the old ring)
                sb.ring = nullptr;
                sb.bytes = 0;
                hipError_t e = hipMalloc(&sb.ring, bytes);
                if (e != hipSuccess) return e;
                sb.bytes = bytes;
            }
            b = sb;
        }
        constexpr size_t tile = (size_t) std::max(G1::LN * G1::PITCH, G2::LN * G2::PITCH);
        constexpr size_t lds = sizeof(float2) * (tile + G1::P + (L1 == L2 ? 0 : G2::P)) + 16;
        hipError_t e = allow_big_lds(fx_stream_kernel<L1, L2>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((fx_stream_kernel<L1, L2>), dim3(wgs), dim3(G1::NT), lds, st, k, static_cast<float2 *>(b.ring), static_cast<FxStreamCtl *>(b.ctl), M, (int) k.batch, D, RING,
                           tw1, tw2, twN);
        return hipGetLastError();
    }

    template <class T> bool fx_stream_fits(int device, int lm, const FxK<T> &k)
    {
        if constexpr (sizeof(T) != 4) return false;
        if (!fx_stream_on() || lm < 16 || lm > 20 || k.load != L_SPLIT || k.store != S_SPLIT) return false;
        const uintptr_t bits = reinterpret_cast<uintptr_t>(k.sa) | reinterpret_cast<uintptr_t>(k.sb) | reinterpret_cast<uintptr_t>(k.da) | reinterpret_cast<uintptr_t>(k.db);
        if ((bits & 7) || (k.sstride & 1) || (k.dstride & 1)) return false;
        const long long tiles = k.batch * (long long) ((1 << lm) >> 14);
        return tiles >= 4LL * fx_tile_wgs(device, 1 << 30) && tiles < (1LL << 30);
    }

    template <class T> hipError_t run_big(int device, int lm, const FxK<T> &k, hipStream_t st, std::string *err)
    {
        typedef typename Cx<T>::type C;
        const int l2 = (lm + 1) / 2, l1 = lm - l2;
        const int M = 1 << lm, M1 = 1 << l1, M2 = 1 << l2;
        const C *twN = fx_twiddles<T>(device, lm + 1, err);
        const C *tw1 = fx_twiddles<T>(device, l1 + 1, err);
        const C *tw2 = fx_twiddles<T>(device, l2 + 1, err);
        if (!twN || !tw1 || !tw2) return hipErrorOutOfMemory;
        if constexpr (sizeof(T) == 4)
            if (fx_stream_fits<T>(device, lm, k))
                switch (lm)
                {
                    case 16: return launch_stream<8, 8>(device, k, lm, tw1, tw2, twN, st);
                    case 17: return launch_stream<8, 9>(device, k, lm, tw1, tw2, twN, st);
                    case 18: return launch_stream<9, 9>(device, k, lm, tw1, tw2, twN, st);
                    case 19: return launch_stream<9, 10>(device, k, lm, tw1, tw2, twN, st);
                    default: return launch_stream<10, 10>(device, k, lm, tw1, tw2, twN, st);
                }
        // scratch for up to 64 MiB of transforms per pass
        const size_t per = sizeof(C) * (size_t) M;
        static const size_t chunk_mb = std::getenv("HCV_FX_CHUNK_MB") ? (size_t) std::atoll(std::getenv("HCV_FX_CHUNK_MB")) : 64;      // (experiment)
        const long long chunk = std::max<long long>(1, std::min<long long>(k.batch, (long long) ((chunk_mb << 20) / per)));
        Scratch s;
        hipError_t e = scratch(device, per * (size_t) chunk, s);
        if (e != hipSuccess) return e;
        C *work = static_cast<C *>(s.a), *post = static_cast<C *>(s.b);
        for (long long q0 = 0; q0 < k.batch; q0 += chunk)
        {
            const int nb = (int) std::min<long long>(chunk, k.batch - q0);
            bool tiled = false;
            if constexpr (sizeof(T) == 4)
                if (fx_tile_on() && l1 >= 8 && l1 <= 10 && M2 % RegTile<8>::LN == 0)
                {
                    tiled = true;
                    switch (l1)
                    {
                        case 8: e = launch_cols_tile<8>(device, k, work, M2, M, q0, nb, tw1, twN, st); break;
                        case 9: e = launch_cols_tile<9>(device, k, work, M2, M, q0, nb, tw1, twN, st); break;
                        default: e = launch_cols_tile<10>(device, k, work, M2, M, q0, nb, tw1, twN, st); break;
                    }
                }
            if (!tiled) switch (l1)
            {
#define FX_CASE(L) case L: e = launch_cols<T, L>(k, work, M2, M, q0, nb, tw1, twN, st); break;
                FX_CASE(7) FX_CASE(8) FX_CASE(9) FX_CASE(10) FX_CASE(11)
#undef FX_CASE
                default: e = hipErrorInvalidValue;
            }
            if (e != hipSuccess) return e;
            tiled = false;
            if constexpr (sizeof(T) == 4)
                if (fx_tile_on() && l2 >= 8 && l2 <= 10 && M1 % RegTile<8>::LN == 0)
                {
                    tiled = true;
                    switch (l2)
                    {
                        case 8: e = launch_rows_tile<8>(device, work, k, post, M1, M, q0, nb, tw2, st); break;
                        case 9: e = launch_rows_tile<9>(device, work, k, post, M1, M, q0, nb, tw2, st); break;
                        default: e = launch_rows_tile<10>(device, work, k, post, M1, M, q0, nb, tw2, st); break;
                    }
                }
            if (!tiled) switch (l2)
            {
#define FX_CASE(L) case L: e = launch_rows<T, L>(work, k, post, M1, M, q0, nb, tw2, st); break;
                FX_CASE(7) FX_CASE(8) FX_CASE(9) FX_CASE(10) FX_CASE(11)
#undef FX_CASE
                default: e = hipErrorInvalidValue;
            }
            if (e != hipSuccess) return e;
            if (k.store == S_POST)
                hipLaunchKernelGGL(fx_post_kernel<T>, dim3((M / 2 + 1 + 255) / 256, nb), dim3(256), 0, st, post, k, M, q0, twN);
        }
        return hipGetLastError();
    }

    template <class T> hipError_t run_typed(int device, const FxCall &c, bool src_f32, hipStream_t st, std::string *err)
    {
        typedef typename Cx<T>::type C;
        const long long n = 1LL << c.log2n;
        if (c.op == FX_UNZIP)
        {
            const long long half = n >> 1, in_len = std::min<long long>((long long) c.in_length, n);
            if (!half) return hipSuccess;
            dim3 grid((unsigned) ((half + 255) / 256), (unsigned) std::min<size_t>(c.batch, 65535));
            if (src_f32 && sizeof(T) == 8)
                hipLaunchKernelGGL((fx_unzip_kernel<T, float>), grid, dim3(256), 0, st, static_cast<const float *>(c.src_a), static_cast<T *>(c.dst_a),
                                   static_cast<T *>(c.dst_b), half, in_len, (long long) c.src_stride, (long long) c.dst_stride, (long long) c.batch);
            else
                hipLaunchKernelGGL((fx_unzip_kernel<T, T>), grid, dim3(256), 0, st, static_cast<const T *>(c.src_a), static_cast<T *>(c.dst_a),
                                   static_cast<T *>(c.dst_b), half, in_len, (long long) c.src_stride, (long long) c.dst_stride, (long long) c.batch);
            return hipGetLastError();
        }
        if (c.op == FX_ZIP)
        {
            const long long half = n >> 1;
            if (!half) return hipSuccess;
            dim3 grid((unsigned) ((half + 255) / 256), (unsigned) std::min<size_t>(c.batch, 65535));
            hipLaunchKernelGGL(fx_zip_kernel<T>, grid, dim3(256), 0, st, static_cast<const T *>(c.src_a), static_cast<const T *>(c.src_b),
                               static_cast<T *>(c.dst_a), half, (long long) c.src_stride, (long long) c.dst_stride, (long long) c.batch);
            return hipGetLastError();
        }

        FxK<T> k = {};
        k.sstride = (long long) c.src_stride;
        k.dstride = (long long) c.dst_stride;
        k.batch = (long long) c.batch;
        k.src_f32 = src_f32 ? 1 : 0;
        const bool complex_op = c.op == FX_FFT || c.op == FX_IFFT;
        const int lm = complex_op ? (int) c.log2n : (int) c.log2n - 1;          // complex points = 2^lm (lm = -1: a 1-sample real transform)
        int kind = 0;
        switch (c.op)
        {
            case FX_FFT:
                k.sa = c.src_a; k.sb = static_cast<const T *>(c.src_b); k.da = static_cast<T *>(c.dst_a); k.db = static_cast<T *>(c.dst_b);
                k.load = L_SPLIT; k.store = S_SPLIT;
                break;
            case FX_IFFT:                                                        // Core.h:1341-1346: exchange the pointers
                k.sa = c.src_b; k.sb = static_cast<const T *>(c.src_a); k.da = static_cast<T *>(c.dst_b); k.db = static_cast<T *>(c.dst_a);
                k.load = L_SPLIT; k.store = S_SPLIT;
                break;
            case FX_RFFT:
                k.sa = c.src_a; k.sb = static_cast<const T *>(c.src_b); k.da = static_cast<T *>(c.dst_a); k.db = static_cast<T *>(c.dst_b);
                k.load = L_SPLIT; k.store = S_POST; kind = 1;
                break;
            case FX_RFFT_ZIP:
                k.sa = c.src_a; k.da = static_cast<T *>(c.dst_a); k.db = static_cast<T *>(c.dst_b);
                k.in_len = std::min<long long>((long long) c.in_length, n);
                k.load = L_ZIP; k.store = S_POST; kind = 1;
                break;
            case FX_RIFFT:
                k.sa = c.src_a; k.sb = static_cast<const T *>(c.src_b); k.da = static_cast<T *>(c.dst_a); k.db = static_cast<T *>(c.dst_b);
                k.load = L_PRE; k.store = S_SPLIT; k.swap_out = 1; kind = 2;
                break;
            case FX_RIFFT_ZIP:
                k.sa = c.src_a; k.sb = static_cast<const T *>(c.src_b); k.da = static_cast<T *>(c.dst_a);
                k.load = L_PRE; k.store = S_ZIP; k.swap_out = 1; kind = 2;
                break;
            default: return hipErrorInvalidValue;
        }
        if (lm < 0) return hipSuccess;                                           // nothing to transform or move
        if (lm <= 1)
        {
            if (complex_op && lm == 0 && c.src_a == c.dst_a) return hipSuccess;
            hipLaunchKernelGGL(fx_tiny_kernel<T>, dim3((unsigned) ((k.batch + 255) / 256)), dim3(256), 0, st, k, kind, (int) c.log2n);
            return hipGetLastError();
        }
        if (lm <= max_lds_log2m<T>())
        {
            const C *tw = fx_twiddles<T>(device, lm + 1, err);
            if (!tw) return hipErrorOutOfMemory;
            return dispatch_lds<T>(lm, k, tw, st);
        }
        return run_big<T>(device, lm, k, st, err);
    }
}

const double2 *fftx_twiddles_f64(int device, int log2n, std::string *err) { return twiddles_f64(device, log2n, err); }

bool fftx_valid(const FxCall &c, std::string *err)
{
    auto fail = [&](const char *m) { if (err) *err = m; return false; };
    if (c.op < 0 || c.op >= FX_NUM_OPS) return fail("hcv_fft_exec: unknown operation");
    if (c.precision < FX_F32 || c.precision > FX_F32_TO_F64) return fail("hcv_fft_exec: unknown precision");
    if (c.precision == FX_F32_TO_F64 && c.op != FX_RFFT_ZIP && c.op != FX_UNZIP) return fail("hcv_fft_exec: float-to-double applies to the out-of-place rfft and unzip only");
    const bool complex_op = c.op == FX_FFT || c.op == FX_IFFT;
    const bool shuffle = c.op == FX_UNZIP || c.op == FX_ZIP;
    const unsigned cap = shuffle ? 30u : (unsigned) kFxMaxComplexLog2 + (complex_op ? 0u : 1u);
    if (c.log2n > cap) return fail("hcv_fft_exec: transform size out of range (complex log2 <= 22, real log2 <= 23)");
    if (c.batch > 0x7fffffffull) return fail("hcv_fft_exec: batch too large");
    if (!c.batch) return true;
    const bool two_src = c.op != FX_RFFT_ZIP && c.op != FX_UNZIP, two_dst = c.op != FX_RIFFT_ZIP && c.op != FX_ZIP;
    if (!c.src_a || !c.dst_a || (two_src && !c.src_b) || (two_dst && !c.dst_b)) return fail("hcv_fft_exec: null operand");
    return true;
}

hipError_t fftx_exec(int device, const FxCall &c, hipStream_t stream, std::string *err)
{
    if (!fftx_valid(c, err)) return hipErrorInvalidValue;
    if (!c.batch) return hipSuccess;
    if (c.precision == FX_F32) return run_typed<float>(device, c, false, stream, err);
    return run_typed<double>(device, c, c.precision == FX_F32_TO_F64, stream, err);
}

} // namespace hcv

#ifdef HCV_FX_PHASE_TIMING
// diagnostic builds only (not declared in include/): reads and clears the phase counters of the four-step passes
extern "C" int hcv_debug_fx_phases(unsigned long long *out8)
{
    unsigned long long zero[8] = { 0 };
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(hcv::g_fx_phase), sizeof(zero)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(hcv::g_fx_phase), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif
