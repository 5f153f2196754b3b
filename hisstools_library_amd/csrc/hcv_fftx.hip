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
        size_t bytes = 0, bytes_b = 0;
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

    // grow-only per-device scratch for the four-step path: `bytes` between the passes and, for the real forward transform
    // (`with_post`), as much again in front of its post pass
    hipError_t scratch(int device, size_t bytes, bool with_post, Scratch &out)
    {
        std::lock_guard<std::mutex> g(gFxMutex);
        Scratch &s = gScratch[device];
        auto grow = [&](void *&p, size_t &have) -> hipError_t
        {
            if (have >= bytes) return hipSuccess;
            if (p) (void) hipFree(p);                                   // hipFree waits for work that still uses the old buffer
            p = nullptr;
            have = 0;
            hipError_t e = hipMalloc(&p, bytes);
            if (e == hipSuccess) have = bytes;
            return e;
        };
        hipError_t e = grow(s.a, s.bytes);
        if (e == hipSuccess && with_post) e = grow(s.b, s.bytes_b);
        if (e != hipSuccess) return e;
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

    template <class T> hipError_t run_big(int device, int lm, const FxK<T> &k, hipStream_t st, std::string *err)
    {
        typedef typename Cx<T>::type C;
        const int l2 = (lm + 1) / 2, l1 = lm - l2;
        const int M = 1 << lm, M1 = 1 << l1, M2 = 1 << l2;
        const C *twN = fx_twiddles<T>(device, lm + 1, err);
        const C *tw1 = fx_twiddles<T>(device, l1 + 1, err);
        const C *tw2 = fx_twiddles<T>(device, l2 + 1, err);
        if (!twN || !tw1 || !tw2) return hipErrorOutOfMemory;
        // Scratch for up to 1 GiB of transforms per pass.  (Round 1 to 3 kept a chunk at 64 MiB so that the scratch between the passes
        // stays in the Infinity Cache.  Measured in round 4: the passes are bound by what a CU's memory pipeline moves — four times
        // the batch through ~10 bytes per cycle and CU, wherever the bytes come from — and a launch of 512 tiles on 256 CUs spends
        // a third of its time filling and draining; 1 GiB chunks: complex 2^16 2.10 -> 2.72, 2^20 1.57 -> 2.06 TB/s)
        const size_t per = sizeof(C) * (size_t) M;
        const long long chunk = std::max<long long>(1, std::min<long long>(k.batch, (long long) ((size_t(1) << 30) / per)));
        Scratch s;
        hipError_t e = scratch(device, per * (size_t) chunk, k.store == S_POST, s);
        if (e != hipSuccess) return e;
        C *work = static_cast<C *>(s.a), *post = static_cast<C *>(s.b);
        for (long long q0 = 0; q0 < k.batch; q0 += chunk)
        {
            const int nb = (int) std::min<long long>(chunk, k.batch - q0);
            switch (l1)
            {
#define FX_CASE(L) case L: e = launch_cols<T, L>(k, work, M2, M, q0, nb, tw1, twN, st); break;
                FX_CASE(7) FX_CASE(8) FX_CASE(9) FX_CASE(10) FX_CASE(11)
#undef FX_CASE
                default: e = hipErrorInvalidValue;
            }
            if (e != hipSuccess) return e;
            switch (l2)
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
