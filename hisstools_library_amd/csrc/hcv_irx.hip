// Spectral IR functions on gfx950 (see hcv_irx.h).  Reference behaviour restated (SpectralFunctions.hpp):
//   real_operation            :86-129   the functor sees DC with bin index 0 and Nyquist with bin index N/2, imaginary
//                                       parts zero, and only the real part of its result is kept for those two
//   copy / conjugate          :143-151,173-180
//   spike / delay_calc        :226-263  bin i times exp(i k i), k = 2 pi (as a double) * -position / N in long double
//   amplitude(_linear)        :153-170  phase 0.5
//   minimum_phase_components  :284-340  0.5 log power -> real inverse FFT -> causal window -> real forward FFT
//   complex_exponential, _conjugate, phase_interpolate  :192-237
//   ir_phase                  :392-413
//
// Everything except the two transforms of the minimum-phase path is element-wise and HBM-bound (one read, one write).
// The minimum-phase path runs as ONE kernel while the half spectrum fits the LDS: load + log power -> inverse FFT ->
// window -> forward FFT -> exponential -> store, so HBM is still read once and written once; longer spectra take the
// element-wise kernels around the four-step transforms of hcv_fftx.hip.

#include "hcv_irx.h"
#include "hcv_fftx.h"
#include "hcv_engine.h"
#include "hcv_fft_device.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>

namespace hcv
{

namespace
{
    enum { E_COPY = 0, E_CONJ, E_SPIKE, E_DELAY, E_AMP, E_AMP_LINEAR, E_LOG_POWER, E_EXP, E_EXP_CONJ, E_INTERP };

    template <class T> struct IrK
    {
        const T *sr, *si;
        T *dr, *di;
        long long sstride, dstride;
        int half;                       // values per array
        int mode;
        double value;                   // spike position / delay
        double min_factor, lin_factor;  // phase_interpolate
        long long batch;
    };

    __device__ __forceinline__ float ir_log(float x) { return logf(x); }
    __device__ __forceinline__ double ir_log(double x) { return log(x); }
    __device__ __forceinline__ float ir_exp(float x) { return expf(x); }
    __device__ __forceinline__ double ir_exp(double x) { return exp(x); }
    __device__ __forceinline__ float ir_sqrt(float x) { return sqrtf(x); }
    __device__ __forceinline__ double ir_sqrt(double x) { return sqrt(x); }
    __device__ __forceinline__ void ir_sincos(float x, float *s, float *c) { sincosf(x, s, c); }
    __device__ __forceinline__ void ir_sincos(double x, double *s, double *c) { sincos(x, s, c); }

    // cos / sin of the reference's spike phase k * i, k = (long double) (2.0 * M_PI) * -position / N: the exact product
    // position * i (two doubles) is reduced to a fraction of a turn first, so bins far out stay accurate; the reference's
    // 2 pi is the double-rounded one, a relative excess of EPS over the true 2 pi that is put back to first order.
    __device__ __forceinline__ void spike_phase(double position, int i, double inv_n, double *c, double *s)
    {
        const double EPS = -3.8981718325193755e-17;
        const double fi = (double) i;
        double hi = position * fi;
        double lo = fma(position, fi, -hi);
        hi *= inv_n;
        lo *= inv_n;
        double fr = hi - rint(hi);
        fr += lo + EPS * hi;
        sincospi(-2.0 * fr, s, c);
    }

    // functor of `mode` on the complex value (a, b) at bin index i
    template <class T>
    __device__ __forceinline__ void ir_apply(const IrK<T> &k, int i, T a, T b, double inv_n, T *ro, T *io)
    {
        switch (k.mode)
        {
            case E_COPY: *ro = a; *io = b; break;
            case E_CONJ: *ro = a; *io = -b; break;
            case E_SPIKE:
            {
                double c, s;
                spike_phase(k.value, i, inv_n, &c, &s);
                *ro = (T) c; *io = (T) s;
                break;
            }
            case E_DELAY:
            {
                double c, s;
                spike_phase(k.value, i, inv_n, &c, &s);
                const T ct = (T) c, st = (T) s;
                *ro = a * ct - b * st; *io = a * st + b * ct;
                break;
            }
            case E_AMP: *ro = ir_sqrt(a * a + b * b); *io = (T) 0; break;
            case E_AMP_LINEAR: *ro = ir_sqrt(a * a + b * b) * ((i & 1) ? (T) -1 : (T) 1); *io = (T) 0; break;
            case E_LOG_POWER:
            {
                const T min_power = (T) 1e-30;
                const T p = a * a + b * b;
                *ro = (T) 0.5 * ir_log(p > min_power ? p : min_power); *io = (T) 0;
                break;
            }
            case E_EXP:
            case E_EXP_CONJ:
            {
                T s, c;
                ir_sincos(b, &s, &c);
                const T e = ir_exp(a);
                *ro = e * c; *io = (k.mode == E_EXP) ? e * s : -(e * s);
                break;
            }
            default:    // E_INTERP
            {
                const double amp = (double) ir_exp(a);
                const double ph = k.lin_factor * (double) i + k.min_factor * (double) b;
                double s, c;
                sincos(ph, &s, &c);
                *ro = (T) (amp * c); *io = (T) (amp * s);
            }
        }
    }

    // bin i of one packed spectrum: DC / Nyquist share bin 0 (real_operation, :86-129)
    template <class T>
    __device__ __forceinline__ void ir_bin(const IrK<T> &k, int i, T a, T b, double inv_n, T *ro, T *io)
    {
        if (i == 0)
        {
            T t;
            ir_apply<T>(k, 0, a, (T) 0, inv_n, ro, &t);
            ir_apply<T>(k, k.half, b, (T) 0, inv_n, io, &t);
        }
        else
            ir_apply<T>(k, i, a, b, inv_n, ro, io);
    }

    // VEC consecutive bins per thread (16-byte loads and stores when VEC > 1; the host picks VEC = 1 unless the pointers,
    // strides and length allow it)
    template <class T, int VEC>
    __global__ __launch_bounds__(256) void ir_elementwise_kernel(IrK<T> k)
    {
        const int i0 = (blockIdx.x * 256 + threadIdx.x) * VEC;
        if (i0 >= k.half) return;
        const double inv_n = 0.5 / (double) k.half;
        typedef T VT __attribute__((ext_vector_type(VEC)));
        for (long long q = blockIdx.y; q < k.batch; q += gridDim.y)
        {
            T a[VEC], b[VEC], ro[VEC], io[VEC];
            if (k.mode != E_SPIKE)
            {
                if (VEC > 1)
                {
                    const VT va = *reinterpret_cast<const VT *>(k.sr + q * k.sstride + i0);
                    const VT vb = *reinterpret_cast<const VT *>(k.si + q * k.sstride + i0);
#pragma unroll
                    for (int v = 0; v < VEC; v++) { a[v] = va[v]; b[v] = vb[v]; }
                }
                else
                {
                    a[0] = k.sr[q * k.sstride + i0];
                    b[0] = k.si[q * k.sstride + i0];
                }
            }
            else
            {
#pragma unroll
                for (int v = 0; v < VEC; v++) a[v] = b[v] = (T) 0;
            }
#pragma unroll
            for (int v = 0; v < VEC; v++) ir_bin<T>(k, i0 + v, a[v], b[v], inv_n, &ro[v], &io[v]);
            if (VEC > 1)
            {
                VT vr, vi;
#pragma unroll
                for (int v = 0; v < VEC; v++) { vr[v] = ro[v]; vi[v] = io[v]; }
                *reinterpret_cast<VT *>(k.dr + q * k.dstride + i0) = vr;
                *reinterpret_cast<VT *>(k.di + q * k.dstride + i0) = vi;
            }
            else
            {
                k.dr[q * k.dstride + i0] = ro[0];
                k.di[q * k.dstride + i0] = io[0];
            }
        }
    }

    // causal window of the real cepstrum held as unzipped samples (minimum_phase_components, :313-332); N = 2 * half
    template <class T>
    __device__ __forceinline__ void cepstral_window(int i, int half, T *r, T *im)
    {
        const double scale = 0.5 / (double) half;              // 1 / fft_size
        const int quarter = half >> 1;                         // fft_size >> 2
        if (i == 0) { *r = (T) (*r * (0.5 * scale)); *im = (T) (*im * scale); }
        else if (i < quarter) { *r = (T) (*r * scale); *im = (T) (*im * scale); }
        else if (i == quarter) { *r = (T) (*r * (0.5 * scale)); *im = (T) 0; }
        else { *r = (T) 0; *im = (T) 0; }
    }

    template <class T>
    __global__ __launch_bounds__(256) void ir_window_kernel(T *re, T *im, long long stride, int half, long long batch)
    {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i >= half) return;
        for (long long q = blockIdx.y; q < batch; q += gridDim.y)
        {
            T r = re[q * stride + i], m = im[q * stride + i];
            cepstral_window<T>(i, half, &r, &m);
            re[q * stride + i] = r;
            im[q * stride + i] = m;
        }
    }

    // -------------------------------------------------------------------------------------------- fused minimum-phase path
    //
    // One thread group per spectrum, everything between the load and the store in LDS / registers:
    //   load + log power -> [real pre-pass on the fly] inverse FFT -> [causal window on the fly] forward FFT -> real
    //   post-pass + exponential -> store.

    template <class T> struct Cx2;
    template <> struct Cx2<float> { typedef float2 type; };
    template <> struct Cx2<double> { typedef double2 type; };

    // first-pass source of the inverse transform: pass_real_trig_table<true> (HISSTools_FFT_Core.h:934-988) of the packed
    // log spectrum in LDS, with re/im exchanged so that the forward kernel acts as the inverse (Core.h:1341-1346)
    template <class T, int LOG2M> struct LogPreLoad
    {
        typedef typename Cx2<T>::type C;
        static constexpr bool is_lds = true;
        LdsBuf<C> s;
        const C *__restrict__ tw;
        __device__ __forceinline__ C operator()(int n) const
        {
            constexpr int M = 1 << LOG2M;
            if (n == 0)
            {
                const C z = s[0];
                return C(z.x - z.y, z.x + z.y);
            }
            const bool lo = n <= M / 2;
            const int k = lo ? n : M - n, m = M - k;
            const C w = tw[k];
            const T c = -w.x, sn = w.y;
            const C z1 = s[k], z2 = s[m];
            const T r3 = z1.x + z2.x, i3 = z1.y + z2.y, r4 = z1.x - z2.x, i4 = z1.y - z2.y;
            const T u1 = (c * i3) + (sn * r4);
            const T u2 = (sn * i3) - (c * r4);
            return lo ? C(u2 + i4, r3 + u1) : C(u2 - i4, r3 - u1);
        }
    };

    // first-pass source of the forward transform: the inverse left (x[2n+1], x[2n]) in LDS; window the cepstrum on the way
    template <class T, int LOG2M> struct WindowLoad
    {
        typedef typename Cx2<T>::type C;
        static constexpr bool is_lds = true;
        LdsBuf<C> s;
        __device__ __forceinline__ C operator()(int n) const
        {
            const C v = s[n];
            T r = v.y, m = v.x;
            cepstral_window<T>(n, 1 << LOG2M, &r, &m);
            return C(r, m);
        }
    };

    template <class T, int LOG2M>
    __global__ __launch_bounds__((FFTGeom<LOG2M>::THREADS), (sizeof(T) == 4 ? 4 : 2)) void ir_minphase_kernel(IrK<T> k, const typename Cx2<T>::type *__restrict__ tw)
    {
        typedef typename Cx2<T>::type C;
        typedef FFTGeom<LOG2M> Gm;
        constexpr int M = Gm::M, TG = Gm::TG, G = Gm::G, EPT = (M + TG - 1) / TG;
        extern __shared__ __attribute__((aligned(16))) unsigned char ir_raw[];
        C *lds = reinterpret_cast<C *>(ir_raw);

        const int g = (TG % 64 == 0) ? __builtin_amdgcn_readfirstlane((int) (threadIdx.x / TG)) : (int) (threadIdx.x / TG), t = threadIdx.x % TG;
        const long long q = (long long) blockIdx.x * G + g;
        const bool live = q < k.batch;
        const LdsBuf<C> s = { lds + g * lds_padded(M) };
        const double inv_n = 0.5 / (double) M;

        // load (all of a thread's loads in flight) + log power spectrum, DC / Nyquist packed in bin 0
        {
            const T *sr = k.sr + (live ? q : 0) * k.sstride, *si = k.si + (live ? q : 0) * k.sstride;
            IrK<T> lp = k;
            lp.mode = E_LOG_POWER;
            T a[EPT], b[EPT];
#pragma unroll
            for (int e = 0; e < EPT; e++)
            {
                const int n = t + e * TG;
                const bool in = live && (M % TG == 0 || n < M);
                a[e] = in ? sr[n] : (T) 1;
                b[e] = in ? si[n] : (T) 0;
            }
#pragma unroll
            for (int e = 0; e < EPT; e++)
            {
                const int n = t + e * TG;
                if (M % TG == 0 || n < M)
                {
                    T ro, io;
                    ir_bin<T>(lp, n, a[e], b[e], inv_n, &ro, &io);
                    s[n] = C(ro, io);
                }
            }
        }
        __syncthreads();
        const LdsIO<C> io = { s };
        LdsFFT<LOG2M, TG, C>::run(LogPreLoad<T, LOG2M>{ s, tw }, io, s, t, tw);          // real cepstrum, as (odd, even) pairs
        LdsFFT<LOG2M, TG, C>::run(WindowLoad<T, LOG2M>{ s }, io, s, t, tw);               // windowed, transformed forward
        if (!live) return;
        // pass_real_trig_table<false> (Core.h:934-988) for the bin pair (j, M-j), then the exponential of each bin
        T *dr = k.dr + q * k.dstride, *di = k.di + q * k.dstride;
        for (int j = t; j <= M / 2; j += TG)
        {
            T ro, io2;
            if (j == 0)
            {
                const C z = s[0];
                const T t1 = z.x + z.y, t2 = z.x - z.y;
                ir_bin<T>(k, 0, t1 + t1, t2 + t2, inv_n, &ro, &io2);
                dr[0] = ro;
                di[0] = io2;
                continue;
            }
            const int m = M - j;
            const C w = tw[j];
            const C z1 = s[j], z2 = s[m];
            const T r3 = z1.x + z2.x, i3 = z1.y + z2.y, r4 = z1.x - z2.x, i4 = z1.y - z2.y;
            const T u1 = (w.x * i3) + (w.y * r4);
            const T u2 = (w.y * i3) - (w.x * r4);
            if (m != j)
            {
                ir_bin<T>(k, j, r3 + u1, u2 + i4, inv_n, &ro, &io2);
                dr[j] = ro;
                di[j] = io2;
            }
            ir_bin<T>(k, m, r3 - u1, u2 - i4, inv_n, &ro, &io2);
            dr[m] = ro;
            di[m] = io2;
        }
    }

    template <class T> constexpr int ir_max_lds_log2m() { return sizeof(T) == 4 ? 14 : 13; }

    template <class T, int L> hipError_t launch_minphase(const IrK<T> &k, const typename Cx2<T>::type *tw, hipStream_t st)
    {
        typedef typename Cx2<T>::type C;
        typedef FFTGeom<L> Gm;
        const size_t lds = sizeof(C) * lds_padded(Gm::M) * Gm::G;
        if (lds > 64 * 1024)
        {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ir_minphase_kernel<T, L>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e != hipSuccess) return e;
        }
        const long long grid = (k.batch + Gm::G - 1) / Gm::G;
        hipLaunchKernelGGL((ir_minphase_kernel<T, L>), dim3((unsigned) grid), dim3(Gm::THREADS), lds, st, k, tw);
        return hipGetLastError();
    }

    template <class T> hipError_t dispatch_minphase(int lm, const IrK<T> &k, const typename Cx2<T>::type *tw, hipStream_t st)
    {
        switch (lm)
        {
#define IR_CASE(L) case L: return launch_minphase<T, L>(k, tw, st);
            IR_CASE(2) IR_CASE(3) IR_CASE(4) IR_CASE(5) IR_CASE(6) IR_CASE(7) IR_CASE(8) IR_CASE(9) IR_CASE(10) IR_CASE(11) IR_CASE(12) IR_CASE(13)
#undef IR_CASE
            case 14:
                if constexpr (sizeof(T) == 4) return launch_minphase<T, 14>(k, tw, st);
                return hipErrorInvalidValue;
            default: return hipErrorInvalidValue;
        }
    }

    template <class T> hipError_t launch_elementwise(const IrK<T> &k, hipStream_t st)
    {
        if (!k.half || !k.batch) return hipSuccess;
        constexpr int VEC = 16 / (int) sizeof(T);
        auto aligned = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        const bool vec = k.half % VEC == 0 && k.sstride % VEC == 0 && k.dstride % VEC == 0 && aligned(k.dr) && aligned(k.di) &&
                         (k.mode == E_SPIKE || (aligned(k.sr) && aligned(k.si)));
        const int per = vec ? VEC : 1;
        dim3 grid((unsigned) ((k.half / per + 255) / 256), (unsigned) std::min<long long>(k.batch, 65535));
        if (vec) hipLaunchKernelGGL((ir_elementwise_kernel<T, VEC>), grid, dim3(256), 0, st, k);
        else hipLaunchKernelGGL((ir_elementwise_kernel<T, 1>), grid, dim3(256), 0, st, k);
        return hipGetLastError();
    }

    template <class T> const typename Cx2<T>::type *ir_twiddles(int device, int log2n, std::string *err);
    template <> const float2 *ir_twiddles<float>(int device, int log2n, std::string *err) { return twiddles(device, log2n, err); }
    template <> const double2 *ir_twiddles<double>(int device, int log2n, std::string *err) { return fftx_twiddles_f64(device, log2n, err); }

    template <class T> hipError_t run_typed(int device, const IrCall &c, hipStream_t st, std::string *err)
    {
        const long long n = 1LL << c.log2n;
        IrK<T> k = {};
        k.sr = static_cast<const T *>(c.src_re);
        k.si = static_cast<const T *>(c.src_im);
        k.dr = static_cast<T *>(c.dst_re);
        k.di = static_cast<T *>(c.dst_im);
        k.half = (int) (n >> 1);
        k.sstride = (long long) (c.src_stride ? c.src_stride : (size_t) k.half);
        k.dstride = (long long) (c.dst_stride ? c.dst_stride : (size_t) k.half);
        k.batch = (long long) c.batch;
        k.value = c.value;
        switch (c.op)
        {
            case IR_COPY: k.mode = E_COPY; return launch_elementwise(k, st);
            case IR_TIME_REVERSE: k.mode = E_CONJ; return launch_elementwise(k, st);
            case IR_SPIKE: k.mode = E_SPIKE; return launch_elementwise(k, st);
            case IR_DELAY: k.mode = c.value != 0.0 ? E_DELAY : E_COPY; return launch_elementwise(k, st);   // :377-384
            case IR_PHASE: break;
            default: return hipErrorInvalidValue;
        }
        // ir_phase, :392-413
        const double phase = c.value;
        if (phase == 0.5)
        {
            k.mode = c.zero_center ? E_AMP : E_AMP_LINEAR;
            return launch_elementwise(k, st);
        }
        // the exponential that follows minimum_phase_components
        int exp_mode = E_INTERP;
        double min_factor = 0.0, lin_factor = 0.0;
        if (phase == 1.0 && c.zero_center) exp_mode = E_EXP_CONJ;
        else if (phase == 0.0) exp_mode = E_EXP;
        else
        {
            // phase_interpolate, :212-224 (N.B. a delay of -1 sample for anything over linear, to avoid wraparound)
            const double delay_factor = (phase <= 0.5) ? 0.0 : 1.0 / (double) n;
            const double ph = std::max(0.0, std::min(1.0, phase));
            min_factor = 1.0 - (2.0 * ph);
            lin_factor = c.zero_center ? 0.0 : (-2.0 * M_PI * (ph - delay_factor));
        }
        const int lm = (int) c.log2n - 1;
        if (lm <= ir_max_lds_log2m<T>())
        {
            // one kernel: HBM is read once and written once
            const typename Cx2<T>::type *tw = ir_twiddles<T>(device, lm + 1, err);
            if (!tw) return hipErrorOutOfMemory;
            IrK<T> f = k;
            f.mode = exp_mode;
            f.min_factor = min_factor;
            f.lin_factor = lin_factor;
            return dispatch_minphase<T>(lm, f, tw, st);
        }
        // longer spectra: minimum_phase_components into dst with the four-step transforms, then the exponential in place
        k.mode = E_LOG_POWER;
        hipError_t e = launch_elementwise(k, st);
        if (e != hipSuccess) return e;
        FxCall f;
        f.precision = sizeof(T) == 4 ? FX_F32 : FX_F64;
        f.log2n = c.log2n;
        f.batch = c.batch;
        f.src_a = f.dst_a = c.dst_re;
        f.src_b = f.dst_b = c.dst_im;
        f.src_stride = f.dst_stride = (size_t) k.dstride;
        f.op = FX_RIFFT;
        e = fftx_exec(device, f, st, err);
        if (e != hipSuccess) return e;
        {
            dim3 grid((unsigned) ((k.half + 255) / 256), (unsigned) std::min<long long>(k.batch, 65535));
            hipLaunchKernelGGL(ir_window_kernel<T>, grid, dim3(256), 0, st, k.dr, k.di, k.dstride, k.half, k.batch);
        }
        f.op = FX_RFFT;
        e = fftx_exec(device, f, st, err);
        if (e != hipSuccess) return e;
        IrK<T> x = k;
        x.sr = k.dr;
        x.si = k.di;
        x.sstride = k.dstride;
        x.mode = exp_mode;
        x.min_factor = min_factor;
        x.lin_factor = lin_factor;
        return launch_elementwise(x, st);
    }
}

namespace
{
    template <class T> __global__ void ir_scale_kernel(T *x, long long n, T scale)
    {
        const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
        if (i < n) x[i] *= scale;
    }
}

hipError_t launch_scale(float *x, long long n, float scale, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(ir_scale_kernel<float>, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, x, n, scale);
    return hipGetLastError();
}

hipError_t launch_scale(double *x, long long n, double scale, hipStream_t st)
{
    if (n > 0) hipLaunchKernelGGL(ir_scale_kernel<double>, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st, x, n, scale);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ the IR products
// ir_convolve_complex / ir_convolve_real / ir_correlate_complex / ir_correlate_real (SpectralFunctions.hpp:415-436 over
// complex_operation :49-61 / real_operation :63-84 and the functors convolve :274-281, correlate :265-272).  Each product and each sum
// is rounded on its own, then the scale — the reference's vector layer has no fused multiply-add, and neither has this kernel
// (__fmul_rn / __fadd_rn and their double forms are never contracted): bit-identical results.
namespace
{
    template <class T> struct IrP
    {
        const T *ar, *ai, *br, *bi;
        T *dr, *di;
        long long astride, bstride, dstride, count, batch;
        int correlate, real_form;
        T scale;
    };
    __device__ __forceinline__ float rn_mul(float a, float b) { return __fmul_rn(a, b); }
    __device__ __forceinline__ double rn_mul(double a, double b) { return __dmul_rn(a, b); }
    __device__ __forceinline__ float rn_add(float a, float b) { return __fadd_rn(a, b); }
    __device__ __forceinline__ double rn_add(double a, double b) { return __dadd_rn(a, b); }

    template <class T> __global__ __launch_bounds__(256) void ir_product_kernel(IrP<T> k)
    {
        for (long long row = blockIdx.y; row < k.batch; row += gridDim.y)
        {
            const T *ar = k.ar + row * k.astride, *ai = k.ai + row * k.astride, *br = k.br + row * k.bstride, *bi = k.bi + row * k.bstride;
            T *dr = k.dr + row * k.dstride, *di = k.di + row * k.dstride;
            for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < k.count; i += (long long) gridDim.x * blockDim.x)
            {
                const T a = ar[i], b = ai[i], c = br[i], d = bi[i];
                T re, im;
                if (k.real_form && i == 0)
                {
                    // bin 0 = (DC, Nyquist): op(dc, ., a, 0, c, 0) and op(nq, ., b, 0, d, 0): two real products (:73-83)
                    re = rn_mul(k.scale, rn_add(rn_mul(a, c), k.correlate ? (T) 0 : -(T) 0));
                    im = rn_mul(k.scale, rn_add(rn_mul(b, d), k.correlate ? (T) 0 : -(T) 0));
                }
                else if (k.correlate)
                {
                    re = rn_mul(k.scale, rn_add(rn_mul(a, c), rn_mul(b, d)));
                    im = rn_mul(k.scale, rn_add(rn_mul(b, c), -rn_mul(a, d)));
                }
                else
                {
                    re = rn_mul(k.scale, rn_add(rn_mul(a, c), -rn_mul(b, d)));
                    im = rn_mul(k.scale, rn_add(rn_mul(b, c), rn_mul(a, d)));
                }
                dr[i] = re;
                di[i] = im;
            }
        }
    }

    template <class T> hipError_t run_product(const IrProduct &c, hipStream_t st)
    {
        IrP<T> k;
        k.ar = static_cast<const T *>(c.a_re); k.ai = static_cast<const T *>(c.a_im);
        k.br = static_cast<const T *>(c.b_re); k.bi = static_cast<const T *>(c.b_im);
        k.dr = static_cast<T *>(c.dst_re); k.di = static_cast<T *>(c.dst_im);
        k.count = (long long) c.count;
        k.batch = (long long) c.batch;
        k.astride = (long long) (c.a_stride ? c.a_stride : c.count);
        k.bstride = c.b_broadcast ? 0 : (long long) (c.b_stride ? c.b_stride : c.count);
        k.dstride = (long long) (c.dst_stride ? c.dst_stride : c.count);
        k.correlate = c.op >= IRP_CORRELATE_COMPLEX;
        k.real_form = c.op & 1;
        k.scale = (T) c.scale;
        const unsigned gx = (unsigned) std::max<long long>(1, std::min<long long>((k.count + 255) / 256, 1024));
        hipLaunchKernelGGL(ir_product_kernel<T>, dim3(gx, (unsigned) std::min<long long>(k.batch, 4096)), dim3(256), 0, st, k);
        return hipGetLastError();
    }
}

bool irp_valid(const IrProduct &c, std::string *err)
{
    auto fail = [&](const char *m) { if (err) *err = m; return false; };
    if (c.op < 0 || c.op >= IRP_NUM_OPS) return fail("hcv_ir_product: unknown operation");
    if (c.precision != FX_F32 && c.precision != FX_F64) return fail("hcv_ir_product: precision must be float or double");
    if (!c.count || (c.count & (c.count - 1))) return fail("hcv_ir_product: power-of-two sizes only (the reference's vector loops drop the remainder of any other)");
    if (c.count > (size_t(1) << 28) || c.batch > 0x7fffffffull) return fail("hcv_ir_product: size out of range");
    if (!c.batch) return true;
    if (!c.a_re || !c.a_im || !c.b_re || !c.b_im || !c.dst_re || !c.dst_im) return fail("hcv_ir_product: null operand");
    return true;
}

hipError_t irp_exec(const IrProduct &c, hipStream_t stream, std::string *err)
{
    if (!irp_valid(c, err)) return hipErrorInvalidValue;
    if (!c.batch) return hipSuccess;
    return c.precision == FX_F32 ? run_product<float>(c, stream) : run_product<double>(c, stream);
}

bool irx_valid(const IrCall &c, std::string *err)
{
    auto fail = [&](const char *m) { if (err) *err = m; return false; };
    if (c.op < 0 || c.op >= IR_NUM_OPS) return fail("hcv_ir_exec: unknown operation");
    if (c.precision != FX_F32 && c.precision != FX_F64) return fail("hcv_ir_exec: precision must be float or double");
    if (c.log2n < 1 || c.log2n > (unsigned) kFxMaxComplexLog2 + 1) return fail("hcv_ir_exec: fft size out of range (2 .. 2^23 samples)");
    if (c.op == IR_PHASE && c.value != 0.5 && c.log2n < 3) return fail("hcv_ir_exec: ir_phase needs an fft size of at least 8");
    if (c.batch > 0x7fffffffull) return fail("hcv_ir_exec: batch too large");
    if (!c.batch) return true;
    if (!c.dst_re || !c.dst_im || (c.op != IR_SPIKE && (!c.src_re || !c.src_im))) return fail("hcv_ir_exec: null operand");
    return true;
}

hipError_t irx_exec(int device, const IrCall &c, hipStream_t stream, std::string *err)
{
    if (!irx_valid(c, err)) return hipErrorInvalidValue;
    if (!c.batch) return hipSuccess;
    return c.precision == FX_F32 ? run_typed<float>(device, c, stream, err) : run_typed<double>(device, c, stream, err);
}

} // namespace hcv
