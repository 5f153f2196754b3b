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

    template <class T>
    __global__ __launch_bounds__(256) void ir_elementwise_kernel(IrK<T> k)
    {
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i >= k.half) return;
        const double inv_n = 0.5 / (double) k.half;
        for (long long q = blockIdx.y; q < k.batch; q += gridDim.y)
        {
            T a = (T) 0, b = (T) 0;
            if (k.mode != E_SPIKE)
            {
                a = k.sr[q * k.sstride + i];
                b = k.si[q * k.sstride + i];
            }
            T ro, io;
            ir_bin<T>(k, i, a, b, inv_n, &ro, &io);
            k.dr[q * k.dstride + i] = ro;
            k.di[q * k.dstride + i] = io;
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

    template <class T> hipError_t launch_elementwise(const IrK<T> &k, hipStream_t st)
    {
        if (!k.half || !k.batch) return hipSuccess;
        dim3 grid((unsigned) ((k.half + 255) / 256), (unsigned) std::min<long long>(k.batch, 65535));
        hipLaunchKernelGGL(ir_elementwise_kernel<T>, grid, dim3(256), 0, st, k);
        return hipGetLastError();
    }

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
        // minimum_phase_components into dst, then the exponential in place on dst
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
        if (phase == 1.0 && c.zero_center) x.mode = E_EXP_CONJ;
        else if (phase == 0.0) x.mode = E_EXP;
        else
        {
            // phase_interpolate, :212-224 (N.B. a delay of -1 sample for anything over linear, to avoid wraparound)
            const double delay_factor = (phase <= 0.5) ? 0.0 : 1.0 / (double) n;
            const double ph = std::max(0.0, std::min(1.0, phase));
            x.mode = E_INTERP;
            x.min_factor = 1.0 - (2.0 * ph);
            x.lin_factor = c.zero_center ? 0.0 : (-2.0 * M_PI * (ph - delay_factor));
        }
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
