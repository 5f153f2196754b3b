// C ABI, second part (include/hisstools_amd.h): the FFT plumbing of the convolution path (hcv_rfft_f32 / hcv_rifft_f32), the
// whole hisstools_* FFT surface (hcv_fft_exec), spectral_processor convolve / correlate / change_phase and the spectral IR
// functions (hcv_ir_exec).  Host variants only move bytes; the arithmetic is in hcv_kernels.hip, hcv_fftx.hip, hcv_irx.hip.

#include "../../include/hisstools_amd.h"
#include "hcv_api_common.h"
#include "hcv_engine.h"
#include "hcv_fftx.h"
#include "hcv_irx.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

using hcv_api::gDefaultDevice;
using hcv_api::set_error;
using hcv_api::tlsError;

// ------------------------------------------------------------------------------------------------ FFT plumbing

static bool fft_size_ok(unsigned log2n)
{
    if (log2n < 5 || log2n > (unsigned) hcv::kMaxFFTLog2)
    {
        set_error("hcv_rfft/rifft: log2n must be in [5, 20]");
        return false;
    }
    return true;
}

// scratch + sub-transform tables for one-off transforms above the LDS limit
struct ScopedBigWork
{
    hcv::BigFFTWork w;
    bool ok = true;
    ScopedBigWork(int dev, unsigned log2n, size_t batch)
    {
        if (!hcv::is_big_fft((int) log2n)) return;
        int l1, l2;
        hcv::big_fft_split((int) log2n, l1, l2);
        std::string err;
        w.tw1 = hcv::twiddles(dev, l1 + 1, &err);
        w.tw2 = hcv::twiddles(dev, l2 + 1, &err);
        w.elems = (size_t(1) << (log2n - 1)) * std::min<size_t>(batch, 16);
        ok = w.tw1 && w.tw2 && hipMalloc(&w.a, sizeof(float2) * w.elems) == hipSuccess && hipMalloc(&w.b, sizeof(float2) * w.elems) == hipSuccess;
        if (!ok) set_error(err.empty() ? "big FFT workspace allocation failed" : err);
    }
    ~ScopedBigWork()
    {
        if (w.a) (void) hipFree(w.a);
        if (w.b) (void) hipFree(w.b);
    }
};

#define HCV_API_TRY(expr)                                                                                              \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess)                                                                                          \
        {                                                                                                              \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                                              \
            ok = false;                                                                                                \
        }                                                                                                              \
    } while (0)

extern "C" int hcv_rfft_f32(const float *in, size_t in_length, size_t in_stride, size_t batch, unsigned log2n, float *realp, float *imagp)
{
    if (!fft_size_ok(log2n) || !batch) return batch ? -1 : 0;
    int dev = 0;
    if (hcv_device_count() <= 0)
    {
        set_error("no HIP device available");
        return -1;
    }
    if (gDefaultDevice >= 0) (void) hipSetDevice(gDefaultDevice);
    (void) hipGetDevice(&dev);
    std::string err;
    const float2 *tw = hcv::twiddles(dev, (int) log2n, &err);
    if (!tw)
    {
        set_error(err);
        return -1;
    }
    const size_t half = size_t(1) << (log2n - 1), n = half * 2;
    const size_t take = std::min(in_length, n);
    bool ok = true;
    float *din = nullptr;
    float2 *dout = nullptr;
    std::vector<float> packed(batch * take);
    for (size_t b = 0; b < batch; b++) std::memcpy(packed.data() + b * take, in + b * in_stride, sizeof(float) * take);
    std::vector<float2> spec(batch * half);
    HCV_API_TRY(hipMalloc(&din, sizeof(float) * std::max<size_t>(1, batch * take)));
    if (ok) HCV_API_TRY(hipMalloc(&dout, sizeof(float2) * batch * half));
    if (ok && take) HCV_API_TRY(hipMemcpy(din, packed.data(), sizeof(float) * batch * take, hipMemcpyHostToDevice));
    ScopedBigWork big(dev, log2n, batch);
    ok = ok && big.ok;
    if (ok) HCV_API_TRY(hcv::launch_rfft_rows((int) log2n, din, (long long) take, (long long) take, (int) batch, dout, tw, &big.w, nullptr));
    if (ok) HCV_API_TRY(hipMemcpy(spec.data(), dout, sizeof(float2) * batch * half, hipMemcpyDeviceToHost));
    if (din) (void) hipFree(din);
    if (dout) (void) hipFree(dout);
    if (!ok) return -1;
    for (size_t e = 0; e < batch * half; e++)
    {
        realp[e] = spec[e].x;
        imagp[e] = spec[e].y;
    }
    return 0;
}

extern "C" int hcv_rifft_f32(const float *realp, const float *imagp, size_t batch, unsigned log2n, float *out)
{
    if (!fft_size_ok(log2n) || !batch) return batch ? -1 : 0;
    int dev = 0;
    if (hcv_device_count() <= 0)
    {
        set_error("no HIP device available");
        return -1;
    }
    if (gDefaultDevice >= 0) (void) hipSetDevice(gDefaultDevice);
    (void) hipGetDevice(&dev);
    std::string err;
    const float2 *tw = hcv::twiddles(dev, (int) log2n, &err);
    if (!tw)
    {
        set_error(err);
        return -1;
    }
    const size_t half = size_t(1) << (log2n - 1), n = half * 2;
    std::vector<float2> spec(batch * half);
    for (size_t e = 0; e < batch * half; e++) spec[e] = make_float2(realp[e], imagp[e]);
    bool ok = true;
    float2 *din = nullptr;
    float *dout = nullptr;
    HCV_API_TRY(hipMalloc(&din, sizeof(float2) * batch * half));
    if (ok) HCV_API_TRY(hipMalloc(&dout, sizeof(float) * batch * n));
    if (ok) HCV_API_TRY(hipMemcpy(din, spec.data(), sizeof(float2) * batch * half, hipMemcpyHostToDevice));
    ScopedBigWork big(dev, log2n, batch);
    ok = ok && big.ok;
    if (ok) HCV_API_TRY(hcv::launch_rifft_rows((int) log2n, din, (int) batch, dout, tw, &big.w, nullptr));
    if (ok) HCV_API_TRY(hipMemcpy(out, dout, sizeof(float) * batch * n, hipMemcpyDeviceToHost));
    if (din) (void) hipFree(din);
    if (dout) (void) hipFree(dout);
    return ok ? 0 : -1;
}

// ------------------------------------------------------------------------------------------------ spectral_processor (next row)
//
// spectral_processor<float>::convolve / correlate, real overloads (SpectralProcessor.hpp:173-184): both inputs are
// transformed with the real FFT kernels, multiplied bin-wise on the device, inverted and arranged per edge mode.
// Host code only sizes the problem and lays the inputs out (zero padding, mirrored edges); it does no arithmetic.

namespace
{
    enum { EDGE_LINEAR = 0, EDGE_WRAP = 1, EDGE_WRAP_CENTRE = 2, EDGE_FOLD = 3, EDGE_FOLD_REPEAT = 4 };

    struct OpSizes                                              // op_sizes, SpectralProcessor.hpp:318-357
    {
        int mode;
        bool fold;
        size_t size1, size2, mn, mx, linear, fold_copy, fft;
        unsigned fft_log2;
    };

    unsigned spectral_log2(size_t size)                         // calc_fft_size_log2, :231-243
    {
        unsigned count = 0;
        while (count < 8 * sizeof(size_t) && (size >> count)) count++;
        if (count && size == (size_t(1) << (count - 1))) return count - 1;
        return count;
    }

    OpSizes op_sizes(size_t n1, size_t n2, int mode)
    {
        OpSizes s;
        s.mode = mode;
        s.fold = mode == EDGE_FOLD || mode == EDGE_FOLD_REPEAT;
        s.size1 = n1;
        s.size2 = n2;
        s.mn = std::min(n1, n2);
        s.mx = std::max(n1, n2);
        s.linear = n1 + n2 - 1;
        s.fold_copy = s.mx + ((s.mn >> 1) << 1);
        s.fft_log2 = spectral_log2(s.fold ? s.fold_copy + (s.mn - 1) : s.linear);
        s.fft = size_t(1) << s.fft_log2;
        return s;
    }

    // device scratch of one spectral_binary call: the folded operand, both spectra, the circular result, four-step work
    struct SpectralWork
    {
        float *fold = nullptr, *t = nullptr;
        float2 *spec = nullptr;
        hcv::BigFFTWork big;
        size_t fold_elems = 0, t_elems = 0, spec_elems = 0, big_elems = 0;
        void release()
        {
            if (fold) (void) hipFree(fold);
            if (t) (void) hipFree(t);
            if (spec) (void) hipFree(spec);
            if (big.a) (void) hipFree(big.a);
            if (big.b) (void) hipFree(big.b);
            *this = SpectralWork();
        }
        // grow-only; false on allocation failure
        bool reserve(int dev, unsigned log2n, std::string *err)
        {
            const size_t fft = size_t(1) << log2n, half = fft >> 1;
            auto grow = [](auto *&p, size_t &have, size_t want)
            {
                if (have >= want) return true;
                if (p) (void) hipFree(p);
                p = nullptr;
                have = 0;
                if (hipMalloc(&p, sizeof(*p) * want) != hipSuccess) return false;
                have = want;
                return true;
            };
            if (!grow(fold, fold_elems, fft) || !grow(t, t_elems, fft) || !grow(spec, spec_elems, 2 * half)) return false;
            if (hcv::is_big_fft((int) log2n))
            {
                int l1, l2;
                hcv::big_fft_split((int) log2n, l1, l2);
                big.tw1 = hcv::twiddles(dev, l1 + 1, err);
                big.tw2 = hcv::twiddles(dev, l2 + 1, err);
                if (!big.tw1 || !big.tw2) return false;
                if (big_elems < half)
                {
                    if (big.a) (void) hipFree(big.a);
                    if (big.b) (void) hipFree(big.b);
                    big.a = big.b = nullptr;
                    big_elems = 0;
                    if (hipMalloc(&big.a, sizeof(float2) * half) != hipSuccess || hipMalloc(&big.b, sizeof(float2) * half) != hipSuccess) return false;
                    big_elems = half;
                }
                big.elems = big_elems;
            }
            return true;
        }
    };
}

extern "C" size_t hcv_spectral_size(size_t size1, size_t size2, int mode)                      // calc_conv_corr_size, :549-560
{
    if (!size1 || !size2 || mode < 0 || mode > EDGE_FOLD_REPEAT) return 0;
    const OpSizes s = op_sizes(size1, size2, mode);
    if (s.fft_log2 > hcv_api::kMaxSpectralLog2) return 0;       // our "max_fft_size" is 2^22
    return mode != EDGE_LINEAR ? s.mx : s.linear;
}

// Both operands and the result are device-resident; everything is enqueued on `st`.
static bool spectral_core(int dev, const float *d1, size_t n1, const float *d2, size_t n2, int mode, bool correlate, float *dout, SpectralWork &w,
                          hipStream_t st)
{
    const OpSizes s = op_sizes(n1, n2, mode);
    // the device FFTs start at 32 points; a larger circular size is equivalent as long as every index below uses it
    const unsigned log2n = std::max(s.fft_log2, 5u);
    const size_t fft = size_t(1) << log2n, half = fft >> 1;
    std::string err;
    const float2 *tw = hcv::twiddles(dev, (int) log2n, &err);
    if (!tw || !w.reserve(dev, log2n, &err))
    {
        set_error(err.empty() ? "spectral_processor: device allocation failed" : err);
        return false;
    }
    bool ok = true;
    // operands: the longer one is mirrored at both ends in the fold modes (copy_fold, :361-377); the FFT loader zero-pads
    const size_t fold_size = s.mn >> 1;
    const int fold_off = mode == EDGE_FOLD_REPEAT ? 0 : 1;
    const float *row1 = d1, *row2 = d2;
    size_t len1 = n1, len2 = n2;
    if (s.fold)
    {
        const bool first = n1 >= n2;
        HCV_API_TRY(hcv::launch_fold_copy(w.fold, first ? d1 : d2, (long long) (first ? n1 : n2), (long long) fold_size, fold_off, st));
        (first ? row1 : row2) = w.fold;
        (first ? len1 : len2) += 2 * fold_size;
    }
    if (ok) HCV_API_TRY(hcv::launch_rfft_rows((int) log2n, row1, (long long) len1, (long long) len1, 1, w.spec, tw, &w.big, st));
    if (ok) HCV_API_TRY(hcv::launch_rfft_rows((int) log2n, row2, (long long) len2, (long long) len2, 1, w.spec + half, tw, &w.big, st));
    if (ok) HCV_API_TRY(hcv::launch_spectral_pointwise(w.spec, w.spec + half, (int) half, 0.25f / (float) fft, correlate ? 1 : 0, st));
    if (ok) HCV_API_TRY(hcv::launch_rifft_rows((int) log2n, w.spec, 1, w.t, tw, &w.big, st));

    auto seg = [&](size_t o_off, size_t off, size_t n, int op)
    {
        if (ok) HCV_API_TRY(hcv::launch_segment_op(dout, w.t, (long long) o_off, (long long) off, (long long) n, op, st));
    };
    auto copy = [&](size_t o_off, size_t off, size_t n) { seg(o_off, off, n, 0); };
    auto wrap = [&](size_t o_off, size_t last, size_t n) { seg(o_off, last - n, n, 1); };     // adds t[last-n .. last)
    auto zero = [&](size_t a, size_t b) { if (b > a) seg(a, 0, b - a, 2); };

    if (n1 == 1 && n2 == 1)
        copy(0, 0, 1);                                          // circular product of two 1-sample signals
    else if (!correlate)
    {
        const size_t min_m1 = s.mn - 1;                         // arrange_convolve, :448-486
        switch (mode)
        {
            case EDGE_LINEAR: copy(0, 0, s.linear); break;
            case EDGE_WRAP: copy(0, 0, s.mx); wrap(0, s.linear, min_m1); break;
            case EDGE_WRAP_CENTRE:
            {
                const size_t wrapped = min_m1 >> 1;
                copy(0, wrapped, s.mx);
                wrap(0, s.linear, min_m1 - wrapped);
                wrap(s.mx - wrapped, wrapped, wrapped);
                break;
            }
            default: copy(0, min_m1, s.mx); break;
        }
    }
    else
    {
        const size_t size2_m1 = s.size2 - 1;                    // arrange_correlate, :488-545 (fft = the size actually used)
        switch (mode)
        {
            case EDGE_LINEAR: copy(0, 0, s.size1); copy(s.size1, fft - size2_m1, size2_m1); break;
            case EDGE_WRAP:
                copy(0, 0, s.size1);
                zero(s.size1, s.size2);
                wrap(s.mx - size2_m1, fft, size2_m1);
                break;
            case EDGE_WRAP_CENTRE:
            {
                const size_t w1 = (s.mn - 1) >> 1;
                const size_t w2 = std::min(size2_m1, s.mx - w1);
                const size_t w3 = size2_m1 - w2;
                const size_t offset = w3 ? 0 : s.mx - (size2_m1 + w1);
                zero(0, s.mx);
                copy(0, w1, s.size1 - w1);
                copy(s.mx - w1, 0, w1);
                wrap(offset, fft, w2);
                wrap(s.mx - w3, fft - w2, w3);
                break;
            }
            default:
                if (s.size1 >= s.size2)
                    copy(0, 0, s.mx);
                else
                {
                    const size_t cs = s.mx - 1;
                    copy(0, 0, 1);
                    copy(1, fft - cs, cs);
                }
                break;
        }
    }
    return ok;
}

static bool spectral_ready(size_t n1, size_t n2, int mode, int &dev, size_t &result)
{
    result = hcv_spectral_size(n1, n2, mode);
    if (!result) return false;                                  // the reference returns without touching `out` (:651-652)
    if (hcv_device_count() <= 0)
    {
        set_error("no HIP device available (no CPU fallback)");
        result = (size_t) -1;
        return false;
    }
    if (gDefaultDevice >= 0) (void) hipSetDevice(gDefaultDevice);
    (void) hipGetDevice(&dev);
    return true;
}

static int spectral_binary(const float *in1, size_t n1, const float *in2, size_t n2, int mode, bool correlate, float *out)
{
    int dev = 0;
    size_t result = 0;
    if (!spectral_ready(n1, n2, mode, dev, result)) return result == (size_t) -1 ? -1 : 0;
    // beyond the convolution engine's largest FFT the general FFT surface takes over (hcv_api_spectral.hip)
    if (op_sizes(n1, n2, mode).fft_log2 > (unsigned) hcv::kMaxFFTLog2) return hcv_api::spectral_real_general_f32(in1, n1, in2, n2, mode, correlate, out);
    bool ok = true;
    float *d1 = nullptr, *d2 = nullptr, *dout = nullptr;
    SpectralWork w;
    HCV_API_TRY(hipMalloc(&d1, sizeof(float) * n1));
    if (ok) HCV_API_TRY(hipMalloc(&d2, sizeof(float) * n2));
    if (ok) HCV_API_TRY(hipMalloc(&dout, sizeof(float) * result));
    if (ok) HCV_API_TRY(hipMemcpy(d1, in1, sizeof(float) * n1, hipMemcpyHostToDevice));
    if (ok) HCV_API_TRY(hipMemcpy(d2, in2, sizeof(float) * n2, hipMemcpyHostToDevice));
    ok = ok && spectral_core(dev, d1, n1, d2, n2, mode, correlate, dout, w, nullptr);
    if (ok) HCV_API_TRY(hipMemcpy(out, dout, sizeof(float) * result, hipMemcpyDeviceToHost));
    if (d1) (void) hipFree(d1);
    if (d2) (void) hipFree(d2);
    if (dout) (void) hipFree(dout);
    w.release();
    return ok ? 0 : -1;
}

// HBM-resident operands: scratch is cached per device (grow-only) and the calls of one device are serialised by a mutex
// while they enqueue; calls on different streams that overlap in time must be ordered by the caller.
static int spectral_binary_dev(const float *d1, size_t n1, const float *d2, size_t n2, int mode, bool correlate, float *dout, void *stream, int sync)
{
    int dev = 0;
    size_t result = 0;
    if (!spectral_ready(n1, n2, mode, dev, result)) return result == (size_t) -1 ? -1 : 0;
    if (op_sizes(n1, n2, mode).fft_log2 > (unsigned) hcv::kMaxFFTLog2)
    {
        set_error("hcv_spectral_*_f32_dev: circular sizes above 2^20 are served by the host-pointer entry points only");
        return -1;
    }
    static std::mutex mutex;
    static std::map<int, SpectralWork> cache;
    std::lock_guard<std::mutex> g(mutex);
    hipStream_t st = static_cast<hipStream_t>(stream);
    bool ok = spectral_core(dev, d1, n1, d2, n2, mode, correlate, dout, cache[dev], st);
    if (ok && sync) HCV_API_TRY(hipStreamSynchronize(st));
    return ok ? 0 : -1;
}

extern "C" int hcv_spectral_convolve_f32(const float *in1, size_t size1, const float *in2, size_t size2, int mode, float *out)
{
    return spectral_binary(in1, size1, in2, size2, mode, false, out);
}

extern "C" int hcv_spectral_correlate_f32(const float *in1, size_t size1, const float *in2, size_t size2, int mode, float *out)
{
    return spectral_binary(in1, size1, in2, size2, mode, true, out);
}

extern "C" int hcv_spectral_convolve_f32_dev(const float *in1, size_t size1, const float *in2, size_t size2, int mode, float *out, void *stream, int sync)
{
    return spectral_binary_dev(in1, size1, in2, size2, mode, false, out, stream, sync);
}

extern "C" int hcv_spectral_correlate_f32_dev(const float *in1, size_t size1, const float *in2, size_t size2, int mode, float *out, void *stream, int sync)
{
    return spectral_binary_dev(in1, size1, in2, size2, mode, true, out, stream, sync);
}

// ------------------------------------------------------------------------------------------------ the full FFT surface (next row 2)
//
// HISSTools_FFT.h:87-369 as one batched entry point; the kernels are in hcv_fftx.hip.  The host variant only moves
// bytes: upload the source extent, run, download the destination extent.

namespace
{
    struct FftOperand
    {
        size_t len = 0, elem = 0;       // elements per transform, bytes per element
        bool two = false;               // split (a and b) or samples (a only)
    };

    bool use_default_device(int &dev)
    {
        if (hcv_device_count() <= 0)
        {
            set_error("no HIP device available");
            return false;
        }
        if (gDefaultDevice >= 0) (void) hipSetDevice(gDefaultDevice);
        return hipGetDevice(&dev) == hipSuccess;
    }

    hcv::FxCall to_fx(const hcv_fft_call &c)
    {
        hcv::FxCall f;
        f.op = c.op; f.precision = c.precision; f.log2n = c.log2n; f.batch = c.batch;
        f.src_a = c.src_a; f.src_b = c.src_b; f.dst_a = c.dst_a; f.dst_b = c.dst_b;
        f.src_stride = c.src_stride; f.dst_stride = c.dst_stride; f.in_length = c.in_length;
        return f;
    }

    // shapes of the two operands, and default (dense) strides
    void fft_operands(hcv::FxCall &f, FftOperand &src, FftOperand &dst)
    {
        const size_t n = size_t(1) << f.log2n, half = n >> 1;
        const size_t real_bytes = f.precision == hcv::FX_F32 ? 4 : 8;
        const bool complex_op = f.op == hcv::FX_FFT || f.op == hcv::FX_IFFT;
        const size_t split_len = complex_op ? n : half;
        src.elem = dst.elem = real_bytes;
        switch (f.op)
        {
            case hcv::FX_RFFT_ZIP:
            case hcv::FX_UNZIP:
                f.in_length = std::min(f.in_length, n);
                src.len = f.in_length; src.two = false;
                src.elem = f.precision == hcv::FX_F64 ? 8 : 4;
                dst.len = half; dst.two = true;
                break;
            case hcv::FX_RIFFT_ZIP:
            case hcv::FX_ZIP:
                src.len = half; src.two = true;
                dst.len = half ? n : 0; dst.two = false;
                break;
            default:
                src.len = dst.len = split_len;
                src.two = dst.two = true;
        }
        if (!f.src_stride) f.src_stride = src.len;
        if (!f.dst_stride) f.dst_stride = dst.len;
    }
}

extern "C" int hcv_fft_exec_dev(const hcv_fft_call *call, void *stream, int sync)
{
    if (!call)
    {
        set_error("hcv_fft_exec_dev: null descriptor");
        return -1;
    }
    int dev = 0;
    if (!use_default_device(dev)) return -1;
    hcv::FxCall f = to_fx(*call);
    std::string err;
    if (!hcv::fftx_valid(f, &err))
    {
        set_error(err);
        return -1;
    }
    FftOperand src, dst;
    fft_operands(f, src, dst);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hcv::fftx_exec(dev, f, st, &err);
    if (e == hipSuccess && sync) e = hipStreamSynchronize(st);
    if (e != hipSuccess)
    {
        set_error(err.empty() ? std::string("hcv_fft_exec_dev: ") + hipGetErrorString(e) : err);
        return -1;
    }
    return 0;
}

extern "C" int hcv_fft_exec(const hcv_fft_call *call)
{
    if (!call)
    {
        set_error("hcv_fft_exec: null descriptor");
        return -1;
    }
    int dev = 0;
    if (!use_default_device(dev)) return -1;
    hcv::FxCall f = to_fx(*call);
    std::string err;
    if (!hcv::fftx_valid(f, &err))
    {
        set_error(err);
        return -1;
    }
    if (!f.batch) return 0;
    FftOperand src, dst;
    fft_operands(f, src, dst);
    const bool in_place = call->src_a == call->dst_a;
    if (in_place && (src.two != dst.two || src.elem != dst.elem || (src.two && call->src_b != call->dst_b) || f.src_stride != f.dst_stride))
    {
        set_error("hcv_fft_exec: operands may alias only exactly (same layout, both arrays)");
        return -1;
    }
    const size_t src_extent = src.len ? (f.batch - 1) * f.src_stride + src.len : 0;
    const size_t dst_extent = dst.len ? (f.batch - 1) * f.dst_stride + dst.len : 0;
    if (!dst_extent) return 0;

    bool ok = true;
    void *d[4] = { nullptr, nullptr, nullptr, nullptr };            // src a, src b, dst a, dst b
    auto up = [&](void *&p, const void *host, size_t elems, size_t elem_bytes, bool copy)
    {
        if (!ok) return;
        HCV_API_TRY(hipMalloc(&p, std::max<size_t>(16, elems * elem_bytes)));
        if (ok && copy && elems) HCV_API_TRY(hipMemcpy(p, host, elems * elem_bytes, hipMemcpyHostToDevice));
    };
    up(d[0], call->src_a, src_extent, src.elem, true);
    if (src.two) up(d[1], call->src_b, src_extent, src.elem, true);
    if (in_place)
    {
        d[2] = d[0];
        d[3] = d[1];
    }
    else
    {
        // gaps between strided destination rows keep the caller's bytes
        const bool gaps = f.dst_stride != dst.len && f.batch > 1;
        up(d[2], call->dst_a, dst_extent, dst.elem, gaps);
        if (dst.two) up(d[3], call->dst_b, dst_extent, dst.elem, gaps);
    }
    if (ok)
    {
        f.src_a = d[0]; f.src_b = d[1]; f.dst_a = d[2]; f.dst_b = d[3];
        hipError_t e = hcv::fftx_exec(dev, f, nullptr, &err);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess)
        {
            set_error(err.empty() ? std::string("hcv_fft_exec: ") + hipGetErrorString(e) : err);
            ok = false;
        }
    }
    if (ok) HCV_API_TRY(hipMemcpy(call->dst_a, d[2], dst_extent * dst.elem, hipMemcpyDeviceToHost));
    if (ok && dst.two) HCV_API_TRY(hipMemcpy(call->dst_b, d[3], dst_extent * dst.elem, hipMemcpyDeviceToHost));
    if (!in_place)
    {
        if (d[2]) (void) hipFree(d[2]);
        if (d[3]) (void) hipFree(d[3]);
    }
    if (d[0]) (void) hipFree(d[0]);
    if (d[1]) (void) hipFree(d[1]);
    return ok ? 0 : -1;
}

// ------------------------------------------------------------------------------------------------ spectral IR functions (next row 4)
//
// SpectralFunctions.hpp:365-413 on batches of packed half spectra; kernels in hcv_irx.hip.  The host variants only move bytes.

namespace
{
    hcv::IrCall to_ir(const hcv_ir_call &c)
    {
        hcv::IrCall r;
        r.op = c.op; r.precision = c.precision; r.log2n = c.log2n; r.batch = c.batch;
        r.src_re = c.src_re; r.src_im = c.src_im; r.dst_re = c.dst_re; r.dst_im = c.dst_im;
        r.src_stride = c.src_stride; r.dst_stride = c.dst_stride; r.value = c.value; r.zero_center = c.zero_center;
        return r;
    }

    // device buffer helper for the host-pointer entries
    struct DevBuf
    {
        void *p = nullptr;
        ~DevBuf() { if (p) (void) hipFree(p); }
        bool alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(16, bytes)) == hipSuccess; }
    };
}

extern "C" int hcv_ir_exec_dev(const hcv_ir_call *call, void *stream, int sync)
{
    if (!call)
    {
        set_error("hcv_ir_exec_dev: null descriptor");
        return -1;
    }
    int dev = 0;
    if (!use_default_device(dev)) return -1;
    std::string err;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hcv::irx_exec(dev, to_ir(*call), st, &err);
    if (e == hipSuccess && sync) e = hipStreamSynchronize(st);
    if (e != hipSuccess)
    {
        set_error(err.empty() ? std::string("hcv_ir_exec_dev: ") + hipGetErrorString(e) : err);
        return -1;
    }
    return 0;
}

extern "C" int hcv_ir_exec(const hcv_ir_call *call)
{
    if (!call)
    {
        set_error("hcv_ir_exec: null descriptor");
        return -1;
    }
    int dev = 0;
    if (!use_default_device(dev)) return -1;
    hcv::IrCall c = to_ir(*call);
    std::string err;
    if (!hcv::irx_valid(c, &err))
    {
        set_error(err);
        return -1;
    }
    if (!c.batch) return 0;
    const size_t half = (size_t(1) << c.log2n) >> 1, elem = c.precision == hcv::FX_F32 ? 4 : 8;
    if (!c.src_stride) c.src_stride = half;
    if (!c.dst_stride) c.dst_stride = half;
    const size_t src_extent = (c.batch - 1) * c.src_stride + half, dst_extent = (c.batch - 1) * c.dst_stride + half;
    const bool has_src = c.op != hcv::IR_SPIKE;
    const bool in_place = has_src && call->src_re == call->dst_re && call->src_im == call->dst_im && c.src_stride == c.dst_stride;
    DevBuf sr, si, dr, di;
    bool ok = true;
    if (has_src)
    {
        ok = sr.alloc(src_extent * elem) && si.alloc(src_extent * elem);
        if (ok) HCV_API_TRY(hipMemcpy(sr.p, call->src_re, src_extent * elem, hipMemcpyHostToDevice));
        if (ok) HCV_API_TRY(hipMemcpy(si.p, call->src_im, src_extent * elem, hipMemcpyHostToDevice));
    }
    if (ok && !in_place)
    {
        ok = dr.alloc(dst_extent * elem) && di.alloc(dst_extent * elem);
        const bool gaps = c.dst_stride != half && c.batch > 1;
        if (ok && gaps) HCV_API_TRY(hipMemcpy(dr.p, call->dst_re, dst_extent * elem, hipMemcpyHostToDevice));
        if (ok && gaps) HCV_API_TRY(hipMemcpy(di.p, call->dst_im, dst_extent * elem, hipMemcpyHostToDevice));
    }
    if (!ok)
    {
        if (tlsError.empty()) set_error("hcv_ir_exec: device allocation failed");
        return -1;
    }
    c.src_re = sr.p; c.src_im = si.p;
    c.dst_re = in_place ? sr.p : dr.p;
    c.dst_im = in_place ? si.p : di.p;
    hipError_t e = hcv::irx_exec(dev, c, nullptr, &err);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess)
    {
        set_error(err.empty() ? std::string("hcv_ir_exec: ") + hipGetErrorString(e) : err);
        return -1;
    }
    HCV_API_TRY(hipMemcpy(call->dst_re, c.dst_re, dst_extent * elem, hipMemcpyDeviceToHost));
    if (ok) HCV_API_TRY(hipMemcpy(call->dst_im, c.dst_im, dst_extent * elem, hipMemcpyDeviceToHost));
    return ok ? 0 : -1;
}

// ---- the IR products (SpectralFunctions.hpp:415-436; kernel in hcv_irx.hip)
namespace
{
    hcv::IrProduct to_irp(const hcv_ir_product_call &c)
    {
        hcv::IrProduct r;
        r.op = c.op; r.precision = c.precision; r.batch = c.batch;
        r.count = (c.op & 1) ? c.size >> 1 : c.size;
        r.a_re = c.a_re; r.a_im = c.a_im; r.b_re = c.b_re; r.b_im = c.b_im; r.dst_re = c.dst_re; r.dst_im = c.dst_im;
        r.a_stride = c.a_stride; r.b_stride = c.b_stride; r.dst_stride = c.dst_stride; r.b_broadcast = c.b_broadcast; r.scale = c.scale;
        return r;
    }
}

extern "C" int hcv_ir_product_exec_dev(const hcv_ir_product_call *call, void *stream, int sync)
{
    if (!call)
    {
        set_error("hcv_ir_product_exec_dev: null descriptor");
        return -1;
    }
    int dev = 0;
    if (!use_default_device(dev)) return -1;
    std::string err;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hcv::irp_exec(to_irp(*call), st, &err);
    if (e == hipSuccess && sync) e = hipStreamSynchronize(st);
    if (e != hipSuccess)
    {
        set_error(err.empty() ? std::string("hcv_ir_product_exec_dev: ") + hipGetErrorString(e) : err);
        return -1;
    }
    return 0;
}

extern "C" int hcv_ir_product_exec(const hcv_ir_product_call *call)
{
    if (!call)
    {
        set_error("hcv_ir_product_exec: null descriptor");
        return -1;
    }
    int dev = 0;
    if (!use_default_device(dev)) return -1;
    hcv::IrProduct c = to_irp(*call);
    std::string err;
    if (!hcv::irp_valid(c, &err))
    {
        set_error(err);
        return -1;
    }
    if (!c.batch) return 0;
    const size_t elem = c.precision == hcv::FX_F32 ? 4 : 8;
    if (!c.a_stride) c.a_stride = c.count;
    if (!c.b_stride) c.b_stride = c.count;
    if (!c.dst_stride) c.dst_stride = c.count;
    const size_t a_ext = (c.batch - 1) * c.a_stride + c.count, b_ext = c.b_broadcast ? c.count : (c.batch - 1) * c.b_stride + c.count;
    const size_t d_ext = (c.batch - 1) * c.dst_stride + c.count;
    DevBuf ar, ai, br, bi, dr, di;
    bool ok = ar.alloc(a_ext * elem) && ai.alloc(a_ext * elem) && br.alloc(b_ext * elem) && bi.alloc(b_ext * elem) && dr.alloc(d_ext * elem) && di.alloc(d_ext * elem);
    if (ok) HCV_API_TRY(hipMemcpy(ar.p, call->a_re, a_ext * elem, hipMemcpyHostToDevice));
    if (ok) HCV_API_TRY(hipMemcpy(ai.p, call->a_im, a_ext * elem, hipMemcpyHostToDevice));
    if (ok) HCV_API_TRY(hipMemcpy(br.p, call->b_re, b_ext * elem, hipMemcpyHostToDevice));
    if (ok) HCV_API_TRY(hipMemcpy(bi.p, call->b_im, b_ext * elem, hipMemcpyHostToDevice));
    const bool gaps = c.dst_stride != c.count && c.batch > 1;
    if (ok && gaps) HCV_API_TRY(hipMemcpy(dr.p, call->dst_re, d_ext * elem, hipMemcpyHostToDevice));
    if (ok && gaps) HCV_API_TRY(hipMemcpy(di.p, call->dst_im, d_ext * elem, hipMemcpyHostToDevice));
    if (!ok)
    {
        if (tlsError.empty()) set_error("hcv_ir_product_exec: device allocation failed");
        return -1;
    }
    c.a_re = ar.p; c.a_im = ai.p; c.b_re = br.p; c.b_im = bi.p; c.dst_re = dr.p; c.dst_im = di.p;
    hipError_t e = hcv::irp_exec(c, nullptr, &err);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess)
    {
        set_error(err.empty() ? std::string("hcv_ir_product_exec: ") + hipGetErrorString(e) : err);
        return -1;
    }
    HCV_API_TRY(hipMemcpy(call->dst_re, c.dst_re, d_ext * elem, hipMemcpyDeviceToHost));
    if (ok) HCV_API_TRY(hipMemcpy(call->dst_im, c.dst_im, d_ext * elem, hipMemcpyDeviceToHost));
    return ok ? 0 : -1;
}

// spectral_processor::calc_fft_size_log2 (SpectralProcessor.hpp:231-242) of round(size * time_multiplier)
static unsigned phase_fft_log2(size_t size, double time_multiplier)
{
    const size_t want = (size_t) std::llround((double) size * time_multiplier);
    unsigned count = 0;
    while (count < 63 && (want >> count)) count++;
    if (count && want == (size_t(1) << (count - 1))) return count - 1;
    return count;
}

extern "C" size_t hcv_spectral_phase_size(size_t size, double time_multiplier)
{
    if (size == 1) return 1;
    const unsigned l2 = phase_fft_log2(size, time_multiplier);
    return l2 > (unsigned) hcv::kFxMaxComplexLog2 + 1 ? 0 : size_t(1) << l2;
}

template <class T> static int change_phase(const T *in, size_t size, double phase, double time_multiplier, T *out)
{
    if (!in || !out || !size)
    {
        set_error("hcv_spectral_change_phase: null or empty input");
        return -1;
    }
    if (size == 1)                                             // SpectralProcessor.hpp:195-199
    {
        out[0] = in[0];
        return 0;
    }
    const unsigned log2n = phase_fft_log2(size, time_multiplier);
    if (log2n < 3 || log2n > (unsigned) hcv::kFxMaxComplexLog2 + 1)
    {
        set_error("hcv_spectral_change_phase: fft size out of range (8 .. 2^23)");
        return -1;
    }
    int dev = 0;
    if (!use_default_device(dev)) return -1;
    const size_t n = size_t(1) << log2n, half = n >> 1, take = std::min(size, n);
    const int prec = sizeof(T) == 4 ? hcv::FX_F32 : hcv::FX_F64;
    DevBuf x, re, im, y;
    bool ok = x.alloc(sizeof(T) * n) && re.alloc(sizeof(T) * half) && im.alloc(sizeof(T) * half) && y.alloc(sizeof(T) * n);
    if (!ok)
    {
        set_error("hcv_spectral_change_phase: device allocation failed");
        return -1;
    }
    HCV_API_TRY(hipMemcpy(x.p, in, sizeof(T) * take, hipMemcpyHostToDevice));
    std::string err;
    hipError_t e = hipSuccess;
    if (ok)
    {
        hcv::FxCall f;
        f.precision = prec; f.log2n = log2n; f.batch = 1;
        f.op = hcv::FX_RFFT_ZIP; f.src_a = x.p; f.dst_a = re.p; f.dst_b = im.p; f.in_length = take; f.src_stride = n; f.dst_stride = half;
        e = hcv::fftx_exec(dev, f, nullptr, &err);
        hcv::IrCall c;
        c.op = hcv::IR_PHASE; c.precision = prec; c.log2n = log2n; c.batch = 1; c.value = phase; c.zero_center = 0;
        c.src_re = c.dst_re = re.p; c.src_im = c.dst_im = im.p;
        if (e == hipSuccess) e = hcv::irx_exec(dev, c, nullptr, &err);
        f.op = hcv::FX_RIFFT_ZIP; f.src_a = re.p; f.src_b = im.p; f.dst_a = y.p; f.dst_b = nullptr; f.src_stride = half; f.dst_stride = n;
        if (e == hipSuccess) e = hcv::fftx_exec(dev, f, nullptr, &err);
        if (e == hipSuccess) e = hcv::launch_scale(static_cast<T *>(y.p), (long long) n, (T) 0.5 / (T) n, nullptr);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess)
        {
            set_error(err.empty() ? std::string("hcv_spectral_change_phase: ") + hipGetErrorString(e) : err);
            return -1;
        }
    }
    if (ok) HCV_API_TRY(hipMemcpy(out, y.p, sizeof(T) * n, hipMemcpyDeviceToHost));
    return ok ? 0 : -1;
}

extern "C" int hcv_spectral_change_phase_f32(const float *in, size_t size, double phase, double time_multiplier, float *out)
{
    return change_phase<float>(in, size, phase, time_multiplier, out);
}

extern "C" int hcv_spectral_change_phase_f64(const double *in, size_t size, double phase, double time_multiplier, double *out)
{
    return change_phase<double>(in, size, phase, time_multiplier, out);
}
