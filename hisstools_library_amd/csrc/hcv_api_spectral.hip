// C ABI, third part (include/hisstools_amd.h): the rest of spectral_processor<T>::convolve / correlate — the real overloads
// in double and the complex overloads in float and double (SpectralProcessor.hpp:164-184, :559-674).  The float real
// overloads keep their own path over the convolution engine's FFT kernels (hcv_api_fft.hip); everything here runs on the
// general FFT surface (hcv_fftx.hip) in the reference's split layout.  Host code sizes the problem and moves bytes; the
// arithmetic (mirrored / padded operands, transforms, bin products, arrangement) is on the device.

#include "../../include/hisstools_amd.h"
#include "hcv_api_common.h"
#include "hcv_fftx.h"

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

using hcv_api::gDefaultDevice;
using hcv_api::set_error;

namespace
{
    enum { EDGE_LINEAR = 0, EDGE_WRAP = 1, EDGE_WRAP_CENTRE = 2, EDGE_FOLD = 3, EDGE_FOLD_REPEAT = 4 };
    using hcv_api::kMaxSpectralLog2;

    struct OpSizes                                              // op_sizes, SpectralProcessor.hpp:318-357
    {
        bool fold;
        size_t size1, size2, mn, mx, linear, fold_copy;
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
        s.fold = mode == EDGE_FOLD || mode == EDGE_FOLD_REPEAT;
        s.size1 = n1;
        s.size2 = n2;
        s.mn = std::min(n1, n2);
        s.mx = std::max(n1, n2);
        s.linear = n1 + n2 - 1;
        s.fold_copy = s.mx + ((s.mn >> 1) << 1);
        s.fft_log2 = spectral_log2(s.fold ? s.fold_copy + (s.mn - 1) : s.linear);
        return s;
    }

    // One operand laid out for the transform (copy_padded + fold + zero fill, SpectralProcessor.hpp:358-392):
    //   dst[0 .. fold)                  mirrored head        signal[fold + off - 1 - j]
    //   dst[fold .. fold + size)        signal = src zero-padded to `size`
    //   dst[fold + size .. + fold)      mirrored tail        signal[size - off - 1 - m]
    //   the rest up to `total`          zeros
    // off = 1 (Fold: the edge sample is not repeated) or 0 (FoldRepeat)
    template <typename T>
    __global__ __launch_bounds__(256) void fold_pad_kernel(T *__restrict__ dst, const T *__restrict__ src, long long n_src, long long size, long long fold,
                                                           int off, long long total)
    {
        const long long j = blockIdx.x * 256LL + threadIdx.x;
        if (j >= total) return;
        long long k = -1;                                       // index into the padded signal
        if (j < fold) k = fold + off - 1 - j;
        else if (j < fold + size) k = j - fold;
        else if (j < size + 2 * fold) k = size - off - 1 - (j - fold - size);
        dst[j] = (k >= 0 && k < n_src) ? src[k] : T(0);
    }

    // io1 = scale * (io1 x io2) or scale * (io1 x conj(io2)), split layout (impl::convolve / impl::correlate,
    // SpectralFunctions.hpp:265-281).  packed: bin 0 of a real spectrum holds (DC, Nyquist), two real products (:63-83).
    template <typename T>
    __global__ __launch_bounds__(256) void split_product_kernel(T *__restrict__ ar, T *__restrict__ ai, const T *__restrict__ br, const T *__restrict__ bi,
                                                                long long n, T scale, int correlate, int packed)
    {
        const long long k = blockIdx.x * 256LL + threadIdx.x;
        if (k >= n) return;
        const T a = ar[k], b = ai[k], c = br[k], d = bi[k];
        if (packed && k == 0)
        {
            ar[0] = scale * (a * c);
            ai[0] = scale * (b * d);
        }
        else if (!correlate)
        {
            ar[k] = scale * (a * c - b * d);
            ai[k] = scale * (b * c + a * d);
        }
        else
        {
            ar[k] = scale * (a * c + b * d);
            ai[k] = scale * (b * c - a * d);
        }
    }

    // op 0: out[o + i] = t[off + i]; op 1: out[o + i] += t[off + i]; op 2: out[o + i] = 0   (copy / wrap / zero, :395-441)
    template <typename T>
    __global__ __launch_bounds__(256) void segment_kernel(T *__restrict__ out, const T *__restrict__ t, long long o_off, long long off, long long n, int op)
    {
        const long long i = blockIdx.x * 256LL + threadIdx.x;
        if (i >= n) return;
        if (op == 0) out[o_off + i] = t[off + i];
        else if (op == 1) out[o_off + i] += t[off + i];
        else out[o_off + i] = T(0);
    }

    inline unsigned blocks_for(size_t n) { return (unsigned) ((n + 255) / 256); }

    // Device scratch of these calls: grow-only buffers cached per device and handed out by position, so that a sequence of
    // calls pays for allocation once (a dozen hipMalloc / hipFree pairs cost more than the transforms of a mid-sized call).
    // One call at a time per process holds the pool.
    struct ScratchPool
    {
        std::mutex mutex;
        std::map<int, std::vector<std::pair<void *, size_t>>> slots;       // device -> (pointer, bytes) by position
    };
    ScratchPool gScratch;

    struct DeviceBuffers
    {
        std::unique_lock<std::mutex> lock;
        std::vector<std::pair<void *, size_t>> *slots = nullptr;
        size_t next = 0;
        explicit DeviceBuffers(int dev) : lock(gScratch.mutex), slots(&gScratch.slots[dev]) {}
        template <typename T>
        T *get(size_t elems, bool &ok)
        {
            if (!ok) return nullptr;
            const size_t bytes = std::max<size_t>(16, elems * sizeof(T));
            if (next >= slots->size()) slots->emplace_back(nullptr, 0);
            std::pair<void *, size_t> &slot = (*slots)[next++];
            if (slot.second < bytes)
            {
                if (slot.first) (void) hipFree(slot.first);
                slot = { nullptr, 0 };
                if (hipMalloc(&slot.first, bytes) != hipSuccess)
                {
                    (void) hipGetLastError();
                    set_error("spectral_processor: device allocation failed");
                    ok = false;
                    return nullptr;
                }
                slot.second = bytes;
            }
            return static_cast<T *>(slot.first);
        }
    };

    template <typename T>
    struct In
    {
        const T *ptr;
        size_t size;
    };

    // arrange_convolve / arrange_correlate (:445-538) as a list of segment operations on one plane (real output, or the
    // real and the imaginary plane of a complex one).  `fft` is the circular size actually used.  The reference's complex
    // instantiation passes `last` where its Split wrap() expects an offset and so reads past the result in the two wrap
    // modes (:401-408 against :429-435); the real overloads' meaning — add the n samples ending at `last` — is used for both.
    template <typename T>
    bool arrange(T *out, const T *t, const OpSizes &s, int mode, bool correlate, size_t fft, bool single, hipStream_t st)
    {
        bool ok = true;
        auto seg = [&](size_t o_off, size_t off, size_t n, int op)
        {
            if (!ok || !n) return;
            hipLaunchKernelGGL(segment_kernel<T>, dim3(blocks_for(n)), dim3(256), 0, st, out, t, (long long) o_off, (long long) off, (long long) n, op);
            ok = hipGetLastError() == hipSuccess;
        };
        auto copy = [&](size_t o_off, size_t off, size_t n) { seg(o_off, off, n, 0); };
        auto wrap = [&](size_t o_off, size_t last, size_t n) { seg(o_off, last - n, n, 1); };
        auto zero = [&](size_t a, size_t b) { if (b > a) seg(a, 0, b - a, 2); };

        if (single)
            copy(0, 0, 1);
        else if (!correlate)
        {
            const size_t min_m1 = s.mn - 1;
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
            const size_t size2_m1 = s.size2 - 1;
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
        if (!ok) set_error("spectral_processor: kernel launch failed");
        return ok;
    }

    template <typename T>
    bool run_fft(int dev, int op, unsigned log2n, const void *sa, const void *sb, void *da, void *db, size_t in_length, hipStream_t st)
    {
        hcv::FxCall c;
        c.op = op;
        c.precision = sizeof(T) == 4 ? hcv::FX_F32 : hcv::FX_F64;
        c.log2n = log2n;
        c.batch = 1;
        c.src_a = sa; c.src_b = sb; c.dst_a = da; c.dst_b = db;
        c.in_length = in_length;
        std::string err;
        const hipError_t e = hcv::fftx_exec(dev, c, st, &err);
        if (e != hipSuccess) set_error(err.empty() ? std::string("spectral_processor: ") + hipGetErrorString(e) : err);
        return e == hipSuccess;
    }

    bool ready(size_t n1, size_t n2, int mode, int &dev, size_t &result)
    {
        result = 0;
        if (!n1 || !n2 || mode < 0 || mode > EDGE_FOLD_REPEAT) return false;
        const OpSizes s = op_sizes(n1, n2, mode);
        if (s.fft_log2 > kMaxSpectralLog2) return false;        // calc_conv_corr_size: nothing is written (:549-560)
        result = mode != EDGE_LINEAR ? s.mx : s.linear;
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

    template <typename T>
    bool upload(T *dst, const T *src, size_t n, bool &ok)
    {
        if (ok && n && hipMemcpy(dst, src, sizeof(T) * n, hipMemcpyHostToDevice) != hipSuccess)
        {
            set_error("spectral_processor: upload failed");
            ok = false;
        }
        return ok;
    }

    // real overloads, binary_op (:616-674)
    template <typename T>
    int real_binary(In<T> in1, In<T> in2, int mode, bool correlate, T *out)
    {
        int dev = 0;
        size_t result = 0;
        if (!ready(in1.size, in2.size, mode, dev, result)) return result == (size_t) -1 ? -1 : 0;
        const OpSizes s = op_sizes(in1.size, in2.size, mode);
        const unsigned log2n = std::max(s.fft_log2, 1u);        // a larger circular size is equivalent: every index below uses it
        const size_t fft = size_t(1) << log2n, half = fft >> 1;
        const bool single = in1.size == 1 && in2.size == 1;
        hipStream_t st = nullptr;
        bool ok = true;
        DeviceBuffers mem(dev);
        T *d1 = mem.get<T>(in1.size, ok), *d2 = mem.get<T>(in2.size, ok), *folded = mem.get<T>(fft, ok);
        T *re1 = mem.get<T>(half, ok), *im1 = mem.get<T>(half, ok), *re2 = mem.get<T>(half, ok), *im2 = mem.get<T>(half, ok);
        T *t = mem.get<T>(fft, ok), *dout = mem.get<T>(result, ok);
        upload(d1, in1.ptr, in1.size, ok);
        upload(d2, in2.ptr, in2.size, ok);
        if (!ok) return -1;

        const T *row1 = d1, *row2 = d2;
        size_t len1 = in1.size, len2 = in2.size;
        if (s.fold)                                             // the longer operand is mirrored at both ends (copy_fold, :368-372)
        {
            const bool first = in1.size >= in2.size;
            const size_t n = first ? in1.size : in2.size, fold = s.mn >> 1;
            hipLaunchKernelGGL(fold_pad_kernel<T>, dim3(blocks_for(n + 2 * fold)), dim3(256), 0, st, folded, first ? d1 : d2, (long long) n, (long long) n,
                               (long long) fold, mode == EDGE_FOLD_REPEAT ? 0 : 1, (long long) (n + 2 * fold));
            (first ? row1 : row2) = folded;
            (first ? len1 : len2) = n + 2 * fold;
        }
        ok = ok && run_fft<T>(dev, hcv::FX_RFFT_ZIP, log2n, row1, nullptr, re1, im1, len1, st);
        ok = ok && run_fft<T>(dev, hcv::FX_RFFT_ZIP, log2n, row2, nullptr, re2, im2, len2, st);
        if (ok)
            hipLaunchKernelGGL(split_product_kernel<T>, dim3(blocks_for(half)), dim3(256), 0, st, re1, im1, re2, im2, (long long) half, T(0.25) / (T) fft,
                               correlate ? 1 : 0, 1);
        ok = ok && run_fft<T>(dev, hcv::FX_RIFFT_ZIP, log2n, re1, im1, t, nullptr, 0, st);
        ok = ok && arrange<T>(dout, t, s, mode, correlate, fft, single, st);
        if (ok && hipMemcpy(out, dout, sizeof(T) * result, hipMemcpyDeviceToHost) != hipSuccess)
        {
            set_error("spectral_processor: download failed");
            ok = false;
        }
        return ok ? 0 : -1;
    }

    // complex overloads, binary_op (:559-614): each operand is (real, imaginary) with possibly different lengths, both
    // zero-padded to the longer one
    template <typename T>
    int complex_binary(In<T> r1, In<T> i1, In<T> r2, In<T> i2, int mode, bool correlate, T *r_out, T *i_out)
    {
        const size_t size1 = std::max(r1.size, i1.size), size2 = std::max(r2.size, i2.size);
        int dev = 0;
        size_t result = 0;
        if (!ready(size1, size2, mode, dev, result)) return result == (size_t) -1 ? -1 : 0;
        const OpSizes s = op_sizes(size1, size2, mode);
        const unsigned log2n = std::max(s.fft_log2, 1u);
        const size_t fft = size_t(1) << log2n;
        const bool single = size1 == 1 && size2 == 1;
        // two single samples: the reference's shortcut is the plain complex product for correlate as well (:590-595)
        if (single) correlate = false;
        hipStream_t st = nullptr;
        bool ok = true;
        DeviceBuffers mem(dev);
        const In<T> ins[4] = { r1, i1, r2, i2 };
        T *raw[4], *plane[4];
        for (int k = 0; k < 4; k++)
        {
            raw[k] = mem.get<T>(ins[k].size, ok);
            plane[k] = mem.get<T>(fft, ok);
            upload(raw[k], ins[k].ptr, ins[k].size, ok);
        }
        T *dr = mem.get<T>(result, ok), *di = mem.get<T>(result, ok);
        if (!ok) return -1;

        const bool fold1 = s.fold && size1 >= size2, fold2 = s.fold && !fold1;          // :562-566
        const size_t fold_size = s.mn >> 1;
        const int off = mode == EDGE_FOLD_REPEAT ? 0 : 1;
        for (int k = 0; k < 4; k++)
        {
            const size_t size = k < 2 ? size1 : size2;
            const size_t fold = (k < 2 ? fold1 : fold2) ? fold_size : 0;
            hipLaunchKernelGGL(fold_pad_kernel<T>, dim3(blocks_for(fft)), dim3(256), 0, st, plane[k], raw[k], (long long) ins[k].size, (long long) size,
                               (long long) fold, off, (long long) fft);
        }
        ok = ok && run_fft<T>(dev, hcv::FX_FFT, log2n, plane[0], plane[1], plane[0], plane[1], 0, st);
        ok = ok && run_fft<T>(dev, hcv::FX_FFT, log2n, plane[2], plane[3], plane[2], plane[3], 0, st);
        if (ok)
            hipLaunchKernelGGL(split_product_kernel<T>, dim3(blocks_for(fft)), dim3(256), 0, st, plane[0], plane[1], plane[2], plane[3], (long long) fft,
                               T(1) / (T) fft, correlate ? 1 : 0, 0);
        ok = ok && run_fft<T>(dev, hcv::FX_IFFT, log2n, plane[0], plane[1], plane[0], plane[1], 0, st);
        ok = ok && arrange<T>(dr, plane[0], s, mode, correlate, fft, single, st);
        ok = ok && arrange<T>(di, plane[1], s, mode, correlate, fft, single, st);
        if (ok && (hipMemcpy(r_out, dr, sizeof(T) * result, hipMemcpyDeviceToHost) != hipSuccess ||
                   hipMemcpy(i_out, di, sizeof(T) * result, hipMemcpyDeviceToHost) != hipSuccess))
        {
            set_error("spectral_processor: download failed");
            ok = false;
        }
        return ok ? 0 : -1;
    }
}

int hcv_api::spectral_real_general_f32(const float *in1, size_t size1, const float *in2, size_t size2, int mode, bool correlate, float *out)
{
    return real_binary<float>({ in1, size1 }, { in2, size2 }, mode, correlate, out);
}

extern "C" int hcv_spectral_convolve_f64(const double *in1, size_t size1, const double *in2, size_t size2, int mode, double *out)
{
    return real_binary<double>({ in1, size1 }, { in2, size2 }, mode, false, out);
}

extern "C" int hcv_spectral_correlate_f64(const double *in1, size_t size1, const double *in2, size_t size2, int mode, double *out)
{
    return real_binary<double>({ in1, size1 }, { in2, size2 }, mode, true, out);
}

extern "C" int hcv_spectral_convolve_complex_f32(const float *r1, size_t nr1, const float *i1, size_t ni1, const float *r2, size_t nr2, const float *i2,
                                                 size_t ni2, int mode, float *r_out, float *i_out)
{
    return complex_binary<float>({ r1, nr1 }, { i1, ni1 }, { r2, nr2 }, { i2, ni2 }, mode, false, r_out, i_out);
}

extern "C" int hcv_spectral_correlate_complex_f32(const float *r1, size_t nr1, const float *i1, size_t ni1, const float *r2, size_t nr2, const float *i2,
                                                  size_t ni2, int mode, float *r_out, float *i_out)
{
    return complex_binary<float>({ r1, nr1 }, { i1, ni1 }, { r2, nr2 }, { i2, ni2 }, mode, true, r_out, i_out);
}

extern "C" int hcv_spectral_convolve_complex_f64(const double *r1, size_t nr1, const double *i1, size_t ni1, const double *r2, size_t nr2, const double *i2,
                                                 size_t ni2, int mode, double *r_out, double *i_out)
{
    return complex_binary<double>({ r1, nr1 }, { i1, ni1 }, { r2, nr2 }, { i2, ni2 }, mode, false, r_out, i_out);
}

extern "C" int hcv_spectral_correlate_complex_f64(const double *r1, size_t nr1, const double *i1, size_t ni1, const double *r2, size_t nr2, const double *i2,
                                                  size_t ni2, int mode, double *r_out, double *i_out)
{
    return complex_binary<double>({ r1, nr1 }, { i1, ni1 }, { r2, nr2 }, { i2, ni2 }, mode, true, r_out, i_out);
}
