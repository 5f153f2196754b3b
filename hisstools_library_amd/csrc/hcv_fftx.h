// The general hisstools_* FFT surface on the device (HISSTools_FFT.h:87-369): complex and real transforms in float
// and double, in place on split data or out of place from / to interleaved ("zipped") samples, plus zip / unzip.
// The convolution engine keeps its own specialised float kernels (hcv_kernels.hip); this file serves the rest of the
// reference's FFT API with one parameterised kernel family.
#pragma once

#include <hip/hip_runtime.h>

#include <string>

namespace hcv
{
    enum FxOp
    {
        FX_FFT = 0,         // hisstools_fft    in-place complex forward            (split -> split)
        FX_IFFT = 1,        // hisstools_ifft   in-place complex inverse            (split -> split)
        FX_RFFT = 2,        // hisstools_rfft   3-arg, in place on unzipped data    (split -> packed split spectrum, x2)
        FX_RIFFT = 3,       // hisstools_rifft  3-arg, in place                     (packed split spectrum -> split samples)
        FX_RFFT_ZIP = 4,    // hisstools_rfft   5-arg: zero-padding unzip + rfft    (samples -> packed split spectrum)
        FX_RIFFT_ZIP = 5,   // hisstools_rifft  4-arg: rifft + zip                  (packed split spectrum -> samples)
        FX_UNZIP = 6,       // hisstools_unzip / hisstools_unzip_zero               (samples -> split)
        FX_ZIP = 7,         // hisstools_zip                                        (split -> samples)
        FX_NUM_OPS
    };

    enum FxPrecision
    {
        FX_F32 = 0,
        FX_F64 = 1,
        FX_F32_TO_F64 = 2   // float samples in, double split out (FX_RFFT_ZIP and FX_UNZIP only)
    };

    constexpr int kFxMaxComplexLog2 = 22;           // four-step with two LDS sub-transforms of <= 2048 points

    // One batched operation on device-resident data.  Strides count elements of the respective array between
    // consecutive transforms.  For split operands a/b are the real/imaginary arrays; for sample operands only a is used.
    struct FxCall
    {
        int op = FX_FFT, precision = FX_F32;
        unsigned log2n = 0;                        // transform size: complex points for FX_FFT/FX_IFFT, real samples otherwise
        size_t batch = 1;
        const void *src_a = nullptr, *src_b = nullptr;
        void *dst_a = nullptr, *dst_b = nullptr;
        size_t src_stride = 0, dst_stride = 0;
        size_t in_length = 0;                      // valid samples per transform for FX_RFFT_ZIP / FX_UNZIP (rest is zero padding)
    };

    // Enqueues the operation on `stream` of `device` (scratch and twiddle tables are cached per device and ordered on
    // the stream, so concurrent use of the big-size path from several streams is serialised by the caller).
    hipError_t fftx_exec(int device, const FxCall &call, hipStream_t stream, std::string *err);
    bool fftx_valid(const FxCall &call, std::string *err);
    const double2 *fftx_twiddles_f64(int device, int log2n, std::string *err);    // exp(-2 pi i m / 2^log2n), m < 2^(log2n-1), cached per device
}
