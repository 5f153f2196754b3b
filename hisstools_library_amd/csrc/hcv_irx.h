// Spectral IR functions on the device (SpectralFunctions.hpp:365-413): ir_copy, ir_spike, ir_delay, ir_time_reverse and
// ir_phase (minimum / linear / maximum / interpolated phase) on batches of packed half spectra, float or double.
#pragma once

#include <hip/hip_runtime.h>

#include <string>

namespace hcv
{
    enum IrOp { IR_COPY = 0, IR_SPIKE = 1, IR_DELAY = 2, IR_TIME_REVERSE = 3, IR_PHASE = 4, IR_NUM_OPS };

    // One batched operation on device-resident packed half spectra of 2^log2n real samples (2^(log2n-1) values per
    // array, bin 0 = (DC, Nyquist)).  Strides count elements between consecutive spectra.  src may equal dst.
    struct IrCall
    {
        int op = IR_COPY, precision = 0;           // precision: FX_F32 / FX_F64
        unsigned log2n = 0;
        size_t batch = 1;
        const void *src_re = nullptr, *src_im = nullptr;   // unused by IR_SPIKE
        void *dst_re = nullptr, *dst_im = nullptr;
        size_t src_stride = 0, dst_stride = 0;
        double value = 0.0;                        // spike position | delay (samples) | phase (0 minimum .. 0.5 linear .. 1 maximum)
        int zero_center = 0;                       // IR_PHASE only
    };

    // x[i] *= scale (scale_vector, SpectralProcessor.hpp:246-253)
    hipError_t launch_scale(float *x, long long n, float scale, hipStream_t stream);
    hipError_t launch_scale(double *x, long long n, double scale, hipStream_t stream);

    // The IR products (SpectralFunctions.hpp:415-436): dst = scale * a * b (convolve) or scale * a * conj(b) (correlate), batched.
    // `count` = values per array: fft_size for the complex forms, fft_size / 2 for the real forms (bin 0 = (DC, Nyquist): two real products).
    enum IrProductOp { IRP_CONVOLVE_COMPLEX = 0, IRP_CONVOLVE_REAL = 1, IRP_CORRELATE_COMPLEX = 2, IRP_CORRELATE_REAL = 3, IRP_NUM_OPS };
    struct IrProduct
    {
        int op = IRP_CONVOLVE_COMPLEX, precision = 0;
        size_t count = 0, batch = 1;
        const void *a_re = nullptr, *a_im = nullptr, *b_re = nullptr, *b_im = nullptr;
        void *dst_re = nullptr, *dst_im = nullptr;
        size_t a_stride = 0, b_stride = 0, dst_stride = 0;      // elements between consecutive spectra, 0 = dense; b_stride may be given as
        int b_broadcast = 0;                                    // ... one spectrum for the whole batch (b_broadcast != 0)
        double scale = 1.0;
    };
    bool irp_valid(const IrProduct &call, std::string *err);
    hipError_t irp_exec(const IrProduct &call, hipStream_t stream, std::string *err);

    bool irx_valid(const IrCall &call, std::string *err);
    hipError_t irx_exec(int device, const IrCall &call, hipStream_t stream, std::string *err);
}
