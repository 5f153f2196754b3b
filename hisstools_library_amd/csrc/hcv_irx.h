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

    bool irx_valid(const IrCall &call, std::string *err);
    hipError_t irx_exec(int device, const IrCall &call, hipStream_t stream, std::string *err);
}
