// State shared by the two translation units of the C ABI (hcv_api.hip: the convolution classes; hcv_api_fft.hip: the FFT,
// spectral_processor and spectral IR entry points).
#pragma once

#include <hip/hip_runtime.h>

#include <string>

namespace hcv_api
{
    extern thread_local std::string tlsError;     // text behind hcv_last_error()
    extern int gDefaultDevice;                    // hcv_set_default_device(), -1 = the current HIP device
    inline void set_error(const std::string &s) { tlsError = s; }

    // spectral_processor: largest circular size of the one-shot convolution / correlation (complex transforms of the general
    // FFT surface reach 2^22); the float real overloads use the convolution engine's kernels up to 2^20 and the general path
    // above (hcv_api_spectral.hip)
    constexpr unsigned kMaxSpectralLog2 = 22;
    int spectral_real_general_f32(const float *in1, size_t size1, const float *in2, size_t size2, int mode, bool correlate, float *out);
}
