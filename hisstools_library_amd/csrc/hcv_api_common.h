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
}
