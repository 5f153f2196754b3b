// Drop-in for HIRT_Multichannel_Convolution/TimeDomainConvolve.h:15-31.
#pragma once

#include "../hisstools_amd.h"
#include "ConvolveErrors.h"

#include <cstdint>
#include <stdexcept>

namespace HISSTools
{
    class TimeDomainConvolve
    {
    public:

        TimeDomainConvolve(uintptr_t offset, uintptr_t length) : mHandle(hcv_timedomain_create(offset, length))
        {
            if (!mHandle) throw std::runtime_error(hcv_last_error());
        }
        ~TimeDomainConvolve() { hcv_timedomain_destroy(mHandle); }

        TimeDomainConvolve(TimeDomainConvolve& obj) = delete;
        TimeDomainConvolve& operator = (TimeDomainConvolve& obj) = delete;
        TimeDomainConvolve(TimeDomainConvolve&& obj) = delete;
        TimeDomainConvolve& operator = (TimeDomainConvolve&& obj) = delete;

        ConvolveError setLength(uintptr_t length) { return static_cast<ConvolveError>(hcv_timedomain_set_length(mHandle, length)); }
        void setOffset(uintptr_t offset) { hcv_timedomain_set_offset(mHandle, offset); }

        ConvolveError set(const float *input, uintptr_t length) { return static_cast<ConvolveError>(hcv_timedomain_set(mHandle, input, length)); }
        void reset() { hcv_timedomain_reset(mHandle); }

        bool process(const float *in, float *out, uintptr_t numSamples) { return hcv_timedomain_process(mHandle, in, out, numSamples) > 0; }

    private:

        hcv_timedomain *mHandle;
    };
}
