// Drop-in for HIRT_Multichannel_Convolution/NToMonoConvolve.h:13-34.
#pragma once

#include "MonoConvolve.h"
#include "ConvolveErrors.h"

#include <algorithm>
#include <cstdint>
#include <vector>

namespace HISSTools
{
    class NToMonoConvolve
    {
    public:

        NToMonoConvolve(uint32_t input_chans, uintptr_t maxLength, LatencyMode latency)
        : mHandle(hcv_ntomono_create(input_chans, maxLength, static_cast<int>(latency)))
        {
            if (!mHandle) throw std::runtime_error(hcv_last_error());
        }
        ~NToMonoConvolve() { hcv_ntomono_destroy(mHandle); }

        NToMonoConvolve(const NToMonoConvolve&) = delete;
        NToMonoConvolve& operator = (const NToMonoConvolve&) = delete;

        ConvolveError resize(uint32_t inChan, uintptr_t impulse_length) { return static_cast<ConvolveError>(hcv_ntomono_resize(mHandle, inChan, impulse_length)); }
        ConvolveError set(uint32_t inChan, const float *input, uintptr_t impulse_length, bool resize)
        {
            return static_cast<ConvolveError>(hcv_ntomono_set(mHandle, inChan, input, impulse_length, resize ? 1 : 0));
        }
        ConvolveError reset(uint32_t inChan) { return static_cast<ConvolveError>(hcv_ntomono_reset(mHandle, inChan)); }

        void process(const float * const* ins, float *out, float *temp, size_t numSamples, size_t active_in_chans)
        {
            hcv_ntomono_process(mHandle, ins, out, temp, numSamples, active_in_chans);
        }

    private:

        hcv_ntomono *mHandle;
    };
}
