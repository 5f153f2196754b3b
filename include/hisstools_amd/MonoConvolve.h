// Drop-in for HIRT_Multichannel_Convolution/MonoConvolve.h:14-48 (LatencyMode in the global namespace, as there).
#pragma once

#include "../hisstools_amd.h"
#include "PartitionedConvolve.h"
#include "TimeDomainConvolve.h"
#include "ConvolveErrors.h"
#include "MemorySwap.h"

#include <cstdint>
#include <stdexcept>

enum LatencyMode
{
    kLatencyZero,
    kLatencyShort,
    kLatencyMedium,
} ;

namespace HISSTools
{
    class MonoConvolve
    {
    public:

        MonoConvolve(uintptr_t maxLength, LatencyMode latency) : mHandle(hcv_mono_create(maxLength, static_cast<int>(latency)))
        {
            if (!mHandle) throw std::runtime_error(hcv_last_error());
        }

        // throws std::runtime_error("invalid FFT size or order" / "no valid FFT sizes given") like MonoConvolve.cpp:207-229
        MonoConvolve(uintptr_t maxLength, bool zeroLatency, uint32_t A, uint32_t B = 0, uint32_t C = 0, uint32_t D = 0) : mHandle(nullptr)
        {
            setPartitions(maxLength, zeroLatency, A, B, C, D);
        }

        ~MonoConvolve() { if (mHandle) hcv_mono_destroy(mHandle); }

        // Moveable but not copyable

        MonoConvolve(MonoConvolve& obj) = delete;
        MonoConvolve& operator = (MonoConvolve& obj) = delete;
        MonoConvolve(MonoConvolve&& obj) : mHandle(obj.mHandle) { obj.mHandle = nullptr; if (mHandle) hcv_mono_reset(mHandle); }
        MonoConvolve& operator = (MonoConvolve&& obj)
        {
            if (this != &obj)
            {
                if (mHandle) hcv_mono_destroy(mHandle);
                mHandle = obj.mHandle;
                obj.mHandle = nullptr;
                if (mHandle) hcv_mono_reset(mHandle);          // moved-to objects restart (MonoConvolve.cpp:49-78)
            }
            return *this;
        }

        void setResetOffset(intptr_t offset = -1) { hcv_mono_set_reset_offset(mHandle, offset); }

        ConvolveError resize(uintptr_t length) { return static_cast<ConvolveError>(hcv_mono_resize(mHandle, length)); }
        ConvolveError set(const float *input, uintptr_t length, bool requestResize)
        {
            return static_cast<ConvolveError>(hcv_mono_set(mHandle, input, length, requestResize ? 1 : 0));
        }
        ConvolveError reset() { return static_cast<ConvolveError>(hcv_mono_reset(mHandle)); }

        void process(const float *in, float *temp, float *out, uintptr_t numSamples, bool accumulate = false)
        {
            hcv_mono_process(mHandle, in, temp, out, numSamples, accumulate ? 1 : 0);
        }

        void setPartitions(uintptr_t maxLength, bool zeroLatency, uint32_t A, uint32_t B = 0, uint32_t C = 0, uint32_t D = 0)
        {
            char err[128] = { 0 };
            hcv_mono *h = hcv_mono_create_custom(maxLength, zeroLatency ? 1 : 0, A, B, C, D, err, sizeof(err));
            if (!h) throw std::runtime_error(err[0] ? err : hcv_last_error());
            if (mHandle) hcv_mono_destroy(mHandle);
            mHandle = h;
        }

    private:

        hcv_mono *mHandle;
    };
}
