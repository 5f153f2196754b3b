// Drop-in for HIRT_Multichannel_Convolution/PartitionedConvolve.h:23-41 — same class, namespace and public
// signatures; the work runs on the MI355X through the C ABI (hisstools_amd.h).  Header-only: link libhisstools_amd.so.
#pragma once

#include "../hisstools_amd.h"
#include "HISSTools_FFT.h"           // (PartitionedConvolve.h:4 of the reference: its callers see the hisstools_* transforms)
#include "ConvolveErrors.h"

#include <cstdint>
#include <stdexcept>

namespace HISSTools
{
    class PartitionedConvolve
    {
    public:

        PartitionedConvolve(uintptr_t maxFFTSize, uintptr_t maxLength, uintptr_t offset, uintptr_t length)
        : mHandle(hcv_partitioned_create(maxFFTSize, maxLength, offset, length))
        {
            if (!mHandle) throw std::runtime_error(hcv_last_error());
        }
        ~PartitionedConvolve() { hcv_partitioned_destroy(mHandle); }

        // Non-moveable and copyable (as the reference)

        PartitionedConvolve(PartitionedConvolve& obj) = delete;
        PartitionedConvolve& operator = (PartitionedConvolve& obj) = delete;
        PartitionedConvolve(PartitionedConvolve&& obj) = delete;
        PartitionedConvolve& operator = (PartitionedConvolve&& obj) = delete;

        ConvolveError setFFTSize(uintptr_t FFTSize) { return static_cast<ConvolveError>(hcv_partitioned_set_fft_size(mHandle, FFTSize)); }
        ConvolveError setLength(uintptr_t length) { return static_cast<ConvolveError>(hcv_partitioned_set_length(mHandle, length)); }
        void setOffset(uintptr_t offset) { hcv_partitioned_set_offset(mHandle, offset); }
        void setResetOffset(intptr_t offset = -1) { hcv_partitioned_set_reset_offset(mHandle, offset); }

        ConvolveError set(const float *input, uintptr_t length) { return static_cast<ConvolveError>(hcv_partitioned_set(mHandle, input, length)); }
        void reset() { hcv_partitioned_reset(mHandle); }

        bool process(const float *in, float *out, uintptr_t numSamples) { return hcv_partitioned_process(mHandle, in, out, numSamples) > 0; }

    private:

        hcv_partitioned *mHandle;
    };
}
