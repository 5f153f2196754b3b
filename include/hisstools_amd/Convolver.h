// Drop-in for HIRT_Multichannel_Convolution/Convolver.h:13-69: recompile callers against this header and link
// libhisstools_amd.so; the matrix then lives on the MI355X (IR spectra resident in HBM).
#pragma once

#include "MemorySwap.h"              // (Convolver.h:4 of the reference makes MemorySwap / thread_lock visible to its callers)
#include "NToMonoConvolve.h"
#include "ConvolveErrors.h"

#include <cstdint>
#include <vector>

namespace HISSTools
{
    class Convolver
    {
    public:

        Convolver(uint32_t numIns, uint32_t numOuts, LatencyMode latency)
        : mHandle(hcv_convolver_create(numIns, numOuts, static_cast<int>(latency)))
        {
            if (!mHandle) throw std::runtime_error(hcv_last_error());
        }

        Convolver(uint32_t numIO, LatencyMode latency) : mHandle(hcv_convolver_create_parallel(numIO, static_cast<int>(latency)))
        {
            if (!mHandle) throw std::runtime_error(hcv_last_error());
        }

        virtual ~Convolver() throw() { hcv_convolver_destroy(mHandle); }

        Convolver(const Convolver&) = delete;
        Convolver& operator = (const Convolver&) = delete;

        // Clear IRs

        void clear(bool resize) { hcv_convolver_clear(mHandle, resize ? 1 : 0); }
        void clear(uint32_t inChan, uint32_t outChan, bool resize) { hcv_convolver_clear_chan(mHandle, inChan, outChan, resize ? 1 : 0); }

        // DSP Engine Reset

        void reset() { hcv_convolver_reset(mHandle); }
        ConvolveError reset(uint32_t inChan, uint32_t outChan) { return static_cast<ConvolveError>(hcv_convolver_reset_chan(mHandle, inChan, outChan)); }

        // Resize and set IR

        ConvolveError resize(uint32_t inChan, uint32_t outChan, uintptr_t impulseLength)
        {
            return static_cast<ConvolveError>(hcv_convolver_resize(mHandle, inChan, outChan, impulseLength));
        }

        ConvolveError set(uint32_t inChan, uint32_t outChan, const float* input, uintptr_t length, bool resize)
        {
            return static_cast<ConvolveError>(hcv_convolver_set_f32(mHandle, inChan, outChan, input, length, resize ? 1 : 0));
        }
        ConvolveError set(uint32_t inChan, uint32_t outChan, const double* input, uintptr_t length, bool resize)
        {
            return static_cast<ConvolveError>(hcv_convolver_set_f64(mHandle, inChan, outChan, input, length, resize ? 1 : 0));
        }

        // DSP

        void process(const double * const* ins, double** outs, size_t numIns, size_t numOuts, size_t numSamples)
        {
            hcv_convolver_process_f64(mHandle, ins, outs, numIns, numOuts, numSamples);
        }
        void process(const float * const*  ins, float** outs, size_t numIns, size_t numOuts, size_t numSamples)
        {
            hcv_convolver_process_f32(mHandle, ins, outs, numIns, numOuts, numSamples);
        }

        // MI355X extension: the underlying handle (device-resident calls, profiling — see hisstools_amd.h)
        hcv_convolver *handle() { return mHandle; }

    private:

        hcv_convolver *mHandle;
    };
}
