// TEST INFRASTRUCTURE — not part of the product path.
//
// C-ABI driver around the unmodified reference's one-shot spectral convolution / correlation
// (SpectralProcessor.hpp:164-184, 445-674; SpectralFunctions.hpp:265-281, 415-436) — the first "next" row of
// SURVEY.md §8f.  Built by oracle/Makefile into oracle/_ref/libhisstools_ref_spectral.so with -mavx: the reference's
// float path does not compile for plain SSE2 (SpectralFunctions.hpp:47-60 instantiates SIMDType<float, 2>, which has no
// arithmetic operators), which is a property of the reference, not of this driver.

#include "HISSTools_FFT/HISSTools_FFT.h"
#include "SpectralProcessor.hpp"

#include <cstdint>

// FFT setup large enough for every edge mode of these sizes (fold modes need up to ~2x the linear size); building the
// reference's 2^24 tables on every call would dominate the test time
static uintptr_t table_size(uintptr_t n1, uintptr_t n2)
{
    uintptr_t need = 4 * (n1 + n2) + 16, size = 32;
    while (size < need) size <<= 1;
    return size;
}

extern "C"
{
    // mode: 0 Linear, 1 Wrap, 2 WrapCentre, 3 Fold, 4 FoldRepeat (SpectralProcessor.hpp:22)
    uintptr_t ref_spectral_size(uintptr_t n1, uintptr_t n2, int mode)
    {
        spectral_processor<float> sp(table_size(n1, n2));
        return sp.convolved_size(n1, n2, static_cast<spectral_processor<float>::EdgeMode>(mode));
    }

    void ref_spectral_convolve_f32(const float *in1, uintptr_t n1, const float *in2, uintptr_t n2, int mode, float *out)
    {
        spectral_processor<float> sp(table_size(n1, n2));
        sp.convolve(out, { in1, n1 }, { in2, n2 }, static_cast<spectral_processor<float>::EdgeMode>(mode));
    }

    void ref_spectral_correlate_f32(const float *in1, uintptr_t n1, const float *in2, uintptr_t n2, int mode, float *out)
    {
        spectral_processor<float> sp(table_size(n1, n2));
        sp.correlate(out, { in1, n1 }, { in2, n2 }, static_cast<spectral_processor<float>::EdgeMode>(mode));
    }
}
