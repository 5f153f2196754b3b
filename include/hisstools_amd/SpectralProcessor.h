// Drop-in for the real convolve / correlate overloads of SpectralProcessor.hpp:13-184 (spectral_processor<float>):
// same class name, EdgeMode enum, in_ptr helper and method signatures; the transforms and products run on the GPU
// through the C ABI (hisstools_amd.h: hcv_spectral_*).  The complex overloads, change_phase and the raw fft/rfft
// members are not provided here.
#pragma once

#include "../hisstools_amd.h"

#include <cstdint>
#include <type_traits>

template <typename T>
class spectral_processor
{
    static_assert(std::is_same<T, float>::value, "the MI355X engine provides spectral_processor<float>");

public:

    enum class EdgeMode { Linear, Wrap, WrapCentre, Fold, FoldRepeat };

    struct in_ptr
    {
        in_ptr(const T* ptr, uintptr_t size) : m_ptr(ptr), m_size(size) {}

        const T* m_ptr;
        const uintptr_t m_size;
    };

    spectral_processor(uintptr_t max_fft_size = 32768) : m_max_fft_size(max_fft_size) {}

    void set_max_fft_size(uintptr_t size) { m_max_fft_size = size; }
    uintptr_t max_fft_size() const { return m_max_fft_size; }

    void convolve(T *output, in_ptr in1, in_ptr in2, EdgeMode mode)
    {
        if (convolved_size(in1.m_size, in2.m_size, mode))
            hcv_spectral_convolve_f32(in1.m_ptr, in1.m_size, in2.m_ptr, in2.m_size, static_cast<int>(mode), output);
    }

    void correlate(T *output, in_ptr in1, in_ptr in2, EdgeMode mode)
    {
        if (correlated_size(in1.m_size, in2.m_size, mode))
            hcv_spectral_correlate_f32(in1.m_ptr, in1.m_size, in2.m_ptr, in2.m_size, static_cast<int>(mode), output);
    }

    uintptr_t convolved_size(uintptr_t size1, uintptr_t size2, EdgeMode mode) const
    {
        // 0 when the FFT this needs exceeds max_fft_size (calc_conv_corr_size, SpectralProcessor.hpp:549-560)
        const uintptr_t needed = hcv_spectral_size(size1, size2, static_cast<int>(mode));
        if (!needed) return 0;
        const bool fold = mode == EdgeMode::Fold || mode == EdgeMode::FoldRepeat;
        const uintptr_t mn = size1 < size2 ? size1 : size2, mx = size1 < size2 ? size2 : size1;
        const uintptr_t span = fold ? mx + ((mn >> 1) << 1) + (mn - 1) : size1 + size2 - 1;
        uintptr_t fft = 1;
        while (fft < span) fft <<= 1;
        return fft > m_max_fft_size ? 0 : needed;
    }

    uintptr_t correlated_size(uintptr_t size1, uintptr_t size2, EdgeMode mode) const { return convolved_size(size1, size2, mode); }

private:

    uintptr_t m_max_fft_size;
};
