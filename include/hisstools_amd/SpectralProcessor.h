// Drop-in for SpectralProcessor.hpp:13-242: the real convolve / correlate overloads (spectral_processor<float>) and
// change_phase (float and double) — same class name, EdgeMode enum, in_ptr helper and method signatures; the transforms,
// products and phase manipulation run on the GPU through the C ABI (hisstools_amd.h: hcv_spectral_*).  The complex
// convolve / correlate overloads and the raw fft/rfft members are not provided here.
#pragma once

#include "../hisstools_amd.h"

#include <cstdint>
#include <type_traits>

template <typename T>
class spectral_processor
{
    static_assert(std::is_same<T, float>::value || std::is_same<T, double>::value, "spectral_processor<float> or <double>");

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
        static_assert(std::is_same<T, float>::value, "convolve / correlate: the MI355X engine provides spectral_processor<float>");
        if (convolved_size(in1.m_size, in2.m_size, mode))
            hcv_spectral_convolve_f32(in1.m_ptr, in1.m_size, in2.m_ptr, in2.m_size, static_cast<int>(mode), output);
    }

    void correlate(T *output, in_ptr in1, in_ptr in2, EdgeMode mode)
    {
        static_assert(std::is_same<T, float>::value, "convolve / correlate: the MI355X engine provides spectral_processor<float>");
        if (correlated_size(in1.m_size, in2.m_size, mode))
            hcv_spectral_correlate_f32(in1.m_ptr, in1.m_size, in2.m_ptr, in2.m_size, static_cast<int>(mode), output);
    }

    // SpectralProcessor.hpp:188-208: `output` receives 2^calc_fft_size_log2(round(size * time_multiplier)) samples
    void change_phase(T *output, const T *input, uintptr_t size, double phase, double time_multiplier = 1.0)
    {
        change_phase_impl(output, input, size, phase, time_multiplier);
    }

    static uintptr_t calc_fft_size_log2(uintptr_t size)                        // SpectralProcessor.hpp:231-242
    {
        uintptr_t count = 0;
        while (size >> count) count++;
        return (count && size == uintptr_t(1) << (count - 1U)) ? count - uintptr_t(1) : count;
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

    static void change_phase_impl(float *o, const float *i, uintptr_t n, double p, double m) { (void) hcv_spectral_change_phase_f32(i, n, p, m, o); }
    static void change_phase_impl(double *o, const double *i, uintptr_t n, double p, double m) { (void) hcv_spectral_change_phase_f64(i, n, p, m, o); }

    uintptr_t m_max_fft_size;
};
