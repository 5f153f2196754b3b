// Drop-in for SpectralProcessor.hpp:13-242: spectral_processor<float> and <double> with the real and the complex convolve /
// correlate overloads, change_phase and the transform members — same class name, EdgeMode enum, in_ptr helper and method
// signatures; the transforms, products and phase manipulation run on the GPU through the C ABI (hisstools_amd.h:
// hcv_spectral_*, hcv_fft_exec).
#pragma once

#include "../hisstools_amd.h"
#include "HISSTools_FFT.h"

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

    using Split = ::Split<T>;                                                   // FFT_SPLIT_COMPLEX_F / _D

    // Transforms (SpectralProcessor.hpp:117-160); size-1 transforms are the reference's special cases
    void fft(Split& io, uintptr_t fft_size_log2) { if (fft_size_log2) hisstools_fft(setup(), &io, fft_size_log2); }
    void ifft(Split& io, uintptr_t fft_size_log2) { if (fft_size_log2) hisstools_ifft(setup(), &io, fft_size_log2); }
    void rfft(Split& io, uintptr_t fft_size_log2) { if (fft_size_log2) hisstools_rfft(setup(), &io, fft_size_log2); }
    void rifft(Split& io, uintptr_t fft_size_log2) { if (fft_size_log2) hisstools_rifft(setup(), &io, fft_size_log2); }

    void rfft(Split& output, const T *input, uintptr_t size, uintptr_t fft_size_log2)
    {
        if (!fft_size_log2)
        {
            output.realp[0] = input[0] * T(2);
            output.imagp[0] = T(0);
        }
        else
            hisstools_rfft(setup(), input, &output, size, fft_size_log2);
    }

    void rifft(T *output, Split& input, uintptr_t fft_size_log2)
    {
        if (!fft_size_log2)
            output[0] = input.realp[0];
        else
            hisstools_rifft(setup(), &input, output, fft_size_log2);
    }

    // Convolution / correlation (SpectralProcessor.hpp:164-184): complex, then real
    void convolve(T *r_out, T *i_out, in_ptr r_in1, in_ptr i_in1, in_ptr r_in2, in_ptr i_in2, EdgeMode mode)
    {
        if (convolved_size(larger(r_in1, i_in1), larger(r_in2, i_in2), mode)) complex_op(false, r_out, i_out, r_in1, i_in1, r_in2, i_in2, mode);
    }

    void convolve(T *output, in_ptr in1, in_ptr in2, EdgeMode mode)
    {
        if (convolved_size(in1.m_size, in2.m_size, mode)) real_op(false, output, in1, in2, mode);
    }

    void correlate(T *r_out, T *i_out, in_ptr r_in1, in_ptr i_in1, in_ptr r_in2, in_ptr i_in2, EdgeMode mode)
    {
        if (correlated_size(larger(r_in1, i_in1), larger(r_in2, i_in2), mode)) complex_op(true, r_out, i_out, r_in1, i_in1, r_in2, i_in2, mode);
    }

    void correlate(T *output, in_ptr in1, in_ptr in2, EdgeMode mode)
    {
        if (correlated_size(in1.m_size, in2.m_size, mode)) real_op(true, output, in1, in2, mode);
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

    static uintptr_t larger(in_ptr a, in_ptr b) { return a.m_size > b.m_size ? a.m_size : b.m_size; }
    static Setup<T> *setup() { return nullptr; }                                // twiddle tables live on the device, per size

    static void real_op(bool corr, float *o, in_ptr a, in_ptr b, EdgeMode m)
    {
        (void) (corr ? hcv_spectral_correlate_f32 : hcv_spectral_convolve_f32)(a.m_ptr, a.m_size, b.m_ptr, b.m_size, static_cast<int>(m), o);
    }
    static void real_op(bool corr, double *o, in_ptr a, in_ptr b, EdgeMode m)
    {
        (void) (corr ? hcv_spectral_correlate_f64 : hcv_spectral_convolve_f64)(a.m_ptr, a.m_size, b.m_ptr, b.m_size, static_cast<int>(m), o);
    }
    static void complex_op(bool corr, float *ro, float *io, in_ptr r1, in_ptr i1, in_ptr r2, in_ptr i2, EdgeMode m)
    {
        (void) (corr ? hcv_spectral_correlate_complex_f32 : hcv_spectral_convolve_complex_f32)(r1.m_ptr, r1.m_size, i1.m_ptr, i1.m_size, r2.m_ptr, r2.m_size,
                                                                                              i2.m_ptr, i2.m_size, static_cast<int>(m), ro, io);
    }
    static void complex_op(bool corr, double *ro, double *io, in_ptr r1, in_ptr i1, in_ptr r2, in_ptr i2, EdgeMode m)
    {
        (void) (corr ? hcv_spectral_correlate_complex_f64 : hcv_spectral_convolve_complex_f64)(r1.m_ptr, r1.m_size, i1.m_ptr, i1.m_size, r2.m_ptr, r2.m_size,
                                                                                              i2.m_ptr, i2.m_size, static_cast<int>(m), ro, io);
    }

    static void change_phase_impl(float *o, const float *i, uintptr_t n, double p, double m) { (void) hcv_spectral_change_phase_f32(i, n, p, m, o); }
    static void change_phase_impl(double *o, const double *i, uintptr_t n, double p, double m) { (void) hcv_spectral_change_phase_f64(i, n, p, m, o); }

    uintptr_t m_max_fft_size;
};
