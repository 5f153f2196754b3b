// TEST INFRASTRUCTURE — not part of the product path.
//
// C-ABI driver around the unmodified reference's one-shot spectral convolution / correlation
// (SpectralProcessor.hpp:164-184, 445-674; SpectralFunctions.hpp:265-281, 415-436) — the first "next" row of
// SURVEY.md §8f.  Built by oracle/Makefile into oracle/_ref/libhisstools_ref_spectral.so with -mavx: the reference's
// float path does not compile for plain SSE2 (SpectralFunctions.hpp:47-60 instantiates SIMDType<float, 2>, which has no
// arithmetic operators), which is a property of the reference, not of this driver.

#include "HISSTools_FFT/HISSTools_FFT.h"
#include "SpectralProcessor.hpp"

#include <cmath>
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

    // the real overloads in double, and the complex overloads in both precisions (SpectralProcessor.hpp:164-167, 176-179).
    // NOTE: in the two wrap modes the reference's complex instantiation reads past the result (its Split wrap() takes an
    // offset where arrange_* passes an end position, :401-408); callers compare those modes against the real overloads.
    void ref_spectral_convolve_f64(const double *in1, uintptr_t n1, const double *in2, uintptr_t n2, int mode, double *out)
    {
        spectral_processor<double> sp(table_size(n1, n2));
        sp.convolve(out, { in1, n1 }, { in2, n2 }, static_cast<spectral_processor<double>::EdgeMode>(mode));
    }

    void ref_spectral_correlate_f64(const double *in1, uintptr_t n1, const double *in2, uintptr_t n2, int mode, double *out)
    {
        spectral_processor<double> sp(table_size(n1, n2));
        sp.correlate(out, { in1, n1 }, { in2, n2 }, static_cast<spectral_processor<double>::EdgeMode>(mode));
    }

#define REF_COMPLEX(T, SUF)                                                                                                                  \
    void ref_spectral_convolve_complex_##SUF(const T *r1, uintptr_t nr1, const T *i1, uintptr_t ni1, const T *r2, uintptr_t nr2, const T *i2,  \
                                             uintptr_t ni2, int mode, T *r_out, T *i_out)                                                    \
    {                                                                                                                                        \
        spectral_processor<T> sp(table_size(nr1 > ni1 ? nr1 : ni1, nr2 > ni2 ? nr2 : ni2));                                                  \
        sp.convolve(r_out, i_out, { r1, nr1 }, { i1, ni1 }, { r2, nr2 }, { i2, ni2 }, static_cast<spectral_processor<T>::EdgeMode>(mode));   \
    }                                                                                                                                        \
    void ref_spectral_correlate_complex_##SUF(const T *r1, uintptr_t nr1, const T *i1, uintptr_t ni1, const T *r2, uintptr_t nr2, const T *i2, \
                                              uintptr_t ni2, int mode, T *r_out, T *i_out)                                                   \
    {                                                                                                                                        \
        spectral_processor<T> sp(table_size(nr1 > ni1 ? nr1 : ni1, nr2 > ni2 ? nr2 : ni2));                                                  \
        sp.correlate(r_out, i_out, { r1, nr1 }, { i1, ni1 }, { r2, nr2 }, { i2, ni2 }, static_cast<spectral_processor<T>::EdgeMode>(mode));  \
    }
    REF_COMPLEX(float, f32)
    REF_COMPLEX(double, f64)
#undef REF_COMPLEX

    // ---------------------------------------------------------------- spectral IR functions (SpectralFunctions.hpp:365-413) and
    // spectral_processor::change_phase (SpectralProcessor.hpp:188-208): fourth "next" row.  op: 0 copy, 1 spike, 2 delay,
    // 3 time_reverse, 4 phase.  In-place calls (out == in) are what the reference's own callers make.
#define REF_IR(SFX, T, SPLIT, SETUP)                                                                                     \
    void ref_ir_##SFX(int op, T *ro, T *io, const T *ri, const T *ii, uintptr_t fft_size, double value, int zero_center)   \
    {                                                                                                                    \
        SPLIT out(ro, io), in(const_cast<T *>(ri), const_cast<T *>(ii));                                                 \
        if (op == 0) ir_copy(&out, &in, fft_size);                                                                       \
        else if (op == 1) ir_spike(&out, fft_size, value);                                                               \
        else if (op == 2) ir_delay(&out, &in, fft_size, value);                                                          \
        else if (op == 3) ir_time_reverse(&out, &in, fft_size);                                                          \
        else                                                                                                             \
        {                                                                                                                \
            uintptr_t log2n = 0;                                                                                         \
            while ((uintptr_t(1) << log2n) < fft_size) log2n++;                                                          \
            SETUP setup;                                                                                                 \
            hisstools_create_setup(&setup, log2n);                                                                       \
            ir_phase(setup, &out, &in, fft_size, value, zero_center != 0);                                               \
            hisstools_destroy_setup(setup);                                                                              \
        }                                                                                                                \
    }                                                                                                                    \
    uintptr_t ref_change_phase_##SFX(const T *in, uintptr_t size, double phase, double time_multiplier, T *out)          \
    {                                                                                                                    \
        uintptr_t log2n = spectral_processor<T>::calc_fft_size_log2((uintptr_t) std::round(size * time_multiplier));     \
        spectral_processor<T> sp(uintptr_t(1) << (log2n > 5 ? log2n : 5));                                               \
        sp.change_phase(out, in, size, phase, time_multiplier);                                                          \
        return size == 1 ? 1 : uintptr_t(1) << log2n;                                                                    \
    }
    REF_IR(f32, float, FFT_SPLIT_COMPLEX_F, FFT_SETUP_F)
    REF_IR(f64, double, FFT_SPLIT_COMPLEX_D, FFT_SETUP_D)
#undef REF_IR

    // ---------------------------------------------------------------- the IR products (SpectralFunctions.hpp:415-436).  op: 0 convolve_complex,
    // 1 convolve_real, 2 correlate_complex, 3 correlate_real; fft_size as the reference's argument (complex forms: that many values per
    // array; real forms: fft_size / 2 values per array, bin 0 = (DC, Nyquist))
#define REF_IRP(SFX, T, SPLIT)                                                                                                          \
    void ref_ir_product_##SFX(int op, T *ro, T *io, const T *r1, const T *i1, const T *r2, const T *i2, uintptr_t fft_size, double scale) \
    {                                                                                                                                   \
        SPLIT out(ro, io), in1(const_cast<T *>(r1), const_cast<T *>(i1)), in2(const_cast<T *>(r2), const_cast<T *>(i2));                \
        if (op == 0) ir_convolve_complex(&out, &in1, &in2, fft_size, (T) scale);                                                        \
        else if (op == 1) ir_convolve_real(&out, &in1, &in2, fft_size, (T) scale);                                                      \
        else if (op == 2) ir_correlate_complex(&out, &in1, &in2, fft_size, (T) scale);                                                  \
        else ir_correlate_real(&out, &in1, &in2, fft_size, (T) scale);                                                                  \
    }
    REF_IRP(f32, float, FFT_SPLIT_COMPLEX_F)
    REF_IRP(f64, double, FFT_SPLIT_COMPLEX_D)
#undef REF_IRP
}
