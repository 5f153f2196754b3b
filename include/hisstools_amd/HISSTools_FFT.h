// Drop-in for HISSTools_FFT/HISSTools_FFT.h (HISSTools_FFT.h:87-369): the same setup / split types and the same
// overload set — complex and real transforms in float and double, in place or out of place, zip / unzip — with every
// transform executed on the GPU through hcv_fft_exec (include/hisstools_amd.h).  A "setup" only remembers the maximum
// size: twiddle tables live on the device and are cached per size.
//
// Differences from the reference: the out-of-place hisstools_rifft leaves its input spectrum untouched; each call moves
// its operands over PCIe, so HBM-resident pipelines should call hcv_fft_exec_dev with device pointers instead.
#pragma once

#include "../hisstools_amd.h"

#include <cstdint>

template <class T> struct Split
{
    Split() {}
    Split(T *real, T *imag) : realp(real), imagp(imag) {}
    T *realp;
    T *imagp;
};

template <class T> struct Setup { uintptr_t max_fft_log2; };

typedef Split<double> FFT_SPLIT_COMPLEX_D;
typedef Split<float> FFT_SPLIT_COMPLEX_F;
typedef Setup<double> *FFT_SETUP_D;
typedef Setup<float> *FFT_SETUP_F;

namespace hisstools_amd_detail
{
    template <class T> struct precision;
    template <> struct precision<float> { enum { value = HCV_FFT_F32 }; };
    template <> struct precision<double> { enum { value = HCV_FFT_F64 }; };

    inline void exec(int op, int prec, uintptr_t log2n, const void *sa, const void *sb, void *da, void *db, uintptr_t in_length)
    {
        hcv_fft_call c = { op, prec, static_cast<unsigned>(log2n), 1, sa, sb, da, db, 0, 0, static_cast<size_t>(in_length) };
        (void) hcv_fft_exec(&c);
    }

    template <class T> void in_place(int op, Split<T> *io, uintptr_t log2n)
    {
        exec(op, precision<T>::value, log2n, io->realp, io->imagp, io->realp, io->imagp, 0);
    }
}

// setups (HISSTools_FFT.h:87-118)
inline void hisstools_create_setup(FFT_SETUP_D *setup, uintptr_t max_fft_log_2) { *setup = new Setup<double>{ max_fft_log_2 }; }
inline void hisstools_create_setup(FFT_SETUP_F *setup, uintptr_t max_fft_log_2) { *setup = new Setup<float>{ max_fft_log_2 }; }
inline void hisstools_destroy_setup(FFT_SETUP_D setup) { delete setup; }
inline void hisstools_destroy_setup(FFT_SETUP_F setup) { delete setup; }

// in-place complex transforms (:130,142,220,232)
inline void hisstools_fft(FFT_SETUP_D, FFT_SPLIT_COMPLEX_D *input, uintptr_t log2n) { hisstools_amd_detail::in_place(HCV_FFT_FFT, input, log2n); }
inline void hisstools_fft(FFT_SETUP_F, FFT_SPLIT_COMPLEX_F *input, uintptr_t log2n) { hisstools_amd_detail::in_place(HCV_FFT_FFT, input, log2n); }
inline void hisstools_ifft(FFT_SETUP_D, FFT_SPLIT_COMPLEX_D *input, uintptr_t log2n) { hisstools_amd_detail::in_place(HCV_FFT_IFFT, input, log2n); }
inline void hisstools_ifft(FFT_SETUP_F, FFT_SPLIT_COMPLEX_F *input, uintptr_t log2n) { hisstools_amd_detail::in_place(HCV_FFT_IFFT, input, log2n); }

// in-place real transforms on unzipped data (:154,166,244,256)
inline void hisstools_rfft(FFT_SETUP_D, FFT_SPLIT_COMPLEX_D *input, uintptr_t log2n) { hisstools_amd_detail::in_place(HCV_FFT_RFFT, input, log2n); }
inline void hisstools_rfft(FFT_SETUP_F, FFT_SPLIT_COMPLEX_F *input, uintptr_t log2n) { hisstools_amd_detail::in_place(HCV_FFT_RFFT, input, log2n); }
inline void hisstools_rifft(FFT_SETUP_D, FFT_SPLIT_COMPLEX_D *input, uintptr_t log2n) { hisstools_amd_detail::in_place(HCV_FFT_RIFFT, input, log2n); }
inline void hisstools_rifft(FFT_SETUP_F, FFT_SPLIT_COMPLEX_F *input, uintptr_t log2n) { hisstools_amd_detail::in_place(HCV_FFT_RIFFT, input, log2n); }

// out-of-place real transforms (:180,194,208,269,282)
inline void hisstools_rfft(FFT_SETUP_D, const double *input, FFT_SPLIT_COMPLEX_D *output, uintptr_t in_length, uintptr_t log2n)
{
    hisstools_amd_detail::exec(HCV_FFT_RFFT_ZIP, HCV_FFT_F64, log2n, input, nullptr, output->realp, output->imagp, in_length);
}
inline void hisstools_rfft(FFT_SETUP_F, const float *input, FFT_SPLIT_COMPLEX_F *output, uintptr_t in_length, uintptr_t log2n)
{
    hisstools_amd_detail::exec(HCV_FFT_RFFT_ZIP, HCV_FFT_F32, log2n, input, nullptr, output->realp, output->imagp, in_length);
}
inline void hisstools_rfft(FFT_SETUP_D, const float *input, FFT_SPLIT_COMPLEX_D *output, uintptr_t in_length, uintptr_t log2n)
{
    hisstools_amd_detail::exec(HCV_FFT_RFFT_ZIP, HCV_FFT_F32_TO_F64, log2n, input, nullptr, output->realp, output->imagp, in_length);
}
inline void hisstools_rifft(FFT_SETUP_D, FFT_SPLIT_COMPLEX_D *input, double *output, uintptr_t log2n)
{
    hisstools_amd_detail::exec(HCV_FFT_RIFFT_ZIP, HCV_FFT_F64, log2n, input->realp, input->imagp, output, nullptr, 0);
}
inline void hisstools_rifft(FFT_SETUP_F, FFT_SPLIT_COMPLEX_F *input, float *output, uintptr_t log2n)
{
    hisstools_amd_detail::exec(HCV_FFT_RIFFT_ZIP, HCV_FFT_F32, log2n, input->realp, input->imagp, output, nullptr, 0);
}

// zip / unzip (:295-369)
inline void hisstools_unzip_zero(const double *input, FFT_SPLIT_COMPLEX_D *output, uintptr_t in_length, uintptr_t log2n)
{
    hisstools_amd_detail::exec(HCV_FFT_UNZIP, HCV_FFT_F64, log2n, input, nullptr, output->realp, output->imagp, in_length);
}
inline void hisstools_unzip_zero(const float *input, FFT_SPLIT_COMPLEX_F *output, uintptr_t in_length, uintptr_t log2n)
{
    hisstools_amd_detail::exec(HCV_FFT_UNZIP, HCV_FFT_F32, log2n, input, nullptr, output->realp, output->imagp, in_length);
}
inline void hisstools_unzip_zero(const float *input, FFT_SPLIT_COMPLEX_D *output, uintptr_t in_length, uintptr_t log2n)
{
    hisstools_amd_detail::exec(HCV_FFT_UNZIP, HCV_FFT_F32_TO_F64, log2n, input, nullptr, output->realp, output->imagp, in_length);
}
inline void hisstools_unzip(const double *input, FFT_SPLIT_COMPLEX_D *output, uintptr_t log2n)
{
    hisstools_amd_detail::exec(HCV_FFT_UNZIP, HCV_FFT_F64, log2n, input, nullptr, output->realp, output->imagp, uintptr_t(1) << log2n);
}
inline void hisstools_unzip(const float *input, FFT_SPLIT_COMPLEX_F *output, uintptr_t log2n)
{
    hisstools_amd_detail::exec(HCV_FFT_UNZIP, HCV_FFT_F32, log2n, input, nullptr, output->realp, output->imagp, uintptr_t(1) << log2n);
}
inline void hisstools_zip(const FFT_SPLIT_COMPLEX_D *input, double *output, uintptr_t log2n)
{
    hisstools_amd_detail::exec(HCV_FFT_ZIP, HCV_FFT_F64, log2n, input->realp, input->imagp, output, nullptr, 0);
}
inline void hisstools_zip(const FFT_SPLIT_COMPLEX_F *input, float *output, uintptr_t log2n)
{
    hisstools_amd_detail::exec(HCV_FFT_ZIP, HCV_FFT_F32, log2n, input->realp, input->imagp, output, nullptr, 0);
}
