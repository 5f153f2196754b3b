// Minimal drop-in for the part of HISSTools_FFT/HISSTools_FFT.h the convolution path uses (float real transforms,
// HISSTools_FFT.h:87-369 / HISSTools_FFT.cpp:182-248): setup create/destroy, 5-arg rfft, 4-arg rifft.  The
// transforms execute on the GPU; a "setup" only remembers the maximum size (twiddle tables are cached per device).
#pragma once

#include "../hisstools_amd.h"

#include <cstdint>
#include <vector>

template <class T> struct Split
{
    Split() {}
    Split(T *real, T *imag) : realp(real), imagp(imag) {}
    T *realp;
    T *imagp;
};

typedef Split<float> FFT_SPLIT_COMPLEX_F;
struct FloatSetup { uintptr_t max_fft_log2; };
typedef FloatSetup *FFT_SETUP_F;

inline void hisstools_create_setup(FFT_SETUP_F *setup, uintptr_t max_fft_log_2) { *setup = new FloatSetup{ max_fft_log_2 }; }
inline void hisstools_destroy_setup(FFT_SETUP_F setup) { delete setup; }

inline void hisstools_rfft(FFT_SETUP_F, const float *input, FFT_SPLIT_COMPLEX_F *output, uintptr_t in_length, uintptr_t log2n)
{
    hcv_rfft_f32(input, in_length, in_length, 1, static_cast<unsigned>(log2n), output->realp, output->imagp);
}

// NB the reference destroys its input spectrum (in-place); this one leaves it intact.
inline void hisstools_rifft(FFT_SETUP_F, FFT_SPLIT_COMPLEX_F *input, float *output, uintptr_t log2n)
{
    hcv_rifft_f32(input->realp, input->imagp, 1, static_cast<unsigned>(log2n), output);
}
