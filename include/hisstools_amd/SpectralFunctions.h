// Drop-in for the IR functions of SpectralFunctions.hpp (:365-436): ir_copy, ir_spike, ir_delay, ir_time_reverse, ir_phase
// and the four products ir_convolve_complex / _real, ir_correlate_complex / _real on packed half spectra (FFT_SPLIT_COMPLEX_F / _D of fft_size / 2 values per array), executed on the GPU through
// hcv_ir_exec (include/hisstools_amd.h).  Same names, argument order and meaning as the reference; `in` may equal `out`.
// Each call moves its operands over PCIe — HBM-resident pipelines call hcv_ir_exec_dev with device pointers instead.
#pragma once

#include "HISSTools_FFT.h"

namespace hisstools_amd_detail
{
    template <class Split> struct split_precision;
    template <> struct split_precision<FFT_SPLIT_COMPLEX_F> { enum { value = HCV_FFT_F32 }; };
    template <> struct split_precision<FFT_SPLIT_COMPLEX_D> { enum { value = HCV_FFT_F64 }; };

    inline unsigned ir_log2(uintptr_t fft_size)
    {
        unsigned l = 0;
        while ((uintptr_t(1) << l) < fft_size) l++;
        return l;
    }

    template <class Split> void ir_exec(int op, Split *out, const Split *in, uintptr_t fft_size, double value, bool zero_center)
    {
        hcv_ir_call c = { op, split_precision<Split>::value, ir_log2(fft_size), 1, in ? in->realp : nullptr, in ? in->imagp : nullptr,
                          out->realp, out->imagp, 0, 0, value, zero_center ? 1 : 0 };
        (void) hcv_ir_exec(&c);
    }
}

template <typename Split> void ir_copy(Split *out, const Split *in, uintptr_t fft_size)                     // :365-369
{
    hisstools_amd_detail::ir_exec(HCV_IR_COPY, out, in, fft_size, 0.0, false);
}

template <typename Split> void ir_spike(Split *out, uintptr_t fft_size, double spike_position)            // :371-375
{
    hisstools_amd_detail::ir_exec<Split>(HCV_IR_SPIKE, out, nullptr, fft_size, spike_position, false);
}

template <typename Split> void ir_delay(Split *out, const Split *in, uintptr_t fft_size, double delay)     // :377-384
{
    hisstools_amd_detail::ir_exec(HCV_IR_DELAY, out, in, fft_size, delay, false);
}

template <typename Split> void ir_time_reverse(Split *out, const Split *in, uintptr_t fft_size)            // :386-390
{
    hisstools_amd_detail::ir_exec(HCV_IR_TIME_REVERSE, out, in, fft_size, 0.0, false);
}

template <typename Setup, typename Split>
void ir_phase(Setup, Split *out, Split *in, uintptr_t fft_size, double phase, bool zero_center = false)   // :392-413
{
    hisstools_amd_detail::ir_exec(HCV_IR_PHASE, out, in, fft_size, phase, zero_center);
}

// ---- the products, :414-436: out = scale * in1 * in2 (convolve) or scale * in1 * conj(in2) (correlate).  The complex forms work on
// fft_size values per array, the real forms on packed half spectra of fft_size real samples (bin 0 = (DC, Nyquist): two real products).
// Power-of-two sizes (the reference's vector loops drop the remainder of any other, SpectralFunctions.hpp:44); out may be in1 or in2.
namespace hisstools_amd_detail
{
    template <class Split, class T> void ir_product(int op, Split *out, Split *in1, Split *in2, uintptr_t fft_size, T scale)
    {
        hcv_ir_product_call c = { op, split_precision<Split>::value, fft_size, 1, in1->realp, in1->imagp, in2->realp, in2->imagp,
                                  out->realp, out->imagp, 0, 0, 0, 0, (double) scale };
        (void) hcv_ir_product_exec(&c);
    }
}

template <typename Split, typename T> void ir_convolve_complex(Split *out, Split *in1, Split *in2, uintptr_t fft_size, T scale)   // :414-418
{
    hisstools_amd_detail::ir_product(HCV_IR_CONVOLVE_COMPLEX, out, in1, in2, fft_size, scale);
}

template <typename Split, typename T> void ir_convolve_real(Split *out, Split *in1, Split *in2, uintptr_t fft_size, T scale)      // :420-424
{
    hisstools_amd_detail::ir_product(HCV_IR_CONVOLVE_REAL, out, in1, in2, fft_size, scale);
}

template <typename Split, typename T> void ir_correlate_complex(Split *out, Split *in1, Split *in2, uintptr_t fft_size, T scale)  // :426-430
{
    hisstools_amd_detail::ir_product(HCV_IR_CORRELATE_COMPLEX, out, in1, in2, fft_size, scale);
}

template <typename Split, typename T> void ir_correlate_real(Split *out, Split *in1, Split *in2, uintptr_t fft_size, T scale)     // :432-436
{
    hisstools_amd_detail::ir_product(HCV_IR_CORRELATE_REAL, out, in1, in2, fft_size, scale);
}
