/* TEST INFRASTRUCTURE — CPU oracle for the partitioned-convolution hot path.
 *
 * A plain-C restatement of the reference algorithm (HISSTools_Library, HIRT_Multichannel_Convolution
 * + HISSTools_FFT).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this; the product (hisstools_library_amd/) never does.
 *
 * Pinning status: PINNED against the unmodified reference compiled from /root/reference
 * (oracle/_ref/libhisstools_ref.so, recipe: oracle/Makefile) and against the golden vectors that
 * reference produced (tests/golden/, generator tests/golden/make_golden.py).  The reference
 * itself ships no tests or golden vectors for this path.
 */
#ifndef HCV_ORACLE_H
#define HCV_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ConvolveErrors.h:4-19 */
enum
{
    HCVO_ERR_NONE = 0,
    HCVO_ERR_IN_CHAN_OUT_OF_RANGE = 1,
    HCVO_ERR_OUT_CHAN_OUT_OF_RANGE = 2,
    HCVO_ERR_MEM_UNAVAILABLE = 3,
    HCVO_ERR_MEM_ALLOC_TOO_SMALL = 4,
    HCVO_ERR_TIME_IMPULSE_TOO_LONG = 5,
    HCVO_ERR_TIME_LENGTH_OUT_OF_RANGE = 6,
    HCVO_ERR_PARTITION_LENGTH_TOO_LARGE = 7,
    HCVO_ERR_FFT_SIZE_MAX_TOO_SMALL = 8,
    HCVO_ERR_FFT_SIZE_MAX_TOO_LARGE = 9,
    HCVO_ERR_FFT_SIZE_MAX_NON_POWER_OF_TWO = 10,
    HCVO_ERR_FFT_SIZE_OUT_OF_RANGE = 11,
    HCVO_ERR_FFT_SIZE_NON_POWER_OF_TWO = 12
};

/* FFT (vDSP packing, forward x2, inverse unnormalised) */
void hcvo_rfft_f32(const float *in, size_t in_len, unsigned log2n, float *realp, float *imagp);
void hcvo_rifft_f32(const float *realp, const float *imagp, unsigned log2n, float *out);
void hcvo_fft_f32(float *realp, float *imagp, unsigned log2n, int inverse);
void hcvo_rfft_f64(const double *in, size_t in_len, unsigned log2n, double *realp, double *imagp);
void hcvo_rifft_f64(const double *realp, const double *imagp, unsigned log2n, double *out);
void hcvo_fft_f64(double *realp, double *imagp, unsigned log2n, int inverse);
/* the rest of the hisstools_* surface (HISSTools_FFT.h:87-369): in-place real transforms, zip / unzip, float -> double */
void hcvo_rfft_inplace_f32(float *re, float *im, unsigned log2n);
void hcvo_rifft_inplace_f32(float *re, float *im, unsigned log2n);
void hcvo_rfft_inplace_f64(double *re, double *im, unsigned log2n);
void hcvo_rifft_inplace_f64(double *re, double *im, unsigned log2n);
void hcvo_unzip_f32(const float *in, float *re, float *im, unsigned log2n);
void hcvo_unzip_f64(const double *in, double *re, double *im, unsigned log2n);
void hcvo_unzip_zero_f32(const float *in, float *re, float *im, size_t in_len, unsigned log2n);
void hcvo_unzip_zero_f64(const double *in, double *re, double *im, size_t in_len, unsigned log2n);
void hcvo_unzip_zero_f32_f64(const float *in, double *re, double *im, size_t in_len, unsigned log2n);
void hcvo_zip_f32(const float *re, const float *im, float *out, unsigned log2n);
void hcvo_zip_f64(const double *re, const double *im, double *out, unsigned log2n);
void hcvo_rfft_f32_f64(const float *in, size_t in_len, unsigned log2n, double *realp, double *imagp);

/* Spectral IR functions (SpectralFunctions.hpp:284-410) on packed half spectra of fft_size real samples, and
 * spectral_processor::change_phase (SpectralProcessor.hpp:188-208).  _f64 variants take double. */
void hcvo_ir_copy_f32(float *ro, float *io, const float *ri, const float *ii, size_t fft_size);
void hcvo_ir_time_reverse_f32(float *ro, float *io, const float *ri, const float *ii, size_t fft_size);
void hcvo_ir_spike_f32(float *ro, float *io, size_t fft_size, double position);
void hcvo_ir_delay_f32(float *ro, float *io, const float *ri, const float *ii, size_t fft_size, double delay);
void hcvo_ir_phase_f32(float *ro, float *io, const float *ri, const float *ii, size_t fft_size, double phase, int zero_center);
size_t hcvo_change_phase_f32(const float *in, size_t size, double phase, double time_multiplier, float *out);
void hcvo_ir_product_f32(int op, float *ro, float *io, const float *r1, const float *i1, const float *r2, const float *i2, size_t fft_size, double scale);
void hcvo_ir_product_f64(int op, double *ro, double *io, const double *r1, const double *i1, const double *r2, const double *i2, size_t fft_size, double scale);
void hcvo_ir_copy_f64(double *ro, double *io, const double *ri, const double *ii, size_t fft_size);
void hcvo_ir_time_reverse_f64(double *ro, double *io, const double *ri, const double *ii, size_t fft_size);
void hcvo_ir_spike_f64(double *ro, double *io, size_t fft_size, double position);
void hcvo_ir_delay_f64(double *ro, double *io, const double *ri, const double *ii, size_t fft_size, double delay);
void hcvo_ir_phase_f64(double *ro, double *io, const double *ri, const double *ii, size_t fft_size, double phase, int zero_center);
size_t hcvo_change_phase_f64(const double *in, size_t size, double phase, double time_multiplier, double *out);

/* The class-level API is declared opaque; tests bind it through ctypes (oracle/oracle.py).
 * f32 entry points: hcvo_part_*, hcvo_td_*, hcvo_mono_*, hcvo_n2m_*, hcvo_conv_* (suffix _f32);
 * f64 ground-truth variants exist for part/td/mono (suffix _f64). */

#ifdef __cplusplus
}
#endif

#endif
