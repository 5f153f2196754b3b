// TEST INFRASTRUCTURE — not part of the product path.
//
// C-ABI driver around the *unmodified* reference sources under /root/reference
// (HIRT_Multichannel_Convolution/*.cpp + HISSTools_FFT/HISSTools_FFT.cpp).  The reference sources
// are compiled where they lie by oracle/Makefile; nothing from /root/reference is copied into
// this repository.  The resulting shared object (oracle/_ref/libhisstools_ref.so) is used
//   * to pin the C restatement in oracle/hcv_oracle.c,
//   * to generate the golden vectors in tests/golden/ (tests/golden/make_golden.py),
//   * as the "reference" CPU baseline timed by bench.py.
//
// Every entry point is a thin handle-based wrapper: one function per public method of the
// reference classes (Convolver.h:23-50, NToMonoConvolve.h:18-24, MonoConvolve.h:30-48,
// PartitionedConvolve.h:23-41, TimeDomainConvolve.h:15-31, HISSTools_FFT.h:87-369).

#include "HIRT_Multichannel_Convolution/Convolver.h"
#include "HIRT_Multichannel_Convolution/MonoConvolve.h"
#include "HIRT_Multichannel_Convolution/NToMonoConvolve.h"
#include "HIRT_Multichannel_Convolution/PartitionedConvolve.h"
#include "HIRT_Multichannel_Convolution/TimeDomainConvolve.h"
#include "HISSTools_FFT/HISSTools_FFT.h"

#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

using namespace HISSTools;

namespace
{
    // The reference FFT only takes its SIMD path for pointers aligned to the SIMD width; give it
    // that (64 B covers SSE/AVX/AVX-512 builds).
    float *aalloc(size_t n)
    {
        void *p = nullptr;
        if (posix_memalign(&p, 64, (n ? n : 1) * sizeof(float))) return nullptr;
        return static_cast<float *>(p);
    }
}

extern "C"
{
    // ---------------------------------------------------------------- FFT (HISSTools_FFT.cpp:182-248)

    // out-of-place real FFT with zero padding: in[in_len] -> realp/imagp[2^(log2n-1)]
    void ref_rfft_f32(const float *in, uintptr_t in_len, uintptr_t log2n, float *realp, float *imagp)
    {
        FFT_SETUP_F setup;
        hisstools_create_setup(&setup, log2n);
        uintptr_t half = uintptr_t(1) << (log2n - 1);
        float *buf = aalloc(2 * half);
        FFT_SPLIT_COMPLEX_F s(buf, buf + half);
        hisstools_rfft(setup, in, &s, in_len, log2n);
        std::memcpy(realp, s.realp, half * sizeof(float));
        std::memcpy(imagp, s.imagp, half * sizeof(float));
        free(buf);
        hisstools_destroy_setup(setup);
    }

    // out-of-place real inverse FFT: realp/imagp[2^(log2n-1)] -> out[2^log2n] (unnormalised)
    void ref_rifft_f32(const float *realp, const float *imagp, uintptr_t log2n, float *out)
    {
        FFT_SETUP_F setup;
        hisstools_create_setup(&setup, log2n);
        uintptr_t half = uintptr_t(1) << (log2n - 1);
        float *buf = aalloc(2 * half);
        float *o = aalloc(2 * half);
        std::memcpy(buf, realp, half * sizeof(float));
        std::memcpy(buf + half, imagp, half * sizeof(float));
        FFT_SPLIT_COMPLEX_F s(buf, buf + half);
        hisstools_rifft(setup, &s, o, log2n);
        std::memcpy(out, o, 2 * half * sizeof(float));
        free(buf);
        free(o);
        hisstools_destroy_setup(setup);
    }

    // in-place complex FFT / iFFT on split data of length 2^log2n
    void ref_fft_f32(float *realp, float *imagp, uintptr_t log2n, int inverse)
    {
        FFT_SETUP_F setup;
        hisstools_create_setup(&setup, log2n);
        uintptr_t len = uintptr_t(1) << log2n;
        float *buf = aalloc(2 * len);
        std::memcpy(buf, realp, len * sizeof(float));
        std::memcpy(buf + len, imagp, len * sizeof(float));
        FFT_SPLIT_COMPLEX_F s(buf, buf + len);
        if (inverse) hisstools_ifft(setup, &s, log2n); else hisstools_fft(setup, &s, log2n);
        std::memcpy(realp, buf, len * sizeof(float));
        std::memcpy(imagp, buf + len, len * sizeof(float));
        free(buf);
        hisstools_destroy_setup(setup);
    }

    void ref_fft_f64(double *realp, double *imagp, uintptr_t log2n, int inverse)
    {
        FFT_SETUP_D setup;
        hisstools_create_setup(&setup, log2n);
        FFT_SPLIT_COMPLEX_D s(realp, imagp);
        if (inverse) hisstools_ifft(setup, &s, log2n); else hisstools_fft(setup, &s, log2n);
        hisstools_destroy_setup(setup);
    }

    void ref_rfft_f64(const double *in, uintptr_t in_len, uintptr_t log2n, double *realp, double *imagp)
    {
        FFT_SETUP_D setup;
        hisstools_create_setup(&setup, log2n);
        FFT_SPLIT_COMPLEX_D s(realp, imagp);
        hisstools_rfft(setup, in, &s, in_len, log2n);
        hisstools_destroy_setup(setup);
    }

    void ref_rifft_f64(double *realp, double *imagp, uintptr_t log2n, double *out)
    {
        FFT_SETUP_D setup;
        hisstools_create_setup(&setup, log2n);
        FFT_SPLIT_COMPLEX_D s(realp, imagp);
        hisstools_rifft(setup, &s, out, log2n);
        hisstools_destroy_setup(setup);
    }

    void ref_unzip_f32(const float *in, float *realp, float *imagp, uintptr_t log2n)
    {
        FFT_SPLIT_COMPLEX_F s(realp, imagp);
        hisstools_unzip(in, &s, log2n);
    }

    void ref_zip_f32(const float *realp, const float *imagp, float *out, uintptr_t log2n)
    {
        FFT_SPLIT_COMPLEX_F s(const_cast<float *>(realp), const_cast<float *>(imagp));
        hisstools_zip(&s, out, log2n);
    }

    void ref_unzip_zero_f32(const float *in, float *realp, float *imagp, uintptr_t in_len, uintptr_t log2n)
    {
        FFT_SPLIT_COMPLEX_F s(realp, imagp);
        hisstools_unzip_zero(in, &s, in_len, log2n);
    }

    // in-place real transforms on split data (HISSTools_FFT.h:154,166,244,256); buffers are copied to aligned memory so
    // the reference takes its SIMD path exactly as a host using ALIGNED_MALLOC would
    void ref_rfft_inplace_f32(float *re, float *im, uintptr_t log2n, int inverse)
    {
        FFT_SETUP_F setup;
        hisstools_create_setup(&setup, log2n);
        uintptr_t half = (uintptr_t(1) << log2n) >> 1;
        float *buf = aalloc(2 * half + 8);
        std::memcpy(buf, re, half * sizeof(float));
        std::memcpy(buf + half, im, half * sizeof(float));
        FFT_SPLIT_COMPLEX_F s(buf, buf + half);
        if (inverse) hisstools_rifft(setup, &s, log2n); else hisstools_rfft(setup, &s, log2n);
        std::memcpy(re, buf, half * sizeof(float));
        std::memcpy(im, buf + half, half * sizeof(float));
        free(buf);
        hisstools_destroy_setup(setup);
    }

    void ref_rfft_inplace_f64(double *re, double *im, uintptr_t log2n, int inverse)
    {
        FFT_SETUP_D setup;
        hisstools_create_setup(&setup, log2n);
        FFT_SPLIT_COMPLEX_D s(re, im);
        if (inverse) hisstools_rifft(setup, &s, log2n); else hisstools_rfft(setup, &s, log2n);
        hisstools_destroy_setup(setup);
    }

    void ref_rfft_f32_f64(const float *in, uintptr_t in_len, uintptr_t log2n, double *realp, double *imagp)
    {
        FFT_SETUP_D setup;
        hisstools_create_setup(&setup, log2n);
        FFT_SPLIT_COMPLEX_D s(realp, imagp);
        hisstools_rfft(setup, in, &s, in_len, log2n);
        hisstools_destroy_setup(setup);
    }

    void ref_unzip_f64(const double *in, double *realp, double *imagp, uintptr_t log2n)
    {
        FFT_SPLIT_COMPLEX_D s(realp, imagp);
        hisstools_unzip(in, &s, log2n);
    }

    void ref_zip_f64(const double *realp, const double *imagp, double *out, uintptr_t log2n)
    {
        FFT_SPLIT_COMPLEX_D s(const_cast<double *>(realp), const_cast<double *>(imagp));
        hisstools_zip(&s, out, log2n);
    }

    void ref_unzip_zero_f64(const double *in, double *realp, double *imagp, uintptr_t in_len, uintptr_t log2n)
    {
        FFT_SPLIT_COMPLEX_D s(realp, imagp);
        hisstools_unzip_zero(in, &s, in_len, log2n);
    }

    void ref_unzip_zero_f32_f64(const float *in, double *realp, double *imagp, uintptr_t in_len, uintptr_t log2n)
    {
        FFT_SPLIT_COMPLEX_D s(realp, imagp);
        hisstools_unzip_zero(in, &s, in_len, log2n);
    }

    // ---------------------------------------------------------------- PartitionedConvolve

    void *ref_part_new(uintptr_t maxFFTSize, uintptr_t maxLength, uintptr_t offset, uintptr_t length)
    {
        return new PartitionedConvolve(maxFFTSize, maxLength, offset, length);
    }
    void ref_part_delete(void *h) { delete static_cast<PartitionedConvolve *>(h); }
    int ref_part_set_fft_size(void *h, uintptr_t n) { return static_cast<PartitionedConvolve *>(h)->setFFTSize(n); }
    int ref_part_set_length(void *h, uintptr_t n) { return static_cast<PartitionedConvolve *>(h)->setLength(n); }
    void ref_part_set_offset(void *h, uintptr_t n) { static_cast<PartitionedConvolve *>(h)->setOffset(n); }
    void ref_part_set_reset_offset(void *h, intptr_t n) { static_cast<PartitionedConvolve *>(h)->setResetOffset(n); }
    int ref_part_set(void *h, const float *ir, uintptr_t len) { return static_cast<PartitionedConvolve *>(h)->set(ir, len); }
    void ref_part_reset(void *h) { static_cast<PartitionedConvolve *>(h)->reset(); }
    int ref_part_process(void *h, const float *in, float *out, uintptr_t n)
    {
        return static_cast<PartitionedConvolve *>(h)->process(in, out, n) ? 1 : 0;
    }

    // ---------------------------------------------------------------- TimeDomainConvolve

    void *ref_td_new(uintptr_t offset, uintptr_t length) { return new TimeDomainConvolve(offset, length); }
    void ref_td_delete(void *h) { delete static_cast<TimeDomainConvolve *>(h); }
    int ref_td_set_length(void *h, uintptr_t n) { return static_cast<TimeDomainConvolve *>(h)->setLength(n); }
    void ref_td_set_offset(void *h, uintptr_t n) { static_cast<TimeDomainConvolve *>(h)->setOffset(n); }
    int ref_td_set(void *h, const float *ir, uintptr_t len) { return static_cast<TimeDomainConvolve *>(h)->set(ir, len); }
    void ref_td_reset(void *h) { static_cast<TimeDomainConvolve *>(h)->reset(); }
    int ref_td_process(void *h, const float *in, float *out, uintptr_t n)
    {
        return static_cast<TimeDomainConvolve *>(h)->process(in, out, n) ? 1 : 0;
    }

    // ---------------------------------------------------------------- MonoConvolve

    // returns nullptr if the reference constructor throws (MonoConvolve.cpp:207-229)
    void *ref_mono_new(uintptr_t maxLength, int latency)
    {
        try { return new MonoConvolve(maxLength, static_cast<LatencyMode>(latency)); }
        catch (std::runtime_error &) { return nullptr; }
    }
    void *ref_mono_new_custom(uintptr_t maxLength, int zeroLatency, uint32_t A, uint32_t B, uint32_t C, uint32_t D, char *err, size_t errlen)
    {
        try { return new MonoConvolve(maxLength, zeroLatency != 0, A, B, C, D); }
        catch (std::runtime_error &e)
        {
            if (err && errlen) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
            return nullptr;
        }
    }
    void ref_mono_delete(void *h) { delete static_cast<MonoConvolve *>(h); }
    void ref_mono_set_reset_offset(void *h, intptr_t n) { static_cast<MonoConvolve *>(h)->setResetOffset(n); }
    int ref_mono_resize(void *h, uintptr_t n) { return static_cast<MonoConvolve *>(h)->resize(n); }
    int ref_mono_set(void *h, const float *ir, uintptr_t len, int resize) { return static_cast<MonoConvolve *>(h)->set(ir, len, resize != 0); }
    int ref_mono_reset(void *h) { return static_cast<MonoConvolve *>(h)->reset(); }
    void ref_mono_process(void *h, const float *in, float *temp, float *out, uintptr_t n, int accumulate)
    {
        static_cast<MonoConvolve *>(h)->process(in, temp, out, n, accumulate != 0);
    }

    // ---------------------------------------------------------------- NToMonoConvolve

    void *ref_n2m_new(uint32_t inChans, uintptr_t maxLength, int latency)
    {
        return new NToMonoConvolve(inChans, maxLength, static_cast<LatencyMode>(latency));
    }
    void ref_n2m_delete(void *h) { delete static_cast<NToMonoConvolve *>(h); }
    int ref_n2m_resize(void *h, uint32_t in, uintptr_t len) { return static_cast<NToMonoConvolve *>(h)->resize(in, len); }
    int ref_n2m_set(void *h, uint32_t in, const float *ir, uintptr_t len, int resize)
    {
        return static_cast<NToMonoConvolve *>(h)->set(in, ir, len, resize != 0);
    }
    int ref_n2m_reset(void *h, uint32_t in) { return static_cast<NToMonoConvolve *>(h)->reset(in); }
    void ref_n2m_process(void *h, const float *const *ins, float *out, float *temp, size_t n, size_t activeIns)
    {
        static_cast<NToMonoConvolve *>(h)->process(ins, out, temp, n, activeIns);
    }

    // ---------------------------------------------------------------- Convolver

    void *ref_conv_new(uint32_t numIns, uint32_t numOuts, int latency)
    {
        return new Convolver(numIns, numOuts, static_cast<LatencyMode>(latency));
    }
    void *ref_conv_new_parallel(uint32_t numIO, int latency)
    {
        return new Convolver(numIO, static_cast<LatencyMode>(latency));
    }
    void ref_conv_delete(void *h) { delete static_cast<Convolver *>(h); }
    void ref_conv_clear(void *h, int resize) { static_cast<Convolver *>(h)->clear(resize != 0); }
    void ref_conv_clear_chan(void *h, uint32_t in, uint32_t out, int resize) { static_cast<Convolver *>(h)->clear(in, out, resize != 0); }
    void ref_conv_reset(void *h) { static_cast<Convolver *>(h)->reset(); }
    int ref_conv_reset_chan(void *h, uint32_t in, uint32_t out) { return static_cast<Convolver *>(h)->reset(in, out); }
    int ref_conv_resize(void *h, uint32_t in, uint32_t out, uintptr_t len) { return static_cast<Convolver *>(h)->resize(in, out, len); }
    int ref_conv_set_f32(void *h, uint32_t in, uint32_t out, const float *ir, uintptr_t len, int resize)
    {
        return static_cast<Convolver *>(h)->set(in, out, ir, len, resize != 0);
    }
    int ref_conv_set_f64(void *h, uint32_t in, uint32_t out, const double *ir, uintptr_t len, int resize)
    {
        return static_cast<Convolver *>(h)->set(in, out, ir, len, resize != 0);
    }
    void ref_conv_process_f32(void *h, const float *const *ins, float **outs, size_t numIns, size_t numOuts, size_t n)
    {
        static_cast<Convolver *>(h)->process(ins, outs, numIns, numOuts, n);
    }
    void ref_conv_process_f64(void *h, const double *const *ins, double **outs, size_t numIns, size_t numOuts, size_t n)
    {
        static_cast<Convolver *>(h)->process(ins, outs, numIns, numOuts, n);
    }

    // ---------------------------------------------------------------- timing helpers (CPU baseline)
    //
    // Stream `total` samples of channel-contiguous audio (ins: [numIns][total], outs: [numOuts][total])
    // through a Convolver in `block`-sample process() calls; returns wall seconds (steady_clock).
    // block must be <= 2048 (reference TimeDomainConvolve wrap defect, TimeDomainConvolve.cpp:138-153).
    double ref_conv_stream_f32(void *h, const float *ins, float *outs, size_t numIns, size_t numOuts, size_t total, size_t block)
    {
        Convolver *c = static_cast<Convolver *>(h);
        std::vector<const float *> ip(numIns);
        std::vector<float *> op(numOuts);
        auto t0 = std::chrono::steady_clock::now();
        for (size_t pos = 0; pos < total; pos += block)
        {
            size_t n = (total - pos) < block ? (total - pos) : block;
            for (size_t i = 0; i < numIns; i++) ip[i] = ins + i * total + pos;
            for (size_t o = 0; o < numOuts; o++) op[o] = outs + o * total + pos;
            c->process(ip.data(), op.data(), numIns, numOuts, n);
        }
        auto t1 = std::chrono::steady_clock::now();
        return std::chrono::duration<double>(t1 - t0).count();
    }

    double ref_part_stream_f32(void *h, const float *in, float *out, size_t total, size_t block)
    {
        PartitionedConvolve *c = static_cast<PartitionedConvolve *>(h);
        auto t0 = std::chrono::steady_clock::now();
        for (size_t pos = 0; pos < total; pos += block)
        {
            size_t n = (total - pos) < block ? (total - pos) : block;
            c->process(in + pos, out + pos, n);
        }
        auto t1 = std::chrono::steady_clock::now();
        return std::chrono::duration<double>(t1 - t0).count();
    }

    double ref_mono_stream_f32(void *h, const float *in, float *out, size_t total, size_t block)
    {
        MonoConvolve *c = static_cast<MonoConvolve *>(h);
        std::vector<float> temp(block);
        auto t0 = std::chrono::steady_clock::now();
        for (size_t pos = 0; pos < total; pos += block)
        {
            size_t n = (total - pos) < block ? (total - pos) : block;
            c->process(in + pos, temp.data(), out + pos, n, false);
        }
        auto t1 = std::chrono::steady_clock::now();
        return std::chrono::duration<double>(t1 - t0).count();
    }
}
