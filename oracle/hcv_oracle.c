/* TEST INFRASTRUCTURE — CPU oracle ("port") for the partitioned-convolution hot path.
 *
 * Plain-C restatement of the reference's algorithm (see hcv_oracle.h for scope and pinning
 * status: PINNED against oracle/_ref and tests/golden/).  Never linked, loaded or called by the
 * product path.
 */
#define _POSIX_C_SOURCE 200809L
#define _DEFAULT_SOURCE          /* M_PI */
#include "hcv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define REAL float
#define FN(x) x##_f32
#define RM(x) x##f                 /* libm function of REAL's precision */
#include "hcv_oracle_body.inc"
#undef REAL
#undef FN
#undef RM

#define REAL double
#define FN(x) x##_f64
#define RM(x) x
#include "hcv_oracle_body.inc"
#undef REAL
#undef FN
#undef RM

/* ------------------------------------------------------------------------------------------------
 * float samples -> double split (HISSTools_FFT.h:208,321; the converting unzip_complex, Core.h:1199-1210)
 * ---------------------------------------------------------------------------------------------- */

void hcvo_unzip_zero_f32_f64(const float *in, double *re, double *im, size_t in_len, unsigned log2n)
{
    size_t n = (size_t) 1 << log2n, half = n >> 1;
    if (in_len > n) in_len = n;
    for (size_t k = 0; k < half; k++)
    {
        re[k] = (2 * k < in_len) ? (double) in[2 * k] : 0.0;
        im[k] = (2 * k + 1 < in_len) ? (double) in[2 * k + 1] : 0.0;
    }
}

void hcvo_rfft_f32_f64(const float *in, size_t in_len, unsigned log2n, double *realp, double *imagp)
{
    hcvo_unzip_zero_f32_f64(in, realp, imagp, in_len, log2n);
    hcvo_rfft_inplace_f64(realp, imagp, log2n);
}

/* ------------------------------------------------------------------------------------------------
 * NToMonoConvolve  (NToMonoConvolve.cpp)
 * ---------------------------------------------------------------------------------------------- */

typedef struct hcvo_n2m_f32
{
    uint32_t num_ins;
    hcvo_mono_f32 **mono;
} hcvo_n2m_f32;

hcvo_n2m_f32 *hcvo_n2m_new_f32(uint32_t in_chans, size_t max_length, int latency)       /* :4-9 */
{
    hcvo_n2m_f32 *c = (hcvo_n2m_f32 *) calloc(1, sizeof(hcvo_n2m_f32));
    c->num_ins = in_chans;
    c->mono = (hcvo_mono_f32 **) calloc(in_chans ? in_chans : 1, sizeof(hcvo_mono_f32 *));
    for (uint32_t i = 0; i < in_chans; i++) c->mono[i] = hcvo_mono_new_f32(max_length, latency);
    return c;
}

void hcvo_n2m_delete_f32(hcvo_n2m_f32 *c)
{
    if (!c) return;
    for (uint32_t i = 0; i < c->num_ins; i++) hcvo_mono_delete_f32(c->mono[i]);
    free(c->mono);
    free(c);
}

/* doChannel (:11-18): range check then forward */
int hcvo_n2m_resize_f32(hcvo_n2m_f32 *c, uint32_t in, size_t len)
{
    return in < c->num_ins ? hcvo_mono_resize_f32(c->mono[in], len) : HCVO_ERR_IN_CHAN_OUT_OF_RANGE;
}
int hcvo_n2m_set_f32(hcvo_n2m_f32 *c, uint32_t in, const float *ir, size_t len, int resize)
{
    return in < c->num_ins ? hcvo_mono_set_f32(c->mono[in], ir, len, resize) : HCVO_ERR_IN_CHAN_OUT_OF_RANGE;
}
int hcvo_n2m_reset_f32(hcvo_n2m_f32 *c, uint32_t in)
{
    return in < c->num_ins ? hcvo_mono_reset_f32(c->mono[in]) : HCVO_ERR_IN_CHAN_OUT_OF_RANGE;
}
void hcvo_n2m_set_reset_offset_f32(hcvo_n2m_f32 *c, intptr_t offset)     /* oracle-only: pin phases */
{
    for (uint32_t i = 0; i < c->num_ins; i++) hcvo_mono_set_reset_offset_f32(c->mono[i], offset);
}

void hcvo_n2m_process_f32(hcvo_n2m_f32 *c, const float *const *ins, float *out, float *temp, size_t n, size_t active_ins)  /* :35-43 */
{
    memset(out, 0, sizeof(float) * n);
    for (uint32_t i = 0; i < c->num_ins && i < active_ins; i++)
        hcvo_mono_process_f32(c->mono[i], ins[i], temp, out, n, 1);
}

/* ------------------------------------------------------------------------------------------------
 * Convolver  (Convolver.cpp)
 * ---------------------------------------------------------------------------------------------- */

typedef struct hcvo_conv_f32
{
    uint32_t num_ins, num_outs;
    int n2m;
    hcvo_n2m_f32 **conv;
    float *temp;                 /* (num_ins + 2) * frame floats, grown on demand (:140,185-195) */
    size_t temp_frame;
} hcvo_conv_f32;

hcvo_conv_f32 *hcvo_conv_new_f32(uint32_t num_ins, uint32_t num_outs, int latency)      /* :5-22 */
{
    hcvo_conv_f32 *c = (hcvo_conv_f32 *) calloc(1, sizeof(hcvo_conv_f32));
    num_ins = num_ins < 1 ? 1 : num_ins;
    c->n2m = 1; c->num_ins = num_ins; c->num_outs = num_outs;
    c->conv = (hcvo_n2m_f32 **) calloc(num_outs ? num_outs : 1, sizeof(hcvo_n2m_f32 *));
    for (uint32_t o = 0; o < num_outs; o++) c->conv[o] = hcvo_n2m_new_f32(num_ins, 16384, latency);
    return c;
}

hcvo_conv_f32 *hcvo_conv_new_parallel_f32(uint32_t num_io, int latency)                 /* :24-41 */
{
    hcvo_conv_f32 *c = (hcvo_conv_f32 *) calloc(1, sizeof(hcvo_conv_f32));
    num_io = num_io < 1 ? 1 : num_io;
    c->n2m = 0; c->num_ins = num_io; c->num_outs = num_io;
    c->conv = (hcvo_n2m_f32 **) calloc(num_io, sizeof(hcvo_n2m_f32 *));
    for (uint32_t o = 0; o < num_io; o++) c->conv[o] = hcvo_n2m_new_f32(1, 16384, latency);
    return c;
}

void hcvo_conv_delete_f32(hcvo_conv_f32 *c)
{
    if (!c) return;
    for (uint32_t o = 0; o < c->num_outs; o++) hcvo_n2m_delete_f32(c->conv[o]);
    free(c->conv);
    free(c->temp);
    free(c);
}

void hcvo_conv_set_reset_offset_f32(hcvo_conv_f32 *c, intptr_t offset)   /* oracle-only */
{
    for (uint32_t o = 0; o < c->num_outs; o++) hcvo_n2m_set_reset_offset_f32(c->conv[o], offset);
}

int hcvo_conv_set_f32(hcvo_conv_f32 *c, uint32_t in, uint32_t out, const float *ir, size_t len, int resize)  /* :114-124 */
{
    if (!c->n2m) in -= out;                      /* unsigned wrap: only in == out survives (:118) */
    return out < c->num_outs ? hcvo_n2m_set_f32(c->conv[out], in, ir, len, resize) : HCVO_ERR_OUT_CHAN_OUT_OF_RANGE;
}

int hcvo_conv_set_f64(hcvo_conv_f32 *c, uint32_t in, uint32_t out, const double *ir, size_t len, int resize)  /* :126-134 */
{
    float *f = (float *) malloc(sizeof(float) * (len ? len : 1));
    for (size_t i = 0; i < len; i++) f[i] = (float) ir[i];
    int err = hcvo_conv_set_f32(c, in, out, f, len, resize);     /* NB a NULL double IR becomes a non-NULL empty float IR */
    free(f);
    return err;
}

int hcvo_conv_resize_f32(hcvo_conv_f32 *c, uint32_t in, uint32_t out, size_t len)      /* :100-110 */
{
    if (!c->n2m) in -= out;
    return out < c->num_outs ? hcvo_n2m_resize_f32(c->conv[out], in, len) : HCVO_ERR_IN_CHAN_OUT_OF_RANGE;
}

int hcvo_conv_reset_chan_f32(hcvo_conv_f32 *c, uint32_t in, uint32_t out)              /* :88-98 */
{
    if (!c->n2m) in -= out;
    return out < c->num_outs ? hcvo_n2m_reset_f32(c->conv[out], in) : HCVO_ERR_OUT_CHAN_OUT_OF_RANGE;
}

void hcvo_conv_reset_f32(hcvo_conv_f32 *c)                                              /* :73-86 */
{
    for (uint32_t o = 0; o < c->num_outs; o++)
    {
        if (c->n2m) for (uint32_t i = 0; i < c->num_ins; i++) hcvo_conv_reset_chan_f32(c, i, o);
        else hcvo_conv_reset_chan_f32(c, o, o);
    }
}

void hcvo_conv_clear_chan_f32(hcvo_conv_f32 *c, uint32_t in, uint32_t out, int resize) /* :66-69 */
{
    hcvo_conv_set_f32(c, in, out, NULL, 0, resize);
}

void hcvo_conv_clear_f32(hcvo_conv_f32 *c, int resize)                                  /* :51-64 */
{
    for (uint32_t o = 0; o < c->num_outs; o++)
    {
        if (c->n2m) for (uint32_t i = 0; i < c->num_ins; i++) hcvo_conv_clear_chan_f32(c, i, o, resize);
        else hcvo_conv_clear_chan_f32(c, o, o, resize);
    }
}

static float *conv_temp(hcvo_conv_f32 *c, size_t n)
{
    if (n > c->temp_frame)
    {
        free(c->temp);
        c->temp = (float *) malloc(sizeof(float) * (c->num_ins + 2) * n);
        c->temp_frame = c->temp ? n : 0;
    }
    return c->temp;
}

void hcvo_conv_process_f32(hcvo_conv_f32 *c, const float *const *ins, float **outs, size_t num_ins, size_t num_outs, size_t n)  /* :138-154 */
{
    float *temp = conv_temp(c, n);
    if (!temp && n) return;
    float *temp1 = temp + (size_t) c->num_ins * c->temp_frame;
    for (size_t o = 0; o < num_outs; o++)
    {
        if (c->n2m)
            hcvo_n2m_process_f32(c->conv[o], ins, outs[o], temp1, n, num_ins);
        else
        {
            const float *one[1] = { ins[o] };
            hcvo_n2m_process_f32(c->conv[o], one, outs[o], temp1, n, 1);
        }
    }
}

void hcvo_conv_process_f64(hcvo_conv_f32 *c, const double *const *ins, double **outs, size_t num_ins, size_t num_outs, size_t n)  /* :156-183 */
{
    float *temp = conv_temp(c, n);
    if (!temp && n) return;
    size_t frame = c->temp_frame;
    float *temp1 = temp + (size_t) c->num_ins * frame, *temp2 = temp1 + frame;
    if (num_ins > c->num_ins) num_ins = c->num_ins;
    if (num_outs > c->num_outs) num_outs = c->num_outs;
    const float **in_ptrs = (const float **) malloc(sizeof(float *) * c->num_ins);
    for (size_t i = 0; i < c->num_ins; i++) in_ptrs[i] = temp + i * frame;
    for (size_t i = 0; i < num_ins; i++)
        for (size_t j = 0; j < n; j++) temp[i * frame + j] = (float) ins[i][j];
    for (size_t o = 0; o < num_outs; o++)
    {
        const float *one[1] = { in_ptrs[o < c->num_ins ? o : 0] };
        if (c->n2m) hcvo_n2m_process_f32(c->conv[o], in_ptrs, temp2, temp1, n, num_ins);
        else hcvo_n2m_process_f32(c->conv[o], one, temp2, temp1, n, 1);
        for (size_t j = 0; j < n; j++) outs[o][j] = temp2[j];
    }
    free(in_ptrs);
}

/* ------------------------------------------------------------------------------------------------
 * Streaming helpers (used for "port" CPU timing and to keep Python loops out of the tests)
 * ---------------------------------------------------------------------------------------------- */

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

/* ins: [num_ins][total], outs: [num_outs][total]; returns wall seconds */
double hcvo_conv_stream_f32(hcvo_conv_f32 *c, const float *ins, float *outs, size_t num_ins, size_t num_outs, size_t total, size_t block)
{
    const float **ip = (const float **) malloc(sizeof(float *) * (num_ins ? num_ins : 1));
    float **op = (float **) malloc(sizeof(float *) * (num_outs ? num_outs : 1));
    double t0 = now_s();
    for (size_t pos = 0; pos < total; pos += block)
    {
        size_t n = (total - pos) < block ? (total - pos) : block;
        for (size_t i = 0; i < num_ins; i++) ip[i] = ins + i * total + pos;
        for (size_t o = 0; o < num_outs; o++) op[o] = outs + o * total + pos;
        hcvo_conv_process_f32(c, ip, op, num_ins, num_outs, n);
    }
    double t1 = now_s();
    free(ip);
    free(op);
    return t1 - t0;
}

/* ------------------------------------------------------------------------------------------------
 * Synthetic signals: raw mt19937 32-bit draws, u = (r >> 8) * 2^-24.
 * Bit-identical to numpy.random.MT19937(seed).random_raw() driven the same way.
 * ---------------------------------------------------------------------------------------------- */

typedef struct { uint32_t mt[624]; int idx; } hcvo_mt;

static void mt_seed(hcvo_mt *s, uint32_t seed)
{
    s->mt[0] = seed;
    for (int i = 1; i < 624; i++) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t) i;
    s->idx = 624;
}

static uint32_t mt_next(hcvo_mt *s)
{
    if (s->idx >= 624)
    {
        for (int i = 0; i < 624; i++)
        {
            uint32_t y = (s->mt[i] & 0x80000000u) | (s->mt[(i + 1) % 624] & 0x7fffffffu);
            s->mt[i] = s->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}

/* audio(ch): seed 777 + ch, x[n] = 2u - 1 */
void hcvo_synth_audio_f32(uint32_t ch, float *x, size_t n)
{
    hcvo_mt s;
    mt_seed(&s, 777u + ch);
    for (size_t i = 0; i < n; i++)
        x[i] = (float) (2.0 * ((double) (mt_next(&s) >> 8) * (1.0 / 16777216.0)) - 1.0);
}

/* IR(in,out): seed 1000*in + out + 1, h[k] = (2u-1) * 10^(-3k/L), scaled to unit L2 norm */
void hcvo_synth_ir_f32(uint32_t in, uint32_t out, float *h, size_t len)
{
    hcvo_mt s;
    mt_seed(&s, 1000u * in + out + 1u);
    double energy = 0.0;
    double *tmp = (double *) malloc(sizeof(double) * (len ? len : 1));
    for (size_t k = 0; k < len; k++)
    {
        double u = (double) (mt_next(&s) >> 8) * (1.0 / 16777216.0);
        tmp[k] = (2.0 * u - 1.0) * pow(10.0, -3.0 * (double) k / (double) len);
        energy += tmp[k] * tmp[k];
    }
    double g = energy > 0.0 ? 1.0 / sqrt(energy) : 0.0;
    for (size_t k = 0; k < len; k++) h[k] = (float) (tmp[k] * g);
    free(tmp);
}

/* ------------------------------------------------------------------------------------------------
 * spectral_processor::convolve / correlate, real float overloads ("next" row)
 * (SpectralProcessor.hpp:164-184 entry points, :318-357 op_sizes, :361-383 fold, :448-539 arrange,
 *  :617-674 binary_op; SpectralFunctions.hpp:63-83 real_operation, :265-281 convolve/correlate ops)
 * ---------------------------------------------------------------------------------------------- */

enum { HCVO_EDGE_LINEAR = 0, HCVO_EDGE_WRAP = 1, HCVO_EDGE_WRAP_CENTRE = 2, HCVO_EDGE_FOLD = 3, HCVO_EDGE_FOLD_REPEAT = 4 };

typedef struct
{
    int mode, fold;
    size_t size1, size2, min, max, linear, fold_copy, fft;
    unsigned fft_log2;
} hcvo_op_sizes;

/* calc_fft_size_log2 (:231-243) */
static unsigned spectral_log2(size_t size)
{
    unsigned count = 0;
    while (count < 8 * sizeof(size_t) && (size >> count)) count++;
    if (count && size == ((size_t) 1 << (count - 1))) return count - 1;
    return count;
}

static hcvo_op_sizes spectral_sizes(size_t n1, size_t n2, int mode)
{
    hcvo_op_sizes s;
    s.mode = mode;
    s.fold = (mode == HCVO_EDGE_FOLD || mode == HCVO_EDGE_FOLD_REPEAT);
    s.size1 = n1; s.size2 = n2;
    s.min = n1 < n2 ? n1 : n2;
    s.max = n1 < n2 ? n2 : n1;
    s.linear = n1 + n2 - 1;
    s.fold_copy = s.max + ((s.min >> 1) << 1);
    s.fft_log2 = spectral_log2(s.fold ? s.fold_copy + (s.min - 1) : s.linear);
    s.fft = (size_t) 1 << s.fft_log2;
    return s;
}

/* calc_conv_corr_size (:549-560) with an explicit FFT-size cap */
size_t hcvo_spectral_size(size_t n1, size_t n2, int mode, size_t max_fft)
{
    if (!n1 || !n2) return 0;
    hcvo_op_sizes s = spectral_sizes(n1, n2, mode);
    if (s.fft > max_fft) return 0;
    return mode != HCVO_EDGE_LINEAR ? s.max : s.linear;
}

#define REAL float
#define FN(x) x##_f32
#include "hcv_oracle_spectral.inc"
#undef REAL
#undef FN

#define REAL double
#define FN(x) x##_f64
#include "hcv_oracle_spectral.inc"
#undef REAL
#undef FN
