// Audio file I/O (third "next" row): WAVE / AIFF / AIFC reading and WAVE / AIFC writing with the
// semantics of the reference's IAudioFile / OAudioFile (AudioFile/{Base,I,O}AudioFile.{h,cpp}), so that real impulse
// responses can be loaded into the convolver and results stored.  Host-side only by nature: nothing here touches the GPU.
//
// Behaviour restated from the reference:
//   reader   IAudioFile.cpp:384-404  RIFF / RIFX + WAVE, FORM + AIFF / AIFC
//            :406-539  AIFF / AIFC: FVER required for AIFC (version 0xA2805140), COMM (channels, frames, bit depth, 80-bit
//                      sampling rate, compression NONE / twos / sowt / fl32 / FL32 / fl64 / FL64), SSND offset
//            :541-595  WAVE: first 16 bytes of "fmt " (format 1 = int, 3 = float), "data"; frames = bytes / frame size
//            :600-667  sample conversion: integers scaled by 2^-(bits-1) through a 32-bit left-justified value, WAVE 8-bit
//                      unsigned (v - 128) / 128, AIFF 8-bit signed, float32 / float64 copied
//   writer   OAudioFile.cpp:374-470  WAVE header (44 bytes) / AIFC header (FVER + COMM with compression name + SSND), both
//                      rewritten after every write that extends the file; AIFF requests produce AIFC
//            :539-560  integer quantisation round(x * 2^(bits-1)) WITHOUT clipping (the reference computes the clip and
//                      discards it, :549), so +1.0 wraps to the most negative value; WAVE 8-bit is clipped
//            :562-660  interleaved or single-channel writes; a single-channel write first zero-extends the file
// Deliberately different: reads past the end of the data deliver zeros (the reference leaves its work buffer's stale
// samples), and AIFC "fl64" is read as 64-bit floats (the reference's reader sets 32 bits for it, :372-376, and so
// misreads the files its own writer produces).

#include "../../include/hisstools_amd.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace
{
    enum { TYPE_NONE = 0, TYPE_AIFF = 1, TYPE_AIFC = 2, TYPE_WAVE = 3 };
    enum { FMT_I8 = 0, FMT_I16, FMT_I24, FMT_I32, FMT_F32, FMT_F64 };
    enum { LITTLE = 0, BIG = 1 };
    enum
    {
        E_OPEN = 1 << 2, E_BAD_FORMAT = 1 << 3, E_UNKNOWN_FORMAT = 1 << 4, E_UNSUPPORTED_PCM = 1 << 5, E_AIFC_VERSION = 1 << 6,
        E_AIFC_FORMAT = 1 << 7, E_WAVE_FORMAT = 1 << 8, E_WRITE = 1 << 9
    };
    const uint32_t kAifcVersion = 0xA2805140u;

    int bits_of(int fmt)
    {
        switch (fmt)
        {
            case FMT_I8: return 8;
            case FMT_I24: return 24;
            case FMT_I32: case FMT_F32: return 32;
            case FMT_F64: return 64;
            default: return 16;
        }
    }
    bool is_float(int fmt) { return fmt == FMT_F32 || fmt == FMT_F64; }

    uint64_t get_uint(const unsigned char *b, int bytes, int endian)
    {
        uint64_t v = 0;
        for (int i = 0; i < bytes; i++) v = (v << 8) | b[endian == BIG ? i : bytes - 1 - i];
        return v;
    }

    void put_uint(unsigned char *b, uint64_t v, int bytes, int endian)
    {
        for (int i = 0; i < bytes; i++) b[endian == BIG ? bytes - 1 - i : i] = (unsigned char) (v >> (8 * i));
    }

    // 80-bit IEEE extended, big-endian (IAudioFile.cpp:187-214 / OAudioFile.cpp:284-340)
    double extended_to_double(const unsigned char *b)
    {
        const bool sign = b[0] & 0x80;
        int exponent = (int) get_uint(b, 2, BIG) & 0x7FFF;
        const uint32_t hi = (uint32_t) get_uint(b + 2, 4, BIG), lo = (uint32_t) get_uint(b + 6, 4, BIG);
        if (!exponent && !hi && !lo) return 0.0;
        if (exponent == 0x7FFF) return HUGE_VAL;
        exponent -= 16383;
        double v = std::ldexp((double) hi, exponent - 31) + std::ldexp((double) lo, exponent - 63);
        return sign ? -v : v;
    }

    void double_to_extended(double num, unsigned char *b)
    {
        int sign = 0, expon = 0;
        uint32_t hi = 0, lo = 0;
        if (num < 0) { sign = 0x8000; num = -num; }
        if (num != 0)
        {
            double mant = std::frexp(num, &expon);
            if (expon > 16384 || !(mant < 1)) expon = sign | 0x7FFF;
            else
            {
                expon += 16382;
                if (expon < 0) { mant = std::ldexp(mant, expon); expon = 0; }
                expon |= sign;
                mant = std::ldexp(mant, 32);
                double whole = std::floor(mant);
                hi = (uint32_t) whole;
                mant = std::ldexp(mant - whole, 32);
                lo = (uint32_t) std::floor(mant);
            }
        }
        put_uint(b, (uint64_t) expon, 2, BIG);
        put_uint(b + 2, hi, 4, BIG);
        put_uint(b + 6, lo, 4, BIG);
    }

    struct Common
    {
        int type = TYPE_NONE, format = FMT_I8, header_endian = LITTLE, audio_endian = LITTLE;
        double rate = 0;
        uint32_t channels = 0, frames = 0;
        long pcm_offset = 0;
        int errors = 0;
        FILE *fp = nullptr;

        size_t byte_depth() const { return (size_t) bits_of(format) / 8; }
        size_t frame_bytes() const { return channels * byte_depth(); }
        void reset()
        {
            *this = Common();
        }
    };
}

struct hcv_audiofile
{
    Common c;
    bool writer = false;
    std::vector<unsigned char> work;
};

namespace
{
    // ------------------------------------------------------------------------------------------------ reading

    bool read_bytes(Common &c, void *dst, size_t n) { return std::fread(dst, 1, n, c.fp) == n; }
    bool skip(Common &c, long n) { return std::fseek(c.fp, n, SEEK_CUR) == 0; }
    long padded(long n) { return n + (n & 1); }

    void set_pcm(Common &c, int bits, bool flt)
    {
        int f = -1;
        if (!flt) f = bits == 8 ? FMT_I8 : bits == 16 ? FMT_I16 : bits == 24 ? FMT_I24 : bits == 32 ? FMT_I32 : -1;
        else f = bits == 32 ? FMT_F32 : bits == 64 ? FMT_F64 : -1;
        if (f < 0) c.errors |= E_UNSUPPORTED_PCM;
        else c.format = f;
    }

    void parse_aiff(Common &c, bool aifc)
    {
        enum { T_VER = 1, T_COMM = 2, T_SSND = 4 };
        unsigned need = T_COMM | T_SSND | (aifc ? T_VER : 0), seen = 0;
        c.header_endian = BIG;
        if (aifc) c.type = TYPE_AIFC;
        unsigned char head[8], chunk[22];
        while (read_bytes(c, head, 8))
        {
            const long size = (long) get_uint(head + 4, 4, BIG);
            const long body = std::ftell(c.fp);
            if (!std::memcmp(head, "FVER", 4))
            {
                seen |= T_VER;
                if (size < 4 || !read_bytes(c, chunk, 4)) { c.errors |= E_BAD_FORMAT; return; }
                if (get_uint(chunk, 4, BIG) != kAifcVersion) { c.errors |= E_AIFC_VERSION; return; }
            }
            else if (!std::memcmp(head, "COMM", 4))
            {
                seen |= T_COMM;
                // plain AIFF: 18 bytes; AIFC adds the 4-byte compression tag, which must be there to be compared
                const long want = aifc ? 22 : 18;
                if (want > size || !read_bytes(c, chunk, (size_t) want)) { c.errors |= E_BAD_FORMAT; return; }
                c.channels = (uint32_t) get_uint(chunk, 2, BIG);
                c.frames = (uint32_t) get_uint(chunk + 2, 4, BIG);
                int bits = (int) get_uint(chunk + 6, 2, BIG);
                c.rate = extended_to_double(chunk + 8);
                bool flt = false;
                c.audio_endian = BIG;
                if (!c.frames) seen |= T_SSND;                  // no audio chunk needed for an empty file
                if (aifc)
                {
                    const unsigned char *t = chunk + 18;
                    if (!std::memcmp(t, "NONE", 4)) {}
                    else if (!std::memcmp(t, "twos", 4)) bits = 16;
                    else if (!std::memcmp(t, "sowt", 4)) { bits = 16; c.audio_endian = LITTLE; }
                    else if (!std::memcmp(t, "fl32", 4) || !std::memcmp(t, "FL32", 4)) { bits = 32; flt = true; }
                    else if (!std::memcmp(t, "fl64", 4) || !std::memcmp(t, "FL64", 4)) { bits = 64; flt = true; }
                    else { c.errors |= E_AIFC_FORMAT; return; }
                }
                else
                    c.type = TYPE_AIFF;
                set_pcm(c, bits, flt);
                if (c.errors) return;
            }
            else if (!std::memcmp(head, "SSND", 4))
            {
                seen |= T_SSND;
                if (size < 4 || !read_bytes(c, chunk, 4)) { c.errors |= E_BAD_FORMAT; return; }
                c.pcm_offset = body + 8 + (long) get_uint(chunk, 4, BIG);
            }
            if (std::fseek(c.fp, body + padded(size), SEEK_SET) != 0) { c.errors |= E_BAD_FORMAT; return; }
        }
        if (~seen & need) c.errors |= E_BAD_FORMAT;
    }

    bool find_chunk(Common &c, const char *tag, long &size)
    {
        unsigned char head[8];
        while (read_bytes(c, head, 8))
        {
            size = (long) get_uint(head + 4, 4, c.header_endian);
            if (!std::memcmp(head, tag, 4)) return true;
            if (!skip(c, padded(size))) return false;
        }
        return false;
    }

    void parse_wave(Common &c, bool rifx)
    {
        c.header_endian = c.audio_endian = rifx ? BIG : LITTLE;
        unsigned char fmt[16];
        long size = 0;
        if (!find_chunk(c, "fmt ", size) || size < 16 || !read_bytes(c, fmt, 16) || !skip(c, padded(size) - 16))
        {
            c.errors |= E_BAD_FORMAT;
            return;
        }
        const unsigned tag = (unsigned) get_uint(fmt, 2, c.header_endian);
        if (tag != 1 && tag != 3) { c.errors |= E_WAVE_FORMAT; return; }
        c.channels = (uint32_t) get_uint(fmt + 2, 2, c.header_endian);
        c.rate = (double) get_uint(fmt + 4, 4, c.header_endian);
        set_pcm(c, (int) get_uint(fmt + 14, 2, c.header_endian), tag == 3);
        if (c.errors) return;
        if (!find_chunk(c, "data", size)) { c.errors |= E_BAD_FORMAT; return; }
        c.frames = c.frame_bytes() ? (uint32_t) ((unsigned long) size / c.frame_bytes()) : 0;
        c.pcm_offset = std::ftell(c.fp);
        c.type = TYPE_WAVE;
    }

    void parse_header(Common &c)
    {
        unsigned char h[12];
        if (!read_bytes(c, h, 12)) { c.errors |= E_BAD_FORMAT; return; }
        if (!std::memcmp(h, "FORM", 4) && (!std::memcmp(h + 8, "AIFF", 4) || !std::memcmp(h + 8, "AIFC", 4)))
            parse_aiff(c, !std::memcmp(h + 8, "AIFC", 4));
        else if ((!std::memcmp(h, "RIFF", 4) || !std::memcmp(h, "RIFX", 4)) && !std::memcmp(h + 8, "WAVE", 4))
            parse_wave(c, !std::memcmp(h, "RIFX", 4));
        else
            c.errors |= E_UNKNOWN_FORMAT;
    }

    template <class T> T decode(const Common &c, const unsigned char *b)
    {
        const int bytes = (int) c.byte_depth();
        switch (c.format)
        {
            case FMT_I8:
                if (c.type == TYPE_WAVE) return (T(b[0]) - T(128)) / T(128);
                return (T) (int32_t) ((uint32_t) b[0] << 24) * (T(-1.0) / (T) (int32_t) 0x80000000);
            case FMT_I16: case FMT_I24: case FMT_I32:
            {
                const uint32_t v = (uint32_t) get_uint(b, bytes, c.audio_endian) << (32 - 8 * bytes);
                return (T) (int32_t) v * (T(-1.0) / (T) (int32_t) 0x80000000);
            }
            case FMT_F32:
            {
                const uint32_t v = (uint32_t) get_uint(b, 4, c.audio_endian);
                float f;
                std::memcpy(&f, &v, 4);
                return (T) f;
            }
            default:
            {
                const uint64_t v = get_uint(b, 8, c.audio_endian);
                double d;
                std::memcpy(&d, &v, 8);
                return (T) d;
            }
        }
    }

    template <class T> void read_audio(hcv_audiofile *h, T *out, uint32_t frames, int channel)
    {
        Common &c = h->c;
        if (!c.fp || !c.frame_bytes()) return;
        const size_t depth = c.byte_depth(), fb = c.frame_bytes();
        const size_t per_frame = channel < 0 ? c.channels : 1;
        const size_t first = channel < 0 ? 0 : (size_t) channel * depth, step = channel < 0 ? depth : fb;
        const uint32_t kChunk = 4096;
        h->work.resize(kChunk * fb);
        while (frames)
        {
            const uint32_t n = std::min(frames, kChunk);
            const size_t got = std::fread(h->work.data(), 1, n * fb, c.fp);
            if (got < n * fb) std::memset(h->work.data() + got, c.format == FMT_I8 && c.type == TYPE_WAVE ? 128 : 0, n * fb - got);
            const unsigned char *p = h->work.data() + first;
            for (size_t i = 0; i < n * per_frame; i++, p += step) out[i] = decode<T>(c, p);
            out += n * per_frame;
            frames -= n;
        }
    }

    // ------------------------------------------------------------------------------------------------ writing

    bool write_bytes(Common &c, const void *src, size_t n) { return std::fwrite(src, 1, n, c.fp) == n; }
    bool put(Common &c, uint64_t v, int bytes, int endian)
    {
        unsigned char b[8];
        put_uint(b, v, bytes, endian);
        return write_bytes(c, b, (size_t) bytes);
    }

    const char *compression_tag(int fmt) { return fmt == FMT_F32 ? "fl32" : fmt == FMT_F64 ? "fl64" : "NONE"; }
    const char *compression_name(int fmt) { return fmt == FMT_F32 ? "32-bit floating point" : fmt == FMT_F64 ? "64-bit floating point" : "not compressed"; }
    uint32_t pstring_bytes(const char *s)
    {
        const uint32_t n = (uint32_t) std::strlen(s);
        return ((n + 1) & 1) ? n + 2 : n + 1;
    }

    void write_header(Common &c)
    {
        bool ok = true;
        const int he = c.header_endian;
        if (c.type == TYPE_WAVE)
        {
            ok &= write_bytes(c, he == LITTLE ? "RIFF" : "RIFX", 4) && put(c, 36, 4, he) && write_bytes(c, "WAVE", 4);
            ok &= write_bytes(c, "fmt ", 4) && put(c, 16, 4, he);
            ok &= put(c, is_float(c.format) ? 3 : 1, 2, he) && put(c, c.channels, 2, he);
            ok &= put(c, (uint32_t) c.rate, 4, he) && put(c, (uint32_t) (c.rate * (double) c.frame_bytes()), 4, he);
            ok &= put(c, (uint16_t) c.frame_bytes(), 2, he) && put(c, (uint64_t) bits_of(c.format), 2, he);
            ok &= write_bytes(c, "data", 4) && put(c, 0, 4, he);
        }
        else
        {
            const char *name = compression_name(c.format);
            ok &= write_bytes(c, "FORM", 4) && put(c, 62 + pstring_bytes(name), 4, he) && write_bytes(c, "AIFC", 4);
            ok &= write_bytes(c, "FVER", 4) && put(c, 4, 4, he) && put(c, kAifcVersion, 4, he);
            ok &= write_bytes(c, "COMM", 4) && put(c, 22 + pstring_bytes(name), 4, he);
            ok &= put(c, c.channels, 2, he) && put(c, c.frames, 4, he) && put(c, (uint64_t) bits_of(c.format), 2, he);
            unsigned char ext[10];
            double_to_extended(c.rate, ext);
            ok &= write_bytes(c, ext, 10) && write_bytes(c, compression_tag(c.format), 4);
            const unsigned char len = (unsigned char) std::strlen(name);
            ok &= write_bytes(c, &len, 1) && write_bytes(c, name, len);
            if ((len + 1) & 1) ok &= put(c, 0, 1, he);
            ok &= write_bytes(c, "SSND", 4) && put(c, 8, 4, he) && put(c, 0, 4, he) && put(c, 0, 4, he);
        }
        c.pcm_offset = std::ftell(c.fp);
        if (!ok) c.errors |= E_WRITE;
    }

    uint32_t write_position(const Common &c)
    {
        if (!c.pcm_offset || !c.frame_bytes()) return 0;
        return (uint32_t) ((size_t) (std::ftell(c.fp) - c.pcm_offset) / c.frame_bytes());
    }

    // OAudioFile::updateHeader (:472-515): after a write that extended the file, patch the sizes (and pad to even length)
    bool update_header(Common &c)
    {
        const uint32_t end = write_position(c);
        if (end <= c.frames) return true;
        c.frames = end;
        const long data_bytes = (long) (c.frame_bytes() * c.frames), data_end = std::ftell(c.fp);
        const int he = c.header_endian;
        bool ok = true;
        if (data_bytes & 1) ok &= put(c, 0, 1, he);
        const uint32_t riff = (uint32_t) ((c.pcm_offset - 8) + padded(data_bytes));
        ok &= std::fseek(c.fp, 4, SEEK_SET) == 0 && put(c, riff, 4, he);
        if (c.type == TYPE_WAVE)
            ok &= std::fseek(c.fp, c.pcm_offset - 4, SEEK_SET) == 0 && put(c, (uint32_t) data_bytes, 4, he);
        else
        {
            ok &= std::fseek(c.fp, 34, SEEK_SET) == 0 && put(c, c.frames, 4, he);
            ok &= std::fseek(c.fp, c.pcm_offset - 12, SEEK_SET) == 0 && put(c, (uint32_t) data_bytes + 8, 4, he);
        }
        ok &= std::fseek(c.fp, data_end, SEEK_SET) == 0;
        return ok;
    }

    // OAudioFile::inputToU32 / inputToU8 (:539-560)
    uint32_t quantise(double x, int bits)
    {
        const double v = std::round(x * (double) (1u << (bits - 1)));
        return (uint32_t) (int64_t) v;                              // no clipping, two's-complement wrap — as the reference
    }
    uint8_t quantise_u8(double x)
    {
        return (uint8_t) std::min(std::max(std::round(x * 128.0 + 128.0), 0.0), 255.0);
    }

    template <class T> void write_audio(hcv_audiofile *h, const T *in, uint32_t frames, int channel)
    {
        Common &c = h->c;
        if (!c.fp || !c.frame_bytes() || !frames) return;
        const size_t depth = c.byte_depth(), fb = c.frame_bytes();
        const uint32_t start = write_position(c), end = start + frames;
        bool ok = true;
        const bool strided = channel >= 0 && c.channels > 1;
        if (strided && end > c.frames)                              // OAudioFile::resize (:517-537): zero-extend first
        {
            const long here = std::ftell(c.fp);
            ok &= std::fseek(c.fp, c.pcm_offset + (long) (fb * c.frames), SEEK_SET) == 0;
            std::vector<unsigned char> zeros(fb * (end - c.frames), 0);
            ok &= write_bytes(c, zeros.data(), zeros.size());
            ok &= std::fseek(c.fp, here, SEEK_SET) == 0;
        }
        const size_t samples = channel < 0 ? (size_t) c.channels * frames : frames;
        std::vector<unsigned char> &buf = h->work;
        buf.resize(samples * depth);
        for (size_t i = 0; i < samples; i++)
        {
            unsigned char *b = buf.data() + i * depth;
            switch (c.format)
            {
                case FMT_I8: b[0] = c.type == TYPE_WAVE ? quantise_u8((double) in[i]) : (unsigned char) quantise((double) in[i], 8); break;
                case FMT_I16: put_uint(b, quantise((double) in[i], 16), 2, c.audio_endian); break;
                case FMT_I24: put_uint(b, quantise((double) in[i], 24), 3, c.audio_endian); break;
                case FMT_I32: put_uint(b, quantise((double) in[i], 32), 4, c.audio_endian); break;
                case FMT_F32:
                {
                    const float f = (float) in[i];
                    uint32_t v;
                    std::memcpy(&v, &f, 4);
                    put_uint(b, v, 4, c.audio_endian);
                    break;
                }
                default:
                {
                    const double d = (double) in[i];
                    uint64_t v;
                    std::memcpy(&v, &d, 8);
                    put_uint(b, v, 8, c.audio_endian);
                }
            }
        }
        if (!strided)
            ok &= write_bytes(c, buf.data(), buf.size());
        else
        {
            const long first = (long) ((size_t) channel * depth), gap = (long) (fb - depth);
            ok &= std::fseek(c.fp, first, SEEK_CUR) == 0;
            for (size_t i = 0; i < samples; i++)
            {
                ok &= write_bytes(c, buf.data() + i * depth, depth);
                ok &= std::fseek(c.fp, i + 1 < samples ? gap : gap - first, SEEK_CUR) == 0;
            }
        }
        ok &= update_header(c);
        if (!ok) c.errors |= E_WRITE;
    }
}

// ---------------------------------------------------------------------------------------------------- C ABI

extern "C" hcv_audiofile *hcv_iaudiofile_open(const char *path)
{
    hcv_audiofile *h = new hcv_audiofile();
    if (!path || !*path) return h;
    h->c.fp = std::fopen(path, "rb");
    if (!h->c.fp)
    {
        h->c.errors |= E_OPEN;
        return h;
    }
    parse_header(h->c);
    std::fseek(h->c.fp, h->c.pcm_offset, SEEK_SET);
    return h;
}

extern "C" hcv_audiofile *hcv_oaudiofile_open(const char *path, int type, int format, unsigned channels, double sampling_rate, int endianness)
{
    hcv_audiofile *h = new hcv_audiofile();
    h->writer = true;
    if (!path || !*path) return h;
    Common &c = h->c;
    c.fp = std::fopen(path, "wb");
    if (!c.fp || format < FMT_I8 || format > FMT_F64 || (type != TYPE_WAVE && type != TYPE_AIFF && type != TYPE_AIFC))
    {
        if (c.fp) std::fclose(c.fp);
        c.fp = nullptr;
        c.errors |= E_OPEN;
        return h;
    }
    c.type = type == TYPE_AIFF ? TYPE_AIFC : type;                 // OAudioFile.cpp:62
    c.format = format;
    const int e = endianness < 0 ? (c.type == TYPE_WAVE ? LITTLE : BIG) : (endianness ? BIG : LITTLE);
    c.header_endian = c.type == TYPE_WAVE ? e : BIG;
    c.audio_endian = e;
    c.rate = sampling_rate;
    c.channels = channels & 0xFFFF;
    write_header(c);
    return h;
}

extern "C" void hcv_audiofile_close(hcv_audiofile *h)
{
    if (!h) return;
    if (h->c.fp) std::fclose(h->c.fp);
    delete h;
}

extern "C" int hcv_audiofile_is_open(const hcv_audiofile *h) { return h && h->c.fp ? 1 : 0; }

extern "C" int hcv_audiofile_get_info(const hcv_audiofile *h, hcv_audiofile_info *out)
{
    if (!h || !out) return -1;
    const Common &c = h->c;
    out->file_type = c.type;
    out->pcm_format = c.format;
    out->header_endianness = c.header_endian;
    out->audio_endianness = c.audio_endian;
    out->sampling_rate = c.rate;
    out->channels = c.channels;
    out->frames = c.frames;
    out->bit_depth = (unsigned) bits_of(c.format);
    out->error_flags = c.errors;
    return 0;
}

extern "C" void hcv_audiofile_seek(hcv_audiofile *h, uint32_t frame)
{
    if (!h || !h->c.fp) return;
    if (h->writer && !h->c.pcm_offset) return;                     // OAudioFile.cpp:92-96
    std::fseek(h->c.fp, h->c.pcm_offset + (long) (h->c.frame_bytes() * frame), SEEK_SET);
}

extern "C" uint32_t hcv_audiofile_position(hcv_audiofile *h)
{
    if (!h || !h->c.fp) return 0;
    return write_position(h->c);
}

extern "C" void hcv_iaudiofile_read_raw(hcv_audiofile *h, void *out, uint32_t frames)
{
    if (!h || !h->c.fp || h->writer) return;
    const size_t want = h->c.frame_bytes() * frames, got = std::fread(out, 1, want, h->c.fp);
    if (got < want) std::memset(static_cast<unsigned char *>(out) + got, 0, want - got);
}

extern "C" void hcv_iaudiofile_read_interleaved_f32(hcv_audiofile *h, float *out, uint32_t frames) { if (h && !h->writer) read_audio(h, out, frames, -1); }
extern "C" void hcv_iaudiofile_read_interleaved_f64(hcv_audiofile *h, double *out, uint32_t frames) { if (h && !h->writer) read_audio(h, out, frames, -1); }
extern "C" void hcv_iaudiofile_read_channel_f32(hcv_audiofile *h, float *out, uint32_t frames, unsigned channel)
{
    if (h && !h->writer && channel < h->c.channels) read_audio(h, out, frames, (int) channel);
}
extern "C" void hcv_iaudiofile_read_channel_f64(hcv_audiofile *h, double *out, uint32_t frames, unsigned channel)
{
    if (h && !h->writer && channel < h->c.channels) read_audio(h, out, frames, (int) channel);
}

extern "C" void hcv_oaudiofile_write_raw(hcv_audiofile *h, const void *in, uint32_t frames)
{
    if (!h || !h->writer || !h->c.fp) return;
    bool ok = write_bytes(h->c, in, h->c.frame_bytes() * frames);
    ok &= update_header(h->c);
    if (!ok) h->c.errors |= E_WRITE;
}

extern "C" void hcv_oaudiofile_write_interleaved_f32(hcv_audiofile *h, const float *in, uint32_t frames) { if (h && h->writer) write_audio(h, in, frames, -1); }
extern "C" void hcv_oaudiofile_write_interleaved_f64(hcv_audiofile *h, const double *in, uint32_t frames) { if (h && h->writer) write_audio(h, in, frames, -1); }
extern "C" void hcv_oaudiofile_write_channel_f32(hcv_audiofile *h, const float *in, uint32_t frames, unsigned channel)
{
    if (h && h->writer && channel < h->c.channels) write_audio(h, in, frames, (int) channel);
}
extern "C" void hcv_oaudiofile_write_channel_f64(hcv_audiofile *h, const double *in, uint32_t frames, unsigned channel)
{
    if (h && h->writer && channel < h->c.channels) write_audio(h, in, frames, (int) channel);
}
