// TEST INFRASTRUCTURE — not part of the product path.
//
// C-ABI driver around the unmodified reference's audio file classes (AudioFile/{Base,I,O}AudioFile.{h,cpp}) — the third
// "next" row of SURVEY.md §8f.  Built by oracle/Makefile into oracle/_ref/libhisstools_ref_audio.so from the sources where
// they lie.  Used to write the golden files of tests/golden/audio and to cross-check the product reader / writer.

#include "AudioFile/IAudioFile.h"
#include "AudioFile/OAudioFile.h"

#include <cstdint>

using namespace HISSTools;

extern "C"
{
    struct ref_audio_info
    {
        int file_type, pcm_format, header_endianness, audio_endianness;
        double sampling_rate;
        unsigned channels, frames, bit_depth;
        int error_flags;
    };

    // mode 0: one writeInterleaved call; 1: writeChannel per channel (seek(0) between); 2: interleaved in two calls
    int ref_audio_write(const char *path, int type, int format, unsigned channels, double rate, int endianness, const double *data, unsigned frames,
                        int mode, int as_float)
    {
        OAudioFile f;
        if (endianness < 0)
            f.open(path, static_cast<BaseAudioFile::FileType>(type), static_cast<BaseAudioFile::PCMFormat>(format), channels, rate);
        else
            f.open(path, static_cast<BaseAudioFile::FileType>(type), static_cast<BaseAudioFile::PCMFormat>(format), channels, rate,
                   static_cast<BaseAudioFile::Endianness>(endianness));
        if (!f.isOpen()) return f.getErrorFlags() | (1 << 30);
        std::vector<float> fl(data, data + (size_t) frames * channels);
        if (mode == 0)
        {
            if (as_float) f.writeInterleaved(fl.data(), frames); else f.writeInterleaved(data, frames);
        }
        else if (mode == 2)
        {
            const unsigned a = frames / 3;
            if (as_float) { f.writeInterleaved(fl.data(), a); f.writeInterleaved(fl.data() + (size_t) a * channels, frames - a); }
            else { f.writeInterleaved(data, a); f.writeInterleaved(data + (size_t) a * channels, frames - a); }
        }
        else
        {
            std::vector<double> ch(frames);
            std::vector<float> chf(frames);
            for (unsigned c = 0; c < channels; c++)
            {
                for (unsigned i = 0; i < frames; i++) { ch[i] = data[(size_t) i * channels + c]; chf[i] = (float) ch[i]; }
                f.seek(0);
                if (as_float) f.writeChannel(chf.data(), frames, c); else f.writeChannel(ch.data(), frames, c);
            }
        }
        int flags = f.getErrorFlags();
        f.close();
        return flags;
    }

    int ref_audio_info(const char *path, ref_audio_info *out)
    {
        IAudioFile f(path);
        out->file_type = f.getFileType();
        out->pcm_format = f.getPCMFormat();
        out->header_endianness = f.getHeaderEndianness();
        out->audio_endianness = f.getAudioEndianness();
        out->sampling_rate = f.getSamplingRate();
        out->channels = f.getChannels();
        out->frames = f.getFrames();
        out->bit_depth = f.getBitDepth();
        out->error_flags = f.getErrorFlags();
        return f.isOpen() ? 0 : -1;
    }

    // channel < 0: interleaved
    int ref_audio_read_f64(const char *path, double *out, unsigned first, unsigned frames, int channel)
    {
        IAudioFile f(path);
        if (!f.isOpen() || f.getIsError()) return -1;
        f.seek(first);
        if (channel < 0) f.readInterleaved(out, frames); else f.readChannel(out, frames, (uint16_t) channel);
        return 0;
    }

    int ref_audio_read_f32(const char *path, float *out, unsigned first, unsigned frames, int channel)
    {
        IAudioFile f(path);
        if (!f.isOpen() || f.getIsError()) return -1;
        f.seek(first);
        if (channel < 0) f.readInterleaved(out, frames); else f.readChannel(out, frames, (uint16_t) channel);
        return 0;
    }
}
