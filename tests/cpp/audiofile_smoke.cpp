// A caller written against the reference's HISSTools::IAudioFile / OAudioFile compiles against the drop-in header, writes
// a 3-channel 24-bit AIFC file one channel at a time, reads it back and loads channel 1 into a Convolver as an impulse
// response.  Exit 0 ok, 1 failure.  (No GPU needed for the file part; the Convolver part is skipped without one.)
#include "hisstools_amd/AudioFile.h"
#include "hisstools_amd/Convolver.h"

#include <cmath>
#include <cstdio>
#include <vector>

int main(int argc, char **argv)
{
    const char *path = argc > 1 ? argv[1] : "/tmp/hcv_audiofile_smoke.aifc";
    const uint32_t frames = 500;
    const uint16_t channels = 3;
    std::vector<std::vector<double>> ch(channels, std::vector<double>(frames));
    for (uint16_t c = 0; c < channels; c++)
        for (uint32_t i = 0; i < frames; i++) ch[c][i] = 0.9 * std::sin(0.01 * (c + 1) * i) * std::exp(-0.004 * i);

    HISSTools::OAudioFile out(path, HISSTools::BaseAudioFile::kAudioFileAIFF, HISSTools::BaseAudioFile::kAudioFileInt24, channels, 48000.0);
    if (!out.isOpen() || out.getFileType() != HISSTools::BaseAudioFile::kAudioFileAIFC) return 1;   // AIFF requests write AIFC
    for (uint16_t c = 0; c < channels; c++)
    {
        out.seek(0);
        out.writeChannel(ch[c].data(), frames, c);
    }
    if (out.getIsError() || out.getFrames() != frames) return 1;
    out.close();

    HISSTools::IAudioFile in(path);
    if (!in.isOpen() || in.getIsError() || in.getChannels() != channels || in.getFrames() != frames || in.getBitDepth() != 24 || in.getSamplingRate() != 48000.0)
    {
        std::printf("header mismatch: %s\n", HISSTools::BaseAudioFile::getErrorString(in.getErrors().empty() ? HISSTools::BaseAudioFile::ERR_NONE : in.getErrors()[0]).c_str());
        return 1;
    }
    std::vector<float> ir(frames);
    in.seek(0);
    in.readChannel(ir.data(), frames, 1);
    for (uint32_t i = 0; i < frames; i++)
        if (std::fabs(ir[i] - ch[1][i]) > 1.0 / (1 << 22)) return 1;          // 24-bit quantisation
    std::vector<double> all((size_t) frames * channels);
    in.seek(0);
    in.readInterleaved(all.data(), frames);
    for (uint32_t i = 0; i < frames; i++)
        if (std::fabs(all[(size_t) i * channels + 2] - ch[2][i]) > 1.0 / (1 << 22)) return 1;

    if (hcv_device_count() <= 0)
    {
        std::printf("audio file round trip ok (no GPU: convolver part skipped)\n");
        return 0;
    }
    HISSTools::Convolver conv(1u, 1u, kLatencyZero);
    if (conv.set(0, 0, ir.data(), frames, true) != CONVOLVE_ERR_NONE) return 1;
    std::vector<float> x(2048, 0.f), y(2048, 0.f);
    x[0] = 1.f;
    const float *ins[1] = { x.data() };
    float *outs[1] = { y.data() };
    conv.process(ins, outs, 1, 1, x.size());
    for (uint32_t i = 0; i < frames; i++)
        if (std::fabs(y[i] - ir[i]) > 2e-6f) return 1;
    std::printf("audio file round trip + convolver load ok\n");
    return 0;
}
