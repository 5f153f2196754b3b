// Compiles against the drop-in headers exactly as a caller of the reference would (same includes modulo the
// directory, same class names and signatures) and, when a GPU is present, runs a tiny 2x2 zero-latency convolution
// with impulse IRs whose exact answer is known.  Exit codes: 0 ok, 2 no GPU (compile/link check only), 1 wrong result.
#include "hisstools_amd/Convolver.h"
#include "hisstools_amd/PartitionedConvolve.h"
#include "hisstools_amd/TimeDomainConvolve.h"
#include "hisstools_amd/HISSTools_FFT.h"
#include "hisstools_amd/SpectralProcessor.h"

#include <cmath>
#include <cstdio>
#include <vector>

int main()
{
    if (hcv_device_count() <= 0)
    {
        std::printf("no GPU: link check only\n");
        return 2;
    }
    const uint32_t nin = 2, nout = 2;
    const size_t L = 20000, S = 3 * 8192;
    HISSTools::Convolver conv(nin, nout, kLatencyZero);
    const size_t delay[2][2] = { { 0, 130 }, { 9000, 17000 } };
    const float gain[2][2] = { { 0.5f, -0.25f }, { 0.75f, 0.6f } };
    std::vector<float> ir(L, 0.f);
    for (uint32_t o = 0; o < nout; o++)
        for (uint32_t i = 0; i < nin; i++)
        {
            ir[delay[o][i]] = gain[o][i];
            if (conv.set(i, o, ir.data(), L, true) != CONVOLVE_ERR_NONE) return 1;
            ir[delay[o][i]] = 0.f;
        }
    if (conv.set(2, 0, ir.data(), L, true) != CONVOLVE_ERR_IN_CHAN_OUT_OF_RANGE) return 1;

    std::vector<std::vector<float>> x(nin, std::vector<float>(S)), y(nout, std::vector<float>(S, 0.f));
    unsigned s = 12345;
    for (auto &row : x)
        for (auto &v : row) { s = s * 1664525u + 1013904223u; v = (float) ((s >> 8) * (1.0 / 16777216.0) * 2.0 - 1.0); }

    for (size_t pos = 0; pos < S; pos += 512)
    {
        const float *ins[2] = { x[0].data() + pos, x[1].data() + pos };
        float *outs[2] = { y[0].data() + pos, y[1].data() + pos };
        conv.process(ins, outs, nin, nout, 512);
    }
    double worst = 0.0;
    for (uint32_t o = 0; o < nout; o++)
        for (size_t n = 0; n < S; n++)
        {
            double t = 0.0;
            for (uint32_t i = 0; i < nin; i++)
                if (n >= delay[o][i]) t += (double) gain[o][i] * x[i][n - delay[o][i]];
            worst = std::fmax(worst, std::fabs(t - y[o][n]));
        }
    std::printf("drop-in Convolver 2x2: max abs error %.3e\n", worst);
    if (!(worst < 5e-6)) return 1;

    // spectral_processor<float>::convolve (SpectralProcessor.hpp:178-181) through the drop-in header
    spectral_processor<float> sp;
    const float a[5] = { 1.f, 2.f, 3.f, 4.f, 5.f }, b[3] = { 1.f, -1.f, 0.5f };
    float lin[7] = { 0 };
    if (sp.convolved_size(5, 3, spectral_processor<float>::EdgeMode::Linear) != 7) return 1;
    sp.convolve(lin, { a, 5 }, { b, 3 }, spectral_processor<float>::EdgeMode::Linear);
    const float expect[7] = { 1.f, 1.f, 1.5f, 2.f, 2.5f, -3.f, 2.5f };
    for (int i = 0; i < 7; i++)
        if (std::fabs(lin[i] - expect[i]) > 1e-5f) return 1;
    std::printf("drop-in spectral_processor::convolve ok\n");

    // the double and the complex overloads (SpectralProcessor.hpp:164-184) and the transform members (:117-160)
    spectral_processor<double> spd;
    using ModeD = spectral_processor<double>::EdgeMode;
    const double ad[5] = { 1., 2., 3., 4., 5. }, bd[3] = { 1., -1., 0.5 }, zero3[3] = { 0., 0., 0. };
    double lind[7] = { 0 }, ro[7] = { 0 }, io[7] = { 0 };
    spd.convolve(lind, { ad, 5 }, { bd, 3 }, ModeD::Linear);
    for (int i = 0; i < 7; i++)
        if (std::fabs(lind[i] - (double) expect[i]) > 1e-12) return 1;
    // (a + j a) * (b + j 0) = a*b + j a*b ; correlate of a real signal with itself peaks at lag 0 with the energy
    spd.convolve(ro, io, { ad, 5 }, { ad, 5 }, { bd, 3 }, { zero3, 0 }, ModeD::Linear);
    for (int i = 0; i < 7; i++)
        if (std::fabs(ro[i] - (double) expect[i]) > 1e-12 || std::fabs(io[i] - (double) expect[i]) > 1e-12) return 1;
    double cr[9] = { 0 }, ci[9] = { 0 };
    spd.correlate(cr, ci, { ad, 5 }, { zero3, 0 }, { ad, 5 }, { zero3, 0 }, ModeD::Linear);
    if (std::fabs(cr[0] - 55.0) > 1e-11 || std::fabs(ci[0]) > 1e-11) return 1;
    float fr[8] = { 1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f }, fi[8] = { 0.f };
    FFT_SPLIT_COMPLEX_F split(fr, fi);
    sp.fft(split, 3);                                           // the transform of an impulse is flat
    for (int i = 0; i < 8; i++)
        if (std::fabs(fr[i] - 1.f) > 1e-6f || std::fabs(fi[i]) > 1e-6f) return 1;
    sp.ifft(split, 3);
    if (std::fabs(fr[0] - 8.f) > 1e-5f) return 1;
    std::printf("drop-in spectral_processor double / complex overloads and transforms ok\n");
    return 0;
}
