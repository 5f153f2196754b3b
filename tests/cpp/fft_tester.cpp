// An FFT_Tester-style programme ("- Test/FFT_Tester/FFT_Tester/main.cpp" in the reference) written against the drop-in
// header: it exercises exactly the calls that tester makes — zip/unzip integer round trips for log2 1..23, then fft, ifft,
// rfft and rifft at every size in a range, in double and float — and additionally checks the numbers (the reference's
// tester only times them): small sizes against a direct O(N^2) DFT in double, all sizes by the inverse round trip.
// Exit code: 0 pass, 1 numerical failure, 2 no GPU (compile / link check only).
#include "hisstools_amd/HISSTools_FFT.h"
#include "hisstools_amd/SpectralFunctions.h"
#include "hisstools_amd/SpectralProcessor.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace
{
    double now()
    {
        return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }

    unsigned long long lcg = 0x9E3779B97F4A7C15ull;
    double noise()
    {
        lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
        return 1.0 - 2.0 * (double) (lcg >> 11) / 9007199254740992.0;
    }

    template <class T> struct Kit;
    template <> struct Kit<double> { typedef FFT_SETUP_D setup; typedef FFT_SPLIT_COMPLEX_D split; static double tol() { return 1e-12; } static const char *name() { return "DOUBLE"; } };
    template <> struct Kit<float> { typedef FFT_SETUP_F setup; typedef FFT_SPLIT_COMPLEX_F split; static double tol() { return 4e-6; } static const char *name() { return "FLOAT"; } };

    template <class T> bool zip_correctness(int min_log2, int max_log2)
    {
        std::vector<T> ptr(size_t(1) << max_log2), re(size_t(1) << (max_log2 - 1)), im(size_t(1) << (max_log2 - 1));
        typename Kit<T>::split split(re.data(), im.data());
        for (int i = min_log2; i < max_log2; i++)
        {
            for (long j = 0; j < (1L << i); j++) ptr[j] = (T) j;
            hisstools_unzip(ptr.data(), &split, i);
            for (long j = 0; j < (1L << (i - 1)); j++)
                if (re[j] != (T) (j << 1) || im[j] != (T) ((j << 1) + 1)) return false;
            for (long j = 0; j < (1L << i); j++) ptr[j] = (T) -1;
            hisstools_zip(&split, ptr.data(), i);
            for (long j = 0; j < (1L << i); j++)
                if (ptr[j] != (T) j) return false;
        }
        return true;
    }

    // direct DFT in double of the complex sequence (re, im), forward sign
    void dft(const std::vector<double> &re, const std::vector<double> &im, std::vector<double> &ore, std::vector<double> &oim)
    {
        const size_t n = re.size();
        ore.assign(n, 0.0);
        oim.assign(n, 0.0);
        for (size_t k = 0; k < n; k++)
            for (size_t j = 0; j < n; j++)
            {
                const double a = -2.0 * M_PI * (double) ((k * j) % n) / (double) n;
                ore[k] += re[j] * std::cos(a) - im[j] * std::sin(a);
                oim[k] += re[j] * std::sin(a) + im[j] * std::cos(a);
            }
    }

    template <class T> bool sweep(int min_log2, int max_log2, double &seconds)
    {
        typename Kit<T>::setup setup;
        hisstools_create_setup(&setup, max_log2);
        std::vector<T> re(size_t(1) << max_log2), im(size_t(1) << max_log2), r0, i0;
        typename Kit<T>::split split(re.data(), im.data());
        bool ok = true;
        const double t0 = now();
        for (int i = min_log2; i < max_log2 && ok; i++)
        {
            const size_t n = size_t(1) << i, half = n >> 1;
            for (size_t j = 0; j < n; j++) { re[j] = (T) noise(); im[j] = (T) noise(); }
            r0.assign(re.begin(), re.begin() + n);
            i0.assign(im.begin(), im.begin() + n);

            // complex: forward against the direct DFT (small sizes), then the inverse must give N * input
            hisstools_fft(setup, &split, i);
            double peak = 0.0, worst = 0.0;
            if (i <= 8)
            {
                std::vector<double> dr(r0.begin(), r0.end()), di(i0.begin(), i0.end()), wr, wi;
                dft(dr, di, wr, wi);
                for (size_t j = 0; j < n; j++)
                {
                    peak = std::fmax(peak, std::fmax(std::fabs(wr[j]), std::fabs(wi[j])));
                    worst = std::fmax(worst, std::fmax(std::fabs(wr[j] - re[j]), std::fabs(wi[j] - im[j])));
                }
                if (worst > Kit<T>::tol() * std::fmax(peak, 1.0)) { std::printf("fft mismatch at log2 %d: %g\n", i, worst / peak); ok = false; }
            }
            hisstools_ifft(setup, &split, i);
            worst = 0.0;
            for (size_t j = 0; j < n; j++)
                worst = std::fmax(worst, std::fmax(std::fabs(re[j] - r0[j] * (T) n), std::fabs(im[j] - i0[j] * (T) n)));
            if (worst > 8 * Kit<T>::tol() * (double) n) { std::printf("ifft(fft) mismatch at log2 %d: %g\n", i, worst / (double) n); ok = false; }

            // real, in place on the unzipped halves: rifft(rfft(x)) = 2N x; DC bin = 2 * sum(x)
            if (!half) continue;
            for (size_t j = 0; j < half; j++) { re[j] = r0[j]; im[j] = i0[j]; }
            double sum = 0.0;
            for (size_t j = 0; j < half; j++) sum += (double) r0[j] + (double) i0[j];
            hisstools_rfft(setup, &split, i);
            if (std::fabs((double) re[0] - 2.0 * sum) > 8 * Kit<T>::tol() * (double) n) { std::printf("rfft DC mismatch at log2 %d\n", i); ok = false; }
            hisstools_rifft(setup, &split, i);
            worst = 0.0;
            for (size_t j = 0; j < half; j++)
                worst = std::fmax(worst, std::fmax(std::fabs(re[j] - r0[j] * (T) (2 * n)), std::fabs(im[j] - i0[j] * (T) (2 * n))));
            if (worst > 16 * Kit<T>::tol() * (double) n) { std::printf("rifft(rfft) mismatch at log2 %d: %g\n", i, worst / (double) n); ok = false; }
        }
        seconds = now() - t0;
        hisstools_destroy_setup(setup);
        return ok;
    }

    // out-of-place overloads, including float samples into a double spectrum
    bool out_of_place()
    {
        const int log2n = 10;
        const size_t n = size_t(1) << log2n, half = n >> 1, in_length = n - 7;
        std::vector<float> xf(in_length), backf(n);
        std::vector<double> xd(in_length), backd(n), rd(half), id(half), rm(half), imx(half);
        std::vector<float> rf(half), jf(half);
        for (size_t j = 0; j < in_length; j++) { xf[j] = (float) noise(); xd[j] = (double) xf[j]; }
        FFT_SETUP_D sd; FFT_SETUP_F sf;
        hisstools_create_setup(&sd, log2n);
        hisstools_create_setup(&sf, log2n);
        FFT_SPLIT_COMPLEX_D spd(rd.data(), id.data()), spm(rm.data(), imx.data());
        FFT_SPLIT_COMPLEX_F spf(rf.data(), jf.data());
        hisstools_rfft(sd, xd.data(), &spd, in_length, log2n);
        hisstools_rfft(sd, xf.data(), &spm, in_length, log2n);           // float in, double out: same numbers as the double overload here
        hisstools_rfft(sf, xf.data(), &spf, in_length, log2n);
        bool ok = true;
        for (size_t j = 0; j < half; j++)
        {
            ok = ok && rd[j] == rm[j] && id[j] == imx[j];
            ok = ok && std::fabs(rd[j] - (double) rf[j]) < 1e-3 && std::fabs(id[j] - (double) jf[j]) < 1e-3;
        }
        hisstools_rifft(sd, &spd, backd.data(), log2n);
        hisstools_rifft(sf, &spf, backf.data(), log2n);
        for (size_t j = 0; j < n; j++)
        {
            const double want = j < in_length ? xd[j] * 2.0 * (double) n : 0.0;
            ok = ok && std::fabs(backd[j] - want) < 1e-9 && std::fabs((double) backf[j] - want) < 2e-2;
        }
        std::vector<double> zr(half), zi(half);
        FFT_SPLIT_COMPLEX_D z(zr.data(), zi.data());
        hisstools_unzip_zero(xf.data(), &z, in_length, log2n);
        for (size_t j = 0; j < half; j++)
        {
            ok = ok && zr[j] == (2 * j < in_length ? (double) xf[2 * j] : 0.0);
            ok = ok && zi[j] == (2 * j + 1 < in_length ? (double) xf[2 * j + 1] : 0.0);
        }
        hisstools_destroy_setup(sd);
        hisstools_destroy_setup(sf);
        return ok;
    }

    // IR_Manipulation_Tester-style: ir_phase in its eight modes on a 2^14-point spectrum, plus spike / delay / reverse identities
    template <class T> bool ir_functions()
    {
        const int log2n = 14;
        const uintptr_t n = uintptr_t(1) << log2n, half = n >> 1;
        typename Kit<T>::setup setup;
        hisstools_create_setup(&setup, log2n);
        std::vector<T> x(n), re(half), im(half), r2(half), i2(half), r3(half), i3(half);
        for (uintptr_t j = 0; j < n; j++) x[j] = (T) (noise() * std::exp(-(double) j / 500.0));
        typename Kit<T>::split spec(re.data(), im.data()), out(r2.data(), i2.data()), tmp(r3.data(), i3.data());
        hisstools_rfft(setup, x.data(), &spec, n, log2n);
        bool ok = true;
        const double tol = Kit<T>::tol() * 8;
        double peak = 0.0;
        for (uintptr_t j = 1; j < half; j++) peak = std::fmax(peak, std::hypot((double) re[j], (double) im[j]));
        const struct { double phase; bool zero; } modes[] = { { 0.1, true }, { 0.9, false }, { 0.0, true }, { 0.0, false }, { 1.0, true }, { 1.0, false }, { 0.5, true }, { 0.5, false } };
        for (const auto &m : modes)
        {
            ir_phase(setup, &out, &spec, n, m.phase, m.zero);
            for (uintptr_t j = 1; j < half && ok; j++)               // every phase setting keeps the magnitude response
            {
                const double a = std::hypot((double) re[j], (double) im[j]), b = std::hypot((double) r2[j], (double) i2[j]);
                if (std::fabs(a - b) > 40 * tol * peak) { std::printf("ir_phase(%g, %d) changed the magnitude at bin %lu\n", m.phase, (int) m.zero, (unsigned long) j); ok = false; }
            }
        }
        // delay by d then by -d is the identity; two time reversals are the identity; a spike delayed is a spike elsewhere
        // (a fractional delay keeps only the real part of the Nyquist bin, so that one slot is not recoverable)
        ir_delay(&out, &spec, n, 17.5);
        ir_delay(&tmp, &out, n, -17.5);
        bool part = std::fabs(r3[0] - re[0]) <= 8 * tol * peak;
        for (uintptr_t j = 1; j < half; j++) part = part && std::fabs(r3[j] - re[j]) <= 8 * tol * peak && std::fabs(i3[j] - im[j]) <= 8 * tol * peak;
        if (!part) std::printf("delay(+d) then delay(-d) is not the identity\n");
        ok = ok && part;
        ir_time_reverse(&out, &spec, n);
        ir_time_reverse(&tmp, &out, n);
        part = true;
        for (uintptr_t j = 0; j < half; j++) part = part && r3[j] == re[j] && i3[j] == im[j];
        if (!part) std::printf("two time reversals are not the identity\n");
        ok = ok && part;
        ir_spike(&out, n, 3.0);
        ir_delay(&tmp, &out, n, 4.0);
        ir_spike(&out, n, 7.0);
        part = true;
        for (uintptr_t j = 0; j < half; j++) part = part && std::fabs(r3[j] - r2[j]) <= 8 * tol && std::fabs(i3[j] - i2[j]) <= 8 * tol;
        if (!part) std::printf("a delayed spike is not the spike at the later position\n");
        ok = ok && part;
        ir_copy(&tmp, &spec, n);
        part = true;
        for (uintptr_t j = 0; j < half; j++) part = part && r3[j] == re[j] && i3[j] == im[j];
        if (!part) std::printf("ir_copy differs\n");
        ok = ok && part;
        // the IR products (SpectralFunctions.hpp:415-436): convolving with the spectrum of a unit spike at sample 0 (all ones; the packed
        // bin 0 = (1, 1)) changes nothing but the scale, exactly; correlating a spectrum with itself leaves no imaginary parts (bin 0's
        // Nyquist slot aside, which is a real product)
        ir_spike(&out, n, 0.0);
        ir_convolve_real(&tmp, &spec, &out, n, (T) 0.5);
        part = true;
        for (uintptr_t j = 0; j < half; j++) part = part && r3[j] == (T) 0.5 * re[j] && i3[j] == (T) 0.5 * im[j];
        if (!part) std::printf("ir_convolve_real with a unit spike is not the scaled input\n");
        ok = ok && part;
        ir_correlate_real(&tmp, &spec, &spec, n, (T) 1);
        part = r3[0] == re[0] * re[0] && i3[0] == im[0] * im[0];
        for (uintptr_t j = 1; j < half; j++) part = part && i3[j] == (T) 0 && r3[j] >= (T) 0;
        if (!part) std::printf("ir_correlate_real of a spectrum with itself is not its power spectrum\n");
        ok = ok && part;
        ir_convolve_complex(&tmp, &spec, &out, half, (T) 1);
        ir_correlate_complex(&out, &tmp, &out, half, (T) 1);
        part = true;
        for (uintptr_t j = 1; j < half; j++) part = part && r2[j] == re[j] && i2[j] == im[j];
        if (!part) std::printf("the complex products with a unit spectrum changed the input\n");
        ok = ok && part;
        // change_phase: the linear-phase version of x is symmetric about the centre of the frame
        spectral_processor<T> sp;
        std::vector<T> y(n);
        sp.change_phase(y.data(), x.data(), n, 0.5);
        double ypk = 0.0;
        for (uintptr_t j = 0; j < n; j++) ypk = std::fmax(ypk, std::fabs((double) y[j]));
        part = true;
        for (uintptr_t j = 1; j < n / 2; j++) part = part && std::fabs((double) y[n / 2 + j] - (double) y[n / 2 - j]) <= 200 * tol * ypk;
        if (!part) std::printf("change_phase(0.5) is not symmetric about the centre\n");
        ok = ok && part;
        if (!ok) std::printf("IR function check failed (%s)\n", Kit<T>::name());
        hisstools_destroy_setup(setup);
        return ok;
    }

    template <class T> bool run(int sweep_max)
    {
        std::printf("****** %s ******\n", Kit<T>::name());
        if (!zip_correctness<T>(1, 24)) { std::printf("zip error\n"); return false; }
        std::printf("FFT Zip Tests Successful\n");
        double s = 0.0;
        const bool ok = sweep<T>(0, sweep_max, s);
        std::printf("FFT Multiple Tests Elapsed %.2f s (log2 0..%d, host buffers)\n", s, sweep_max - 1);
        return ok;
    }
}

int main(int argc, char **argv)
{
    if (hcv_device_count() <= 0)
    {
        std::printf("no GPU: compile/link check only\n");
        return 2;
    }
    const int sweep_max = argc > 1 ? std::atoi(argv[1]) : 22;
    bool ok = run<double>(sweep_max);
    ok = run<float>(sweep_max) && ok;
    ok = out_of_place() && ok;
    ok = ir_functions<double>() && ok;
    ok = ir_functions<float>() && ok;
    std::printf(ok ? "Finished Running\n" : "Errors - did not complete tests\n");
    return ok ? 0 : 1;
}
