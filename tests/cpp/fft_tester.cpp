// An FFT_Tester-style programme ("- Test/FFT_Tester/FFT_Tester/main.cpp" in the reference) written against the drop-in
// header: it exercises exactly the calls that tester makes — zip/unzip integer round trips for log2 1..23, then fft, ifft,
// rfft and rifft at every size in a range, in double and float — and additionally checks the numbers (the reference's
// tester only times them): small sizes against a direct O(N^2) DFT in double, all sizes by the inverse round trip.
// Exit code: 0 pass, 1 numerical failure, 2 no GPU (compile / link check only).
#include "hisstools_amd/HISSTools_FFT.h"

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
    std::printf(ok ? "Finished Running\n" : "Errors - did not complete tests\n");
    return ok ? 0 : 1;
}
