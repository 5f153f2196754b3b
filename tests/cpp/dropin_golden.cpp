// The reference-generated golden vectors (tests/golden/golden_v1.npz, made by the unmodified reference: tests/golden/make_golden.py)
// driven through the C++ DROP-IN HEADERS exactly as a caller of the reference would drive its classes — same class names, namespaces and
// signatures (PartitionedConvolve.h:23-41, TimeDomainConvolve.h:15-31, MonoConvolve.h:30-48, NToMonoConvolve.h:18-24, Convolver.h:23-50 of
// the reference), the double overloads of Convolver and a std::vector<MonoConvolve> whose elements are moved (the reason the reference
// gives MonoConvolve move operations, MonoConvolve.cpp:49-78).
//
// argv[1] = a flat binary export of the .npz (tests/test_cpp_dropin.py writes it: u32 count, then per array u32 name length, name, u32
// ndim, u32 dims, float32 data).  Tolerances as in tests/test_gpu_parity.py: 2e-6 of the channel's peak, 1e-5 for many-input sums and
// the 2044-tap direct sum.  Exit codes: 0 ok, 2 no GPU (compile / link check only), 1 a vector missed its tolerance.
#include "hisstools_amd/Convolver.h"
#include "hisstools_amd/MonoConvolve.h"
#include "hisstools_amd/NToMonoConvolve.h"
#include "hisstools_amd/PartitionedConvolve.h"
#include "hisstools_amd/TimeDomainConvolve.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace
{
    struct Array { std::vector<uint32_t> dims; std::vector<float> v; };
    std::map<std::string, Array> G;
    const double TOL = 2e-6, TOL_SUM = 1e-5;
    int failures = 0;

    bool load(const char *path)
    {
        FILE *f = std::fopen(path, "rb");
        if (!f) return false;
        uint32_t count = 0;
        if (std::fread(&count, 4, 1, f) != 1) return false;
        for (uint32_t k = 0; k < count; k++)
        {
            uint32_t nl = 0, nd = 0;
            if (std::fread(&nl, 4, 1, f) != 1) return false;
            std::string name(nl, ' ');
            if (std::fread(&name[0], 1, nl, f) != nl || std::fread(&nd, 4, 1, f) != 1) return false;
            Array a;
            a.dims.resize(nd);
            size_t n = 1;
            for (uint32_t d = 0; d < nd; d++)
            {
                if (std::fread(&a.dims[d], 4, 1, f) != 1) return false;
                n *= a.dims[d];
            }
            a.v.resize(n);
            if (std::fread(a.v.data(), 4, n, f) != n) return false;
            G[name] = std::move(a);
        }
        std::fclose(f);
        return true;
    }

    const std::vector<float> &g(const std::string &k) { return G.at(k).v; }

    void check(const char *what, const float *y, const float *ref, size_t n, double tol)
    {
        double peak = 0.0, worst = 0.0;
        for (size_t i = 0; i < n; i++) peak = std::fmax(peak, std::fabs((double) ref[i]));
        for (size_t i = 0; i < n; i++) worst = std::fmax(worst, std::fabs((double) y[i] - (double) ref[i]));
        const double rel = worst / (peak > 0 ? peak : 1.0);
        std::printf("%-34s %zu samples, rel err %.3e (tol %.0e) %s\n", what, n, rel, tol, rel < tol ? "ok" : "FAILED");
        if (!(rel < tol)) failures++;
    }

    // the call sizes of a run: one size, or a list taken in turn
    struct Blocks
    {
        std::vector<size_t> sizes;
        size_t k = 0;
        size_t next(size_t left)
        {
            const size_t b = sizes[k++ % sizes.size()];
            return b < left ? b : left;
        }
    };

    template <class Obj> std::vector<float> run_bool(Obj &obj, const std::vector<float> &x, Blocks blocks)
    {
        std::vector<float> y(x.size(), 0.f);
        for (size_t pos = 0; pos < x.size();)
        {
            const size_t n = blocks.next(x.size() - pos);
            obj.process(x.data() + pos, y.data() + pos, n);
            pos += n;
        }
        return y;
    }

    std::vector<float> run_mono(HISSTools::MonoConvolve &m, const std::vector<float> &x, Blocks blocks)
    {
        // (the scratch block a host of the reference keeps in a MemorySwap<float> — MemorySwap.h:19-290, visible through Convolver.h — grown
        // on the control side, taken without waiting on the audio side: the drop-in header gives the same type)
        static MemorySwap<float> scratch(0);
        scratch.grow(x.size());
        std::vector<float> y(x.size(), 0.f);
        for (size_t pos = 0; pos < x.size();)
        {
            const size_t n = blocks.next(x.size() - pos);
            MemorySwap<float>::Ptr temp = scratch.attempt();
            if (temp.get() && temp.getSize() >= n) m.process(x.data() + pos, temp.get(), y.data() + pos, n);
            pos += n;
        }
        return y;
    }
}

int main(int argc, char **argv)
{
    if (hcv_device_count() <= 0)
    {
        std::printf("no GPU: link check only\n");
        return 2;
    }
    if (argc < 2 || !load(argv[1]))
    {
        std::printf("usage: dropin_golden <golden_v1.bin>\n");
        return 1;
    }

    // ---- PartitionedConvolve (PartitionedConvolve.h:23-41)
    {
        HISSTools::PartitionedConvolve p(256, g("part256_ir").size(), 0, 0);
        if (p.set(g("part256_ir").data(), g("part256_ir").size()) != CONVOLVE_ERR_NONE) return 1;
        check("PartitionedConvolve 256", run_bool(p, g("part256_x"), { { 512 } }).data(), g("part256_y").data(), g("part256_y").size(), TOL);
        HISSTools::PartitionedConvolve q(1024, g("part1024_ir").size(), 0, 0);
        if (q.set(g("part1024_ir").data(), g("part1024_ir").size()) != CONVOLVE_ERR_NONE) return 1;
        check("PartitionedConvolve 1024 ragged", run_bool(q, g("part1024_x"), { { 1, 7, 333, 2000, 64 } }).data(), g("part1024_y").data(), g("part1024_y").size(), TOL);
        HISSTools::PartitionedConvolve w(256, 1024, 300, 500);          // an IR window: offset 300, length 500
        if (w.set(g("partwin_ir").data(), g("partwin_ir").size()) != CONVOLVE_ERR_NONE) return 1;
        check("PartitionedConvolve window", run_bool(w, g("partwin_x"), { { 256 } }).data(), g("partwin_y").data(), g("partwin_y").size(), TOL);
    }
    // ---- TimeDomainConvolve (TimeDomainConvolve.h:15-31)
    for (int Lh : { 1, 16, 128, 2044 })
    {
        HISSTools::TimeDomainConvolve t(0, (uintptr_t) Lh);
        t.set(g("td_ir").data(), g("td_ir").size());
        const std::string key = "td" + std::to_string(Lh) + "_y";
        check(("TimeDomainConvolve " + std::to_string(Lh) + " taps").c_str(), run_bool(t, g("td_x"), { { 512 } }).data(), g(key).data(), g(key).size(),
              Lh <= 128 ? TOL : TOL_SUM);
    }
    // ---- MonoConvolve (MonoConvolve.h:30-48): the three latency modes, a custom layout, and elements of a std::vector that were moved
    for (int mode = 0; mode < 3; mode++)
    {
        HISSTools::MonoConvolve m(16384, static_cast<LatencyMode>(mode));
        if (m.set(g("mono_ir").data(), g("mono_ir").size(), true) != CONVOLVE_ERR_NONE) return 1;
        const std::string key = "mono" + std::to_string(mode) + "_y";
        check(("MonoConvolve latency mode " + std::to_string(mode)).c_str(), run_mono(m, g("mono_x"), { { 512 } }).data(), g(key).data(), g(key).size(), TOL);
    }
    {
        HISSTools::MonoConvolve m(11000, false, 512, 2048);
        if (m.set(g("mono_ir").data(), g("mono_ir").size(), false) != CONVOLVE_ERR_NONE) return 1;
        check("MonoConvolve custom (512, 2048)", run_mono(m, g("mono_x"), { { 100, 900, 2048 } }).data(), g("monoc_y").data(), g("monoc_y").size(), TOL);
        bool threw = false;
        try { HISSTools::MonoConvolve bad(11000, false, 2048, 512); } catch (const std::runtime_error &) { threw = true; }      // sizes out of order
        if (!threw) { std::printf("MonoConvolve: invalid sizes did not throw\n"); failures++; }
    }
    {
        std::vector<HISSTools::MonoConvolve> v;
        for (int k = 0; k < 5; k++) v.emplace_back(16384, static_cast<LatencyMode>(k % 3));       // (reallocations move the earlier elements)
        HISSTools::MonoConvolve moved(std::move(v[3]));                                           // mode 0, moved out of the vector
        v[1] = HISSTools::MonoConvolve(16384, kLatencyMedium);                                    // move assignment
        if (moved.set(g("mono_ir").data(), g("mono_ir").size(), true) != CONVOLVE_ERR_NONE) return 1;
        if (v[1].set(g("mono_ir").data(), g("mono_ir").size(), true) != CONVOLVE_ERR_NONE) return 1;
        check("MonoConvolve moved (mode 0)", run_mono(moved, g("mono_x"), { { 512 } }).data(), g("mono0_y").data(), g("mono0_y").size(), TOL);
        check("MonoConvolve move-assigned (mode 2)", run_mono(v[1], g("mono_x"), { { 512 } }).data(), g("mono2_y").data(), g("mono2_y").size(), TOL);
    }
    // ---- NToMonoConvolve 3 -> 1 (NToMonoConvolve.h:18-24)
    {
        const Array &irs = G.at("n2m_irs"), &x = G.at("n2m_x");
        const size_t L = irs.dims[1], S = x.dims[1];
        HISSTools::NToMonoConvolve c(3, 16384, kLatencyZero);
        for (uint32_t i = 0; i < 3; i++)
            if (c.set(i, irs.v.data() + i * L, L, true) != CONVOLVE_ERR_NONE) return 1;
        if (c.set(3, irs.v.data(), L, true) != CONVOLVE_ERR_IN_CHAN_OUT_OF_RANGE) { std::printf("NToMonoConvolve: range check\n"); failures++; }
        std::vector<float> y(S, 0.f), temp(S, 0.f);
        for (size_t pos = 0; pos < S; pos += 512)
        {
            const size_t n = S - pos < 512 ? S - pos : 512;
            const float *ins[3] = { x.v.data() + pos, x.v.data() + S + pos, x.v.data() + 2 * S + pos };
            c.process(ins, y.data() + pos, temp.data(), n, 3);
        }
        check("NToMonoConvolve 3 -> 1", y.data(), g("n2m_y").data(), S, TOL_SUM);
    }
    // ---- Convolver 2 x 3 (Convolver.h:23-50): float and double overloads; 3-channel parallel mode
    {
        const Array &irs = G.at("conv_irs"), &x = G.at("conv_x"), &yr = G.at("conv_y");
        const size_t L = irs.dims[2], S = x.dims[1];
        HISSTools::Convolver c(2, 3, kLatencyZero), d(2, 3, kLatencyZero);
        std::vector<double> ird(L);
        for (uint32_t o = 0; o < 3; o++)
            for (uint32_t i = 0; i < 2; i++)
            {
                const float *h = irs.v.data() + (o * 2 + i) * L;
                if (c.set(i, o, h, L, true) != CONVOLVE_ERR_NONE) return 1;
                for (size_t k = 0; k < L; k++) ird[k] = h[k];
                if (d.set(i, o, ird.data(), L, true) != CONVOLVE_ERR_NONE) return 1;
            }
        std::vector<float> y(3 * S, 0.f);
        std::vector<double> xd(x.v.begin(), x.v.end()), yd(3 * S, 0.0);
        for (size_t pos = 0; pos < S; pos += 512)
        {
            const size_t n = S - pos < 512 ? S - pos : 512;
            const float *ins[2] = { x.v.data() + pos, x.v.data() + S + pos };
            float *outs[3] = { y.data() + pos, y.data() + S + pos, y.data() + 2 * S + pos };
            c.process(ins, outs, 2, 3, n);
            const double *insd[2] = { xd.data() + pos, xd.data() + S + pos };
            double *outsd[3] = { yd.data() + pos, yd.data() + S + pos, yd.data() + 2 * S + pos };
            d.process(insd, outsd, 2, 3, n);
        }
        std::vector<float> ydf(yd.begin(), yd.end());
        for (uint32_t o = 0; o < 3; o++)
        {
            check(("Convolver 2 x 3, output " + std::to_string(o)).c_str(), y.data() + o * S, yr.v.data() + o * S, S, TOL_SUM);
            check(("Convolver 2 x 3 double API, output " + std::to_string(o)).c_str(), ydf.data() + o * S, yr.v.data() + o * S, S, TOL_SUM);
        }
    }
    {
        const Array &irs = G.at("par_irs"), &x = G.at("par_x"), &yr = G.at("par_y");
        const size_t L = irs.dims[1], S = x.dims[1];
        HISSTools::Convolver c(3, kLatencyShort);
        for (uint32_t o = 0; o < 3; o++)
            if (c.set(o, o, irs.v.data() + o * L, L, true) != CONVOLVE_ERR_NONE) return 1;
        std::vector<float> y(3 * S, 0.f);
        for (size_t pos = 0; pos < S; pos += 256)
        {
            const size_t n = S - pos < 256 ? S - pos : 256;
            const float *ins[3] = { x.v.data() + pos, x.v.data() + S + pos, x.v.data() + 2 * S + pos };
            float *outs[3] = { y.data() + pos, y.data() + S + pos, y.data() + 2 * S + pos };
            c.process(ins, outs, 3, 3, n);
        }
        for (uint32_t o = 0; o < 3; o++) check(("Convolver parallel, channel " + std::to_string(o)).c_str(), y.data() + o * S, yr.v.data() + o * S, S, TOL_SUM);
    }
    std::printf("%s\n", failures ? "FAILED" : "all golden vectors within tolerance");
    return failures ? 1 : 0;
}
