// The audio-thread contract, as a host would meet it: HISSTools::Convolver (the drop-in header), an audio thread making paced
// host-pointer process() calls — SCHED_FIFO where the process may raise it — and a control thread (SCHED_OTHER) looping
// set(..., resize = true) with growing impulse responses of up to 10 s on rows the audio thread keeps playing.
//
// The reference's contract (MemorySwap.h:182-185 `attempt`, MonoConvolve.cpp:181-183, ThreadLocks.hpp:51-87): process never waits
// for a control call.  Here the engine's state has an owner instead of a lock (hcv_engine.h: Engine::mOwner): the audio thread makes
// ONE compare-exchange per call; while a stream is running control calls post their swap section and the audio thread runs it between
// two of its blocks.  A control thread that is preempted — HCV_TEST_CTL_STALL_US makes every control call sleep that long right
// where it would otherwise be holding something — therefore cannot hold the audio thread up: that is what this programme measures.
//
//   audio_contract <block> <calls> [stall_us] [ir_seconds_max]
//
// exit 0: every criterion met; 1: a criterion failed; 2: no GPU (compile / link check only).
// Criteria (printed one by one): no call over its budget (block / 48 kHz), the slowest call inside the budget, start_collisions == 0,
// every set() ran as a section on the audio thread (mailbox_runs >= sets), the rows that are never replaced equal (1e-5 of the
// peak) to a second convolver that saw no control call at all, >= 700 set(resize) calls at 32
// samples per call (>= 100 at 128).  The wall-clock criteria are asserted when the host is quiet (load average below 8 for a SCHED_FIFO
// audio thread, below 2 for a time-sharing one, or AUDIO_CONTRACT_STRICT=1) and reported otherwise: a pre-empted audio thread is not the
// library's doing.  The slowest calls are printed with their positions in the stream.
#include "hisstools_amd/Convolver.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>

namespace
{
    using clk = std::chrono::steady_clock;
    double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

    std::vector<float> decaying_noise(size_t n, unsigned seed)
    {
        std::mt19937 g(seed);
        std::vector<float> h(n);
        double norm = 0.0;
        for (size_t k = 0; k < n; k++)
        {
            const double u = (double) (g() >> 8) * (1.0 / 16777216.0);
            const double v = (2.0 * u - 1.0) * std::pow(10.0, -3.0 * (double) k / (double) n);
            h[k] = (float) v;
            norm += v * v;
        }
        const float s = (float) (1.0 / std::sqrt(norm));
        for (float &v : h) v *= s;
        return h;
    }
}

int main(int argc, char **argv)
{
    if (hcv_device_count() <= 0)
    {
        std::printf("no GPU: compile / link check only\n");
        return 2;
    }
    const size_t RB = argc > 1 ? (size_t) std::atoi(argv[1]) : 128;
    const size_t ncalls = argc > 2 ? (size_t) std::atoi(argv[2]) : 1400;
    const int stall_us = argc > 3 ? std::atoi(argv[3]) : 300;
    const double ir_seconds = argc > 4 ? std::atof(argv[4]) : 10.0;
    const uint32_t nin = 16, nout = 16;
    const double fs = 48000.0;
    const double budget = 1e3 * (double) RB / fs;
    if (stall_us > 0)
    {
        char buf[32];
        std::snprintf(buf, sizeof buf, "%d", stall_us);
        setenv("HCV_TEST_CTL_STALL_US", buf, 1);          // (read once by the library, at the first control call)
    }

    // The objects are created for 1.25 s impulse responses and the control thread grows them to `ir_seconds`: a host that does that reserves
    // the control path's memory first, or the regrows have the driver map a gigabyte under the audio thread (hisstools_amd.h: hcv_ctl_reserve)
    hcv_ctl_reserve(hcv_get_default_device() >= 0 ? hcv_get_default_device() : 0, (size_t) 4 << 30);
    // two convolvers with the same sixteen-by-sixteen matrix: `c` meets the control thread, `q` never does
    HISSTools::Convolver c(nin, nout, kLatencyZero), q(nin, nout, kLatencyZero);
    const size_t L_fix = 60000;
    for (uint32_t o = 0; o < nout; o++)
        for (uint32_t i = 0; i < nin; i++)
        {
            const std::vector<float> h = decaying_noise(L_fix, 1000 * i + o + 1);
            if (c.set(i, o, h.data(), h.size(), true) != CONVOLVE_ERR_NONE || q.set(i, o, h.data(), h.size(), true) != CONVOLVE_ERR_NONE)
            {
                std::printf("set failed: %s\n", hcv_last_error());
                return 1;
            }
        }
    std::vector<std::vector<float>> grow;
    for (int k = 1; k <= 8; k++) grow.push_back(decaying_noise((size_t) (ir_seconds * fs * k / 8.0), 77 + k));

    const size_t S = ncalls * RB;
    std::vector<std::vector<float>> x(nin, std::vector<float>(S)), y(nout, std::vector<float>(S)), yq(nout, std::vector<float>(S));
    {
        std::mt19937 g(777);
        for (auto &row : x)
            for (float &v : row) v = (float) ((double) (g() >> 8) * (2.0 / 16777216.0) - 1.0);
    }
    std::vector<const float *> ip(nin);
    std::vector<float *> op(nout);
    auto call = [&](HISSTools::Convolver &cv, std::vector<std::vector<float>> &out, size_t k)
    {
        for (uint32_t i = 0; i < nin; i++) ip[i] = x[i].data() + k * RB;
        for (uint32_t o = 0; o < nout; o++) op[o] = out[o].data() + k * RB;
        cv.process(ip.data(), op.data(), nin, nout, RB);
    };
    // the quiet convolver first (unpaced), then settle `c`, restart it and clear its counters
    for (size_t k = 0; k < ncalls; k++) call(q, yq, k);
    for (size_t k = 0; k < 40; k++) call(c, y, k);
    c.reset();
    hcv_convolver_clear_stats(c.handle());

    std::atomic<bool> stop { false };
    std::atomic<int> sets { 0 }, set_errors { 0 };
    double worst_set_ms = 0.0;
    std::thread control([&]()
    {
        int k = 0;
        while (!stop.load(std::memory_order_acquire))
        {
            const std::vector<float> &h = grow[(size_t) k % grow.size()];
            const auto t0 = clk::now();
            // rows 8 .. 15 only: rows 0 .. 7 keep their impulse responses and are compared with the quiet convolver
            if (c.set((uint32_t) (5 * k) % nin, 8 + (uint32_t) k % 8, h.data(), h.size(), true) != CONVOLVE_ERR_NONE) set_errors++;
            worst_set_ms = std::max(worst_set_ms, ms_since(t0));
            sets++;
            k++;
            sched_yield();
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
    });

    bool fifo = false;
    {
        sched_param sp {};
        sp.sched_priority = 10;
        fifo = pthread_setschedparam(pthread_self(), SCHED_FIFO, &sp) == 0;
    }
    std::vector<double> ts(ncalls);
    const auto t_start = clk::now();
    for (size_t k = 0; k < ncalls; k++)
    {
        const auto due = t_start + std::chrono::duration_cast<clk::duration>(std::chrono::duration<double>((double) (k * RB) / fs));
        while (clk::now() < due) {}
        const auto t0 = clk::now();
        call(c, y, k);
        ts[k] = ms_since(t0);
    }
    // (the counters as the stream ends: a set() still in flight then waits the streaming window out and serves itself, rightly)
    hcv_rt_stats rt;
    hcv_convolver_rt_stats(c.handle(), &rt);
    const int sets_seen = sets.load();
    {
        sched_param sp {};
        (void) pthread_setschedparam(pthread_self(), SCHED_OTHER, &sp);
    }
    stop.store(true, std::memory_order_release);
    control.join();
    std::vector<double> sorted(ts);
    std::sort(sorted.begin(), sorted.end());
    const size_t over = (size_t) std::count_if(ts.begin(), ts.end(), [&](double t) { return t > budget; });
    double load[1] = { 0.0 };
    (void) getloadavg(load, 1);
    // (wall clock: asserted on a quiet host — for a SCHED_FIFO audio thread a load average below 8, for one the container left in the
    // time-sharing class below 2: such a thread has no claim on a CPU inside 0.667 ms on a host that runs anything else — and with an
    // allowance of one call in 2000 for a time-sharing thread even then; AUDIO_CONTRACT_STRICT=1 asserts regardless)
    const bool strict = load[0] < (fifo ? 8.0 : 2.0) || (std::getenv("AUDIO_CONTRACT_STRICT") && std::atoi(std::getenv("AUDIO_CONTRACT_STRICT")));
    const size_t allowance = fifo ? 0 : (ncalls + 1999) / 2000;
    std::printf("%zu-sample calls (budget %.3f ms), %zu of them, audio thread %s, control thread stalled %d us per call, load average %.1f\n", RB, budget, ncalls,
                fifo ? "SCHED_FIFO" : "SCHED_OTHER (SCHED_FIFO refused)", stall_us, load[0]);
    std::printf("  %d set(resize) calls beside them (worst %.1f ms each), %d errors\n", sets.load(), worst_set_ms, set_errors.load());
    std::printf("  calls: p50 %.4f  p99 %.4f  max %.4f ms, %zu over budget\n", sorted[ncalls / 2], sorted[(size_t) (0.99 * (double) ncalls)], sorted.back(), over);
    {
        // (which calls were the slowest: a one-off at the start of the stream reads differently from a spread over it)
        std::vector<size_t> idx(ncalls);
        for (size_t k = 0; k < ncalls; k++) idx[k] = k;
        std::partial_sort(idx.begin(), idx.begin() + std::min<size_t>(4, ncalls), idx.end(), [&](size_t a, size_t b) { return ts[a] > ts[b]; });
        std::printf("  slowest calls:");
        for (size_t k = 0; k < std::min<size_t>(4, ncalls); k++) std::printf("  #%zu %.4f ms", idx[k], ts[idx[k]]);
        std::printf("\n");
    }
    std::printf("  engine: start_collisions %llu, sections run by the audio thread %llu (longest %.1f us, mean %.1f us), by control threads %llu; arena misses %llu\n",
                (unsigned long long) rt.start_collisions, (unsigned long long) rt.mailbox_runs, (double) rt.mailbox_ns_max / 1e3,
                rt.mailbox_runs ? (double) rt.mailbox_ns_total / 1e3 / (double) rt.mailbox_runs : 0.0, (unsigned long long) rt.ctl_sections,
                (unsigned long long) rt.arena_misses);

    int failed = 0;
    auto criterion = [&](bool ok, bool asserted, const char *what)
    {
        std::printf("  [%s] %s\n", ok ? "ok" : (asserted ? "FAILED" : "not met, not asserted on a loaded host"), what);
        if (!ok && asserted) failed++;
    };
    criterion(rt.start_collisions == 0, true, "no process call ever found the engine owned by a control thread (start_collisions == 0)");
    criterion(set_errors.load() == 0, true, "every set() succeeded");
    criterion((int) rt.mailbox_runs >= sets_seen - 1, true, "every set()'s swap section ran on the audio thread, between two of its blocks");
    criterion(rt.ctl_sections == 0, true, "no control thread owned the engine while the stream ran");
    criterion(rt.arena_misses == 0, true, "every regrow was served by the control arena reserved for it (no driver mapping under the stream)");
    // (a control thread stalled for milliseconds per call makes fewer of them)
    criterion(sets_seen >= (RB <= 32 ? (stall_us <= 500 ? 700 : 250) : 100) || ncalls < 1400, true, "enough set(resize) calls met the stream");
    criterion(over <= allowance, strict, fifo ? "no call over its budget" : "no call over its budget (a time-sharing audio thread: one in 2000 allowed)");
    criterion(sorted[ncalls - 1 - std::min(allowance, ncalls - 1)] < budget, strict, fifo ? "the slowest call inside the budget" : "the slowest call outside that allowance inside the budget");
    // rows 0 .. 7: never replaced.  `c` was restarted (reset) before the paced run and fed the same samples as `q` from its start:
    // the same kernels on the same samples
    double worst = 0.0, peak = 0.0;
    for (uint32_t o = 0; o < 8; o++)
        for (size_t n = 0; n < S; n++)
        {
            worst = std::max(worst, (double) std::fabs(y[o][n] - yq[o][n]));
            peak = std::max(peak, (double) std::fabs(yq[o][n]));
        }
    std::printf("  untouched rows against the convolver that met no control call: max |difference| %.3e of peak %.3e\n", worst, peak);
    criterion(worst <= 1e-5 * peak, true, "the rows whose pairs are never replaced play on unchanged right through the swaps and regrows");
    bool finite = true;
    for (uint32_t o = 8; o < nout; o++)
        for (size_t n = 0; n < S; n++) finite = finite && std::isfinite(y[o][n]);
    criterion(finite, true, "the replaced rows stay finite");
    return failed ? 1 : 0;
}
