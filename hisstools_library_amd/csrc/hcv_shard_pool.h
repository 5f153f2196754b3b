// Enqueue threads of a sharded Convolver (hcv_api.hip): one persistent host thread per shard after the first, so that the HIP
// calls of one process() block — a dozen launches, records and waits per engine, 15-100 us of host time each (DESIGN section 8's
// price list) — are issued side by side on the shards' devices instead of one shard after the other from the caller's thread.
//
// The caller's (audio) thread posts a job by bumping each worker's sequence number, runs shard 0 itself and then waits until every
// worker has reported: a bounded spin, then yields (it never sleeps and takes no lock).  A worker spins after its last job for about
// one and a half times the interval between the last two jobs (at least 100 us, at most 3 ms) — the next callback of a paced stream
// finds it awake, without a futex wake and the scheduler's latency per shard — and then sleeps on a futex; the Dekker pair (post:
// store seq, load asleep | worker: store asleep, futex re-checks seq in the kernel) makes a lost wake-up impossible.  Workers take
// the poster's scheduling class and priority when they see them change (a SCHED_FIFO audio thread would otherwise starve the very
// threads it waits for; where the process may not raise them, the yields above are what is left) and its identity as "the audio
// thread" of their engines (hcv_engine.h: set_thread_audio_identity).
#pragma once

#include "hcv_engine.h"

#include <atomic>
#include <chrono>
#include <climits>
#include <cstdint>
#include <memory>
#include <thread>
#include <vector>

#include <linux/futex.h>
#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace hcv
{
    class ShardPool
    {
    public:
        using Fn = bool (*)(void *ctx, int index);

        // workers for indices 1 .. n-1 (index 0 runs on the calling thread); devices[k] becomes worker k's current HIP device
        ShardPool(int n, const int *devices) : mN(n)
        {
            for (int k = 1; k < n; k++)
            {
                mWorkers.emplace_back(new Worker());
                Worker &w = *mWorkers.back();
                w.device = devices[k];
                w.th = std::thread([this, &w, k]() { loop(w, k); });
            }
        }

        ~ShardPool()
        {
            mQuit.store(true);
            for (auto &w : mWorkers)
            {
                w->seq.fetch_add(1);
                wake(*w);
                if (w->th.joinable()) w->th.join();
            }
        }

        ShardPool(const ShardPool &) = delete;
        ShardPool &operator=(const ShardPool &) = delete;

        // fn(ctx, k) for every k in [0, n) whose bit is set in `mask`: k = 0 on this thread, the others on their workers.  Returns
        // when all have finished; false if any of them returned false.
        bool run(Fn fn, void *ctx, uint64_t mask)
        {
            mFn = fn;
            mCtx = ctx;
            mPoster = current_thread_identity();
            {
                // (two syscall-free reads on glibc; published with the jobs' sequence numbers)
                int policy = SCHED_OTHER;
                sched_param sp {};
                if (pthread_getschedparam(pthread_self(), &policy, &sp) == 0)
                {
                    mPolicy = policy;
                    mPriority = sp.sched_priority;
                }
            }
            uint32_t target[64];
            for (int k = 1; k < mN; k++)
            {
                if (!(mask >> k & 1)) continue;
                Worker &w = *mWorkers[(size_t) k - 1];
                target[k] = w.seq.load(std::memory_order_relaxed) + 1;
                w.seq.store(target[k], std::memory_order_seq_cst);
                if (w.asleep.load(std::memory_order_seq_cst)) wake(w);
            }
            bool ok = (mask & 1) ? fn(ctx, 0) : true;
            for (int k = 1; k < mN; k++)
            {
                if (!(mask >> k & 1)) continue;
                Worker &w = *mWorkers[(size_t) k - 1];
                int spins = 0;
                while (w.done.load(std::memory_order_acquire) != target[k])
                {
                    cpu_relax();
                    if (++spins > 20000) sched_yield();     // (~100 us of spinning: from here on the core is offered to whoever is runnable)
                }
                ok = ok && w.ok;
            }
            return ok;
        }

        int size() const { return mN; }

    private:
        struct alignas(128) Worker
        {
            std::atomic<uint32_t> seq { 0 };        // jobs posted
            std::atomic<uint32_t> asleep { 0 };
            alignas(64) std::atomic<uint32_t> done { 0 };   // jobs finished (a line of its own: the poster spins on it)
            bool ok = true;
            int device = 0;
            std::thread th;
        };

        static void wake(Worker &w) { (void) syscall(SYS_futex, reinterpret_cast<uint32_t *>(&w.seq), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0); }

        void loop(Worker &w, int index)
        {
            (void) hipSetDevice(w.device);
            uint32_t seen = 0;
            int policy = SCHED_OTHER, priority = 0;
            auto last_job = std::chrono::steady_clock::now();
            std::chrono::nanoseconds window(100000);
            for (;;)
            {
                // spin, then sleep
                const auto t0 = std::chrono::steady_clock::now();
                uint32_t s;
                int polls = 0;
                while ((s = w.seq.load(std::memory_order_acquire)) == seen)
                {
                    cpu_relax();
                    if ((++polls & 255) == 0 && std::chrono::steady_clock::now() - t0 > window)
                    {
                        w.asleep.store(1, std::memory_order_seq_cst);
                        while ((s = w.seq.load(std::memory_order_seq_cst)) == seen)
                            (void) syscall(SYS_futex, reinterpret_cast<uint32_t *>(&w.seq), FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0);
                        w.asleep.store(0, std::memory_order_seq_cst);
                        break;
                    }
                }
                if (mQuit.load()) return;
                seen = s;
                {
                    // the idle spin covers the stream's period (1.5 x the last interval, 100 us .. 3 ms)
                    const auto now = std::chrono::steady_clock::now();
                    const auto gap = std::chrono::duration_cast<std::chrono::nanoseconds>(now - last_job);
                    last_job = now;
                    window = std::min(std::chrono::nanoseconds(3000000), std::max(std::chrono::nanoseconds(100000), gap + gap / 2));
                }
                if (mPolicy != policy || mPriority != priority)
                {
                    policy = mPolicy;
                    priority = mPriority;
                    sched_param sp {};
                    sp.sched_priority = priority;
                    (void) pthread_setschedparam(pthread_self(), policy, &sp);      // (refused without the privilege: the poster's yields remain)
                }
                set_thread_audio_identity(mPoster);
                w.ok = mFn(mCtx, index);
                w.done.store(s, std::memory_order_release);
            }
        }

        int mN = 0;
        Fn mFn = nullptr;
        void *mCtx = nullptr;
        size_t mPoster = 0;                 // (plain fields: published by the release store of each worker's sequence number)
        int mPolicy = SCHED_OTHER, mPriority = 0;
        std::atomic<bool> mQuit { false };
        std::vector<std::unique_ptr<Worker>> mWorkers;
    };
}
