// Drop-in for ThreadLocks.hpp:51-120 of the reference (included by MemorySwap.h, and through it visible to every caller of
// Convolver.h): `thread_lock` — acquire / attempt / release, non-copyable — and the RAII holder template `lock_hold`.
//
// An implementation of its own: one atomic word taken by exchange; `acquire` spins briefly with a CPU pause hint, then backs off with
// yields and short sleeps (a waiter never burns a core for more than a few microseconds).  `attempt` is one exchange and never waits —
// what an audio thread calls (the reference's MonoConvolve::process, MonoConvolve.cpp:181-183).  Host-only: nothing here touches the GPU;
// the library itself no longer has a lock on its process path at all (hisstools_amd.h: hcv_rt_stats).
#pragma once

#include <atomic>
#include <chrono>
#include <thread>

class thread_lock
{
public:

    thread_lock() {}
    // (as the reference's: the destructor takes the lock, so that nobody is inside a section the object guards when it goes)
    ~thread_lock() { acquire(); }

    thread_lock(const thread_lock&) = delete;
    thread_lock& operator=(const thread_lock&) = delete;

    void acquire()
    {
        for (int spins = 0; ; spins++)
        {
            if (attempt()) return;
            if (spins < 64)
                pause();
            else if (spins < 256)
                std::this_thread::yield();
            else
                std::this_thread::sleep_for(std::chrono::microseconds(spins < 1024 ? 1 : 50));
        }
    }

    bool attempt() { return !mTaken.exchange(true, std::memory_order_acquire); }
    void release() { mTaken.store(false, std::memory_order_release); }

private:

    static void pause()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#elif defined(__aarch64__)
        __asm__ __volatile__("yield" ::: "memory");
#endif
    }

    std::atomic<bool> mTaken { false };
};

// A lock holder by RAII: takes the lock through `acquire_method` when given one, gives it back through `release_method` when it goes
// or when release() is called (once).
template <class T, void (T::*acquire_method)(), void (T::*release_method)()>
class lock_hold
{
public:

    lock_hold() : mLock(nullptr) {}
    lock_hold(T *lock) : mLock(lock) { if (mLock) (mLock->*acquire_method)(); }
    ~lock_hold() { release(); }

    lock_hold(const lock_hold&) = delete;
    lock_hold& operator=(const lock_hold&) = delete;

    void release()
    {
        if (mLock)
        {
            (mLock->*release_method)();
            mLock = nullptr;
        }
    }

private:

    T *mLock;
};
