// The path's utility types as a host of the reference sees them through Convolver.h (MemorySwap.h:19-290, ThreadLocks.hpp:51-120,
// ConvolveSIMD.h:62-107): MemorySwap<T> + Ptr, thread_lock, lock_hold, FloatVector, ALIGNED_MALLOC — the drop-in headers must give the
// same behaviour.  Host-only: runs without a GPU.  Exit 0 = every check passed.
#include "hisstools_amd/Convolver.h"            // must bring MemorySwap / thread_lock with it, as the reference's does
#include "hisstools_amd/ConvolveSIMD.h"

#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } } while (0)

int main()
{
    // ---- MemorySwap: built-in allocation, logical sizes, grow / equal
    {
        MemorySwap<float> m(100);
        {
            auto p = m.access();
            CHECK(p.get() != nullptr && p.getSize() == 100);
            CHECK((reinterpret_cast<uintptr_t>(p.get()) & 15) == 0);             // aligned for vector ops
            p.get()[99] = 3.f;
            // while a Ptr lives, attempt() from elsewhere fails and yields an empty Ptr (MonoConvolve.cpp:181-183 relies on it)
            std::thread([&]() { auto q = m.attempt(); CHECK(q.get() == nullptr && q.getSize() == 0); }).join();
        }
        {
            auto q = m.attempt();
            CHECK(q.get() != nullptr && q.getSize() == 100 && q.get()[99] == 3.f);
        }
        float *before = m.access().get();
        CHECK(m.grow(50).get() == before);                                       // grow: only when larger
        CHECK(m.grow(200).getSize() == 200);
        CHECK(m.equal(200).getSize() == 200);
        CHECK(m.equal(10).getSize() == 10);                                      // equal: whenever different
        m.clear();
        {
            auto p = m.access();                                                  // (ONE Ptr at a time per thread: a second access() would wait for the first)
            CHECK(p.get() == nullptr && p.getSize() == 0);
        }
    }
    // ---- custom allocate / free functions are used and balanced; swap() takes memory it does not own
    {
        int allocs = 0, frees = 0;
        MemorySwap<double>::AllocFunc al = [&](uintptr_t n) { allocs++; return new double[n ? n : 1]; };
        MemorySwap<double>::FreeFunc fr = [&](double *p) { frees++; delete[] p; };
        {
            MemorySwap<double> m(al, fr, 8);
            CHECK(allocs == 1 && m.access().getSize() == 8);
            {
                auto p = m.equal(al, fr, 16);
                CHECK(allocs == 2 && frees == 1 && p.getSize() == 16);
                p.grow(al, fr, 4);                                               // on a held Ptr: nothing to do
                CHECK(allocs == 2);
                p.equal(al, fr, 4);
                CHECK(allocs == 3 && frees == 2 && p.getSize() == 4);
            }
            double mine[5] = { 0, 1, 2, 3, 4 };
            {
                auto p = m.swap(mine, 5);
                CHECK(frees == 3 && p.get() == mine && p.getSize() == 5);
            }
            MemorySwap<double> moved(std::move(m));
            CHECK(moved.access().get() == mine);
            CHECK(m.access().get() == nullptr);
            moved.clear();                                                        // the caller's memory is never freed
            CHECK(frees == 3);
        }
        CHECK(allocs == 3 && frees == 3);
    }
    // ---- std::vector<MemorySwap<T>> (movable): what MonoConvolve keeps per object
    {
        std::vector<MemorySwap<float>> v;
        for (int k = 0; k < 5; k++) v.emplace_back((uintptr_t) (16 + k));
        for (int k = 0; k < 5; k++) CHECK(v[(size_t) k].access().getSize() == (uintptr_t) (16 + k));
    }
    // ---- thread_lock: mutual exclusion; attempt never waits; lock_hold releases once
    {
        thread_lock lock;
        long counter = 0;
        std::vector<std::thread> ts;
        for (int t = 0; t < 4; t++)
            ts.emplace_back([&]() { for (int k = 0; k < 20000; k++) { lock.acquire(); counter++; lock.release(); } });
        for (auto &t : ts) t.join();
        CHECK(counter == 80000);
        CHECK(lock.attempt());
        CHECK(!lock.attempt());
        lock.release();
        {
            lock_hold<thread_lock, &thread_lock::acquire, &thread_lock::release> h(&lock);
            CHECK(!lock.attempt());
            h.release();
            CHECK(lock.attempt());
            lock.release();
            h.release();                                                          // a second release is nothing
        }
        CHECK(lock.attempt());
        lock.release();
    }
    // ---- FloatVector / SIMDVector / ALIGNED_MALLOC
    {
        static_assert(FloatVector::size == 4, "four floats wide, as the reference's");
        float a[5] = { 1, 2, 3, 4, 5 }, b[4] = { 10, 20, 30, 40 }, out[4];
        FloatVector x = FloatVector::unaligned_load(a + 1), y = FloatVector::unaligned_load(b);
        FloatVector z = x * y + FloatVector(1.f) - FloatVector(0.5f);
        z += FloatVector(0.5f);
        z.unaligned_store(out);
        CHECK(out[0] == 21.f && out[1] == 61.f && out[2] == 121.f && out[3] == 201.f);
        CHECK(z.sum() == 404.f);
        float *p = static_cast<float *>(ALIGNED_MALLOC(100 * sizeof(float)));
        CHECK(p && (reinterpret_cast<uintptr_t>(p) & 15) == 0);
        ALIGNED_FREE(p);
    }
    std::printf(failures ? "%d checks failed\n" : "utility types ok\n", failures);
    return failures ? 1 : 0;
}
