// HCV_ORDER_CHECK=1 — the stream / event order of DESIGN.md section 2, asserted while the engine enqueues (a debug aid: off, it costs one load
// of a flag per hooked call).
//
// The engine's correctness on the device rests on an order the host states in three ways: program order on a stream, event records and
// waits between streams, and — for the rings — depth (a slot is written again only after everything that read it is known to be behind
// some event the writer has waited for).  None of that is visible in the results until a box is fast or slow enough to take the other
// order.  This checker keeps, per engine, a vector clock for every stream and event as the calls are made (hipEventRecord,
// hipStreamWaitEvent, the host's synchronize calls: hooked below for the engine's translation units) and a short history of the buffer
// accesses the enqueue code declares next to its launches (ORD_ACCESS: input-spectrum ring slots, partial-sum buffers, timeline spans,
// the history ring).  A read must come after every earlier write of what it reads, a write after every earlier access — in the sense of
// those clocks.  Anything else is reported on stderr and counted (hcv_order_check_violations()); HCV_ORDER_CHECK=2 aborts at the first.
//
// What it does not see: hand-overs inside a launch (the fused blocks' counters; the n x m block's two launches are joined by ORD_MEET where
// the counters join them — the other direction of that pair, a forward launch that arrives blocks late, is settled on the device by
// fwd_publish_kernel's look at `progress` and is outside this model), other engines' streams, and knowledge one host thread gains by
// synchronizing (taken as everybody's).
#pragma once

#include <hip/hip_runtime.h>

#include <array>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace hcv
{

class OrderCheck
{
public:
    static constexpr int kStreams = 32;
    using Clock = std::array<uint32_t, kStreams>;

    void add_stream(hipStream_t s, const char *name)
    {
        if (mIds.count(s) || mNames.size() >= (size_t) kStreams) return;
        mIds[s] = (int) mNames.size();
        mNames.push_back(name);
        mClk.emplace_back();
        mClk.back().fill(0);
    }
    bool knows(hipStream_t s) const { return mIds.count(s) != 0; }

    void record(hipEvent_t e, hipStream_t s)
    {
        const int i = id(s);
        if (i < 0) return;
        absorb_host(i);
        mEv[e] = mClk[i];
    }
    void wait(hipStream_t s, hipEvent_t e)
    {
        const int i = id(s);
        auto it = mEv.find(e);
        if (i < 0 || it == mEv.end()) return;           // (an event of another engine, or one never recorded: nothing to learn)
        join(mClk[i], it->second);
    }
    // `to` goes on behind everything enqueued on `from` so far by a hand-over the host does not see (in-launch counters)
    void meet(hipStream_t from, hipStream_t to)
    {
        const int a = id(from), b = id(to);
        if (a >= 0 && b >= 0) join(mClk[b], mClk[a]);
    }
    void host_sync_stream(hipStream_t s)
    {
        const int i = id(s);
        if (i >= 0) join(mHost, mClk[i]);
    }
    void host_sync_event(hipEvent_t e)
    {
        auto it = mEv.find(e);
        if (it != mEv.end()) join(mHost, it->second);
    }
    void host_sync_all()
    {
        for (const Clock &c : mClk) join(mHost, c);
    }

    // elements [lo, hi) of `buf`, positions taken modulo `ring` when ring > 0
    int access(hipStream_t s, const void *buf, long long lo, long long hi, long long ring, bool write, const char *what)
    {
        const int i = id(s);
        if (i < 0 || hi <= lo) return 0;
        absorb_host(i);
        const uint32_t t = ++mClk[i][i];
        int bad = 0;
        if (ring > 0 && hi - lo >= ring)
        {
            lo = 0;
            hi = ring;
        }
        else if (ring > 0)
        {
            const long long len = hi - lo;
            lo = ((lo % ring) + ring) % ring;
            hi = lo + len;
        }
        std::deque<Acc> &h = mHist[buf];
        for (const Acc &a : h)
        {
            if (!(write || a.write)) continue;
            if (!overlap(lo, hi, a.lo, a.hi, ring)) continue;
            if (mClk[i][a.stream] >= a.t) continue;     // this stream knows of it: ordered
            bad++;
            report(what, write, s, lo, hi, a);
        }
        h.push_back(Acc{lo, hi, a_stream(i), t, write, what});
        if (h.size() > 192) h.pop_front();
        return bad;
    }

private:
    struct Acc
    {
        long long lo, hi;
        int stream;
        uint32_t t;
        bool write;
        const char *what;
    };
    static int a_stream(int i) { return i; }
    static bool overlap(long long lo, long long hi, long long alo, long long ahi, long long ring)
    {
        if (ring <= 0) return lo < ahi && alo < hi;
        // (both start inside [0, ring) and may run past its end)
        for (long long sh = -ring; sh <= ring; sh += ring)
            if (lo < ahi + sh && alo + sh < hi) return true;
        return false;
    }
    int id(hipStream_t s) const
    {
        auto it = mIds.find(s);
        return it == mIds.end() ? -1 : it->second;
    }
    static void join(Clock &a, const Clock &b)
    {
        for (int k = 0; k < kStreams; k++) a[k] = a[k] < b[k] ? b[k] : a[k];
    }
    void absorb_host(int i) { join(mClk[i], mHost); }
    void report(const char *what, bool write, hipStream_t s, long long lo, long long hi, const Acc &a);

    std::unordered_map<hipStream_t, int> mIds;
    std::vector<const char *> mNames;
    std::vector<Clock> mClk;
    Clock mHost = {};
    std::unordered_map<hipEvent_t, Clock> mEv;
    std::unordered_map<const void *, std::deque<Acc>> mHist;
};

// ---- the process-wide side: which engine a stream belongs to (streams are never shared between engines)
struct OrderRegistry
{
    std::atomic<int> mode{-1};                          // -1 not read yet, 0 off, 1 report, 2 abort
    std::atomic<long long> violations{0};
    std::mutex mu;
    std::unordered_map<hipStream_t, OrderCheck *> by_stream;
};
inline OrderRegistry &order_registry()
{
    static OrderRegistry r;
    return r;
}
inline int order_mode()
{
    OrderRegistry &r = order_registry();
    int m = r.mode.load(std::memory_order_relaxed);
    if (m < 0)
    {
        const char *e = std::getenv("HCV_ORDER_CHECK");
        m = e ? std::atoi(e) : 0;
        if (m < 0) m = 0;
        r.mode.store(m, std::memory_order_relaxed);
    }
    return m;
}
inline void OrderCheck::report(const char *what, bool write, hipStream_t s, long long lo, long long hi, const Acc &a)
{
    OrderRegistry &r = order_registry();
    const long long n = r.violations.fetch_add(1) + 1;
    if (n <= 24)
        std::fprintf(stderr, "[hcv] order check: %s %s [%lld, %lld) on stream '%s' is not ordered behind the %s %s [%lld, %lld) on stream '%s'\n",
                     write ? "write" : "read", what, lo, hi, mNames[(size_t) id(s)], a.write ? "write" : "read", a.what, a.lo, a.hi, mNames[(size_t) a.stream]);
    if (order_mode() >= 2) std::abort();
}

// (one lock for all engines: a debug mode)
template <class F>
inline void with_order_check(hipStream_t s, F &&f)
{
    if (order_mode() <= 0) return;
    OrderRegistry &r = order_registry();
    std::lock_guard<std::mutex> g(r.mu);
    auto it = r.by_stream.find(s);
    if (it != r.by_stream.end()) f(*it->second);
}
inline void order_register(OrderCheck *oc, hipStream_t s, const char *name)
{
    if (order_mode() <= 0 || !s) return;
    OrderRegistry &r = order_registry();
    std::lock_guard<std::mutex> g(r.mu);
    oc->add_stream(s, name);
    r.by_stream[s] = oc;
}
inline void order_unregister(OrderCheck *oc)
{
    if (order_mode() <= 0) return;
    OrderRegistry &r = order_registry();
    std::lock_guard<std::mutex> g(r.mu);
    for (auto it = r.by_stream.begin(); it != r.by_stream.end();)
        it = it->second == oc ? r.by_stream.erase(it) : std::next(it);
}

// ---- the hooks (the engine's translation units call the HIP names; the macros at the end route them through here)
inline hipError_t ord_event_record(hipEvent_t e, hipStream_t s)
{
    const hipError_t rc = hipEventRecord(e, s);
    with_order_check(s, [&](OrderCheck &oc) { oc.record(e, s); });
    return rc;
}
inline hipError_t ord_stream_wait_event(hipStream_t s, hipEvent_t e, unsigned flags)
{
    const hipError_t rc = hipStreamWaitEvent(s, e, flags);
    // (HCV_ORDER_CHECK_DROP=k: the checker — not the device — overlooks every k-th wait; the test suite's proof that a missing wait is found)
    static const int drop = std::getenv("HCV_ORDER_CHECK_DROP") ? std::atoi(std::getenv("HCV_ORDER_CHECK_DROP")) : 0;
    static std::atomic<long long> waits{0};
    if (drop > 0 && order_mode() > 0 && (waits.fetch_add(1) + 1) % drop == 0) return rc;
    with_order_check(s, [&](OrderCheck &oc) { oc.wait(s, e); });
    return rc;
}
inline hipError_t ord_stream_synchronize(hipStream_t s)
{
    const hipError_t rc = hipStreamSynchronize(s);
    with_order_check(s, [&](OrderCheck &oc) { oc.host_sync_stream(s); });
    return rc;
}
inline hipError_t ord_event_synchronize(hipEvent_t e)
{
    const hipError_t rc = hipEventSynchronize(e);
    if (order_mode() > 0)
    {
        // (an event does not say whose it is: every engine that has recorded it learns of the host's wait)
        OrderRegistry &r = order_registry();
        std::lock_guard<std::mutex> g(r.mu);
        OrderCheck *last = nullptr;
        for (auto &kv : r.by_stream)
            if (kv.second != last)
            {
                kv.second->host_sync_event(e);
                last = kv.second;
            }
    }
    return rc;
}
inline hipError_t ord_event_query_done(hipEvent_t e)
{
    const hipError_t rc = hipEventQuery(e);
    if (rc == hipSuccess && order_mode() > 0)
    {
        OrderRegistry &r = order_registry();
        std::lock_guard<std::mutex> g(r.mu);
        OrderCheck *last = nullptr;
        for (auto &kv : r.by_stream)
            if (kv.second != last)
            {
                kv.second->host_sync_event(e);
                last = kv.second;
            }
    }
    return rc;
}

#define ORD_ACCESS(stream, buf, lo, hi, ring, write, what)                                                                                         \
    do                                                                                                                                             \
    {                                                                                                                                              \
        if (::hcv::order_mode() > 0)                                                                                                               \
            ::hcv::with_order_check((stream), [&](::hcv::OrderCheck &oc_) { oc_.access((stream), (buf), (lo), (hi), (ring), (write), (what)); }); \
    } while (0)
#define ORD_MEET(from, to)                                                                                                                         \
    do                                                                                                                                             \
    {                                                                                                                                              \
        if (::hcv::order_mode() > 0) ::hcv::with_order_check((to), [&](::hcv::OrderCheck &oc_) { oc_.meet((from), (to)); });                      \
    } while (0)

}  // namespace hcv

// From here on the engine's code says hipEventRecord / hipStreamWaitEvent / hipStreamSynchronize / hipEventSynchronize / hipEventQuery and gets
// the hooked forms (all of them with their full argument lists in this code base).
#define hipEventRecord(e, s) ::hcv::ord_event_record((e), (s))
#define hipStreamWaitEvent(s, e, f) ::hcv::ord_stream_wait_event((s), (e), (f))
#define hipStreamSynchronize(s) ::hcv::ord_stream_synchronize((s))
#define hipEventSynchronize(e) ::hcv::ord_event_synchronize((e))
#define hipEventQuery(e) ::hcv::ord_event_query_done((e))
