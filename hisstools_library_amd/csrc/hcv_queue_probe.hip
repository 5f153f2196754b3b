// Which hardware queue does a stream feed?  HIP hands every stream of a process one of a handful of hardware queues (four by default) —
// the least used one at the moment the stream is made, so which streams end up together depends on everything the process made before
// (other engines, the framework around us) and, between equally used queues, on their addresses.  Work of two streams in one queue runs in
// the order it was enqueued: an engine of twelve streams whose five busy ones — the pivot stage's two lanes, the two far-tail rungs and
// the main stream's emit — happen to pair the emit with a rung's 0.3 ms multiply-accumulate loses a quarter of its throughput (c5 on the
// ladder as bench.py's second engine: 0.150 ms per step in such processes against 0.118 in the others, profiles/r05_queue_probe.txt).
//
// There is no API that tells the queue of a stream; there is an experiment: hold stream A with a kernel that waits for a word in host
// memory (bounded: 2 ms, released after ~0.15), send an empty kernel down stream B and watch whether it gets through.  spread_streams() places
// an engine's busy streams at its creation — nothing of the engine is enqueued yet — by what that experiment has found about the pooled
// streams, running it only for streams never seen before and only while no engine of the device is streaming (QueueMap below).
#include "hcv_engine_impl.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>

namespace hcv {
namespace {

__global__ void hold_kernel(const volatile unsigned *release, unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();                   // (100 MHz)
    while (!*release && wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

__global__ void touch_kernel() {}

struct Probe
{
    unsigned *flag = nullptr, *flag_dev = nullptr;
    std::vector<hipEvent_t> ev;
    bool ok = false;

    bool init()
    {
        if (hipHostMalloc((void **) &flag, sizeof(unsigned), hipHostMallocMapped) != hipSuccess) return false;
        if (hipHostGetDevicePointer((void **) &flag_dev, flag, 0) != hipSuccess) return false;
        ok = true;
        return true;
    }
    ~Probe()
    {
        for (hipEvent_t e : ev) (void) hipEventDestroy(e);
        if (flag) (void) hipHostFree(flag);
        (void) hipGetLastError();
    }

    // bit i set: work sent down reps[i] waited for the kernel holding c — one hardware queue.  -1: the experiment itself failed.
    long shares(hipStream_t c, const std::vector<hipStream_t> &reps)
    {
        while (ev.size() < reps.size())
        {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return -1;
            ev.push_back(e);
        }
        *reinterpret_cast<volatile unsigned *>(flag) = 0;
        hold_kernel<<<1, 64, 0, c>>>(flag_dev, 200000ull);           // (2 ms at most; released below after ~0.15 ms)
        for (size_t i = 0; i < reps.size(); i++)
        {
            touch_kernel<<<1, 64, 0, reps[i]>>>();
            if (hipEventRecord(ev[i], reps[i]) != hipSuccess) return -1;
        }
        long pending = (1l << reps.size()) - 1;
        const auto t0 = std::chrono::steady_clock::now();
        while (pending && std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(150))
            for (size_t i = 0; i < reps.size(); i++)
                if ((pending >> i & 1) && hipEventQuery(ev[i]) == hipSuccess) pending &= ~(1l << i);
        // (one more look after the window: a host thread that lost its core inside it comes back to find everything that COULD get through
        // through, while what sits behind the held kernel still waits — the kernel holds for 2 ms on its own)
        for (size_t i = 0; i < reps.size(); i++)
            if ((pending >> i & 1) && hipEventQuery(ev[i]) == hipSuccess) pending &= ~(1l << i);
        (void) hipGetLastError();                                   // (hipErrorNotReady of the queries)
        *reinterpret_cast<volatile unsigned *>(flag) = 1;
        bool fine = hipStreamSynchronize(c) == hipSuccess;
        for (hipStream_t r : reps) fine = hipStreamSynchronize(r) == hipSuccess && fine;
        return fine ? pending : -1;
    }
};

}  // namespace

// What the experiment has found, per device and for the life of the process (round 6): streams are pooled and never destroyed
// (hcv_engine.hip: stream_take / stream_give) and keep the hardware queue they were given, so a stream is classified ONCE — the engines a
// process makes after its first few find every pooled stream known and run no experiment at all.  Each queue seen has a representative
// stream of its own that is never handed out (the touch kernels go down those, never down a live engine's stream).  And the experiment
// parks a kernel on a hardware queue for 0.15 ms: queues are process-wide, so while ANY engine of the device is streaming (a process call
// within the last 0.4 s, Engine::audio_enter) unknown streams stay unknown — the placement is made from what is known, or left as it is.
struct QueueMap
{
    std::mutex mtx;
    std::vector<hipStream_t> reps;                                  // class -> its representative
    std::map<hipStream_t, int> cls;
};
static QueueMap &queue_map(int device)
{
    static std::mutex m;
    static std::map<int, QueueMap *> maps;
    std::lock_guard<std::mutex> g(m);
    QueueMap *&q = maps[device];
    if (!q) q = new QueueMap();
    return *q;
}
static std::atomic<long long> gDeviceAudioNs[64];
void note_device_streaming(int device, long long now_ns)
{
    if (device >= 0 && device < 64) gDeviceAudioNs[device].store(now_ns, std::memory_order_relaxed);
}
static bool device_streaming(int device)
{
    if (device < 0 || device >= 64) return false;
    const long long now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    const long long last = gDeviceAudioNs[device].load(std::memory_order_relaxed);
    return last != 0 && now - last < 400000000ll;
}

// roles[0 .. n): the engine's busy streams, none of them in use yet.  On return they feed pairwise different hardware queues as far as the
// process has queues to give, none of them the queue of `anchor` — but roles[share] (share >= 0), which feeds exactly that one.  A role that
// cannot be served keeps the stream it had.  Returns the number of streams replaced (0: the ones at hand were fine), -1 when the placement
// could not be made (an experiment failed, or was not allowed and too little was known: nothing changed).
int spread_streams(hipStream_t anchor, hipStream_t **roles, int n, int share)
{
    int device = 0;
    (void) hipGetDevice(&device);
    static const bool allow = !(std::getenv("HCV_QUEUE_PROBE") && std::atoi(std::getenv("HCV_QUEUE_PROBE")) == 0);
    if (!allow || n <= 0 || n > 6) return -1;
    QueueMap &qm = queue_map(device);
    std::lock_guard<std::mutex> g(qm.mtx);                          // (engine creations of a device one after the other: a control path)
    const bool may_probe = !device_streaming(device);
    Probe px;
    bool warmed = false;
    int experiments = 0;
    // the class of an idle stream: what is known, or — where an experiment may run now — found out.  -1: the experiment failed; -2: unknown
    // and no experiment allowed
    auto classify = [&](hipStream_t s) -> int
    {
        auto it = qm.cls.find(s);
        if (it != qm.cls.end()) return it->second;
        if (!may_probe) return -2;
        if (!px.ok && !px.init()) return -1;
        if (!warmed)
        {
            // (the first launch of a kernel loads its code: not inside the watched window)
            *px.flag = 1;
            hold_kernel<<<1, 64, 0, s>>>(px.flag_dev, 1ull);
            touch_kernel<<<1, 64, 0, s>>>();
            if (hipStreamSynchronize(s) != hipSuccess) return -1;
            warmed = true;
        }
        experiments++;
        const long m = qm.reps.empty() ? 0 : px.shares(s, qm.reps);
        if (m < 0) return -1;
        for (size_t i = 0; i < qm.reps.size(); i++)
            if (m >> i & 1) return qm.cls[s] = (int) i;
        if (qm.reps.size() >= 12) return qm.cls[s] = (int) qm.reps.size();    // (more queues than anybody needs: all further ones count as one)
        // a queue not seen before: a stream of its own stands for it from now on (`s` itself goes to whoever asked)
        hipStream_t rep = nullptr;
        for (int tries = 0; tries < 24; tries++)
        {
            hipStream_t c = nullptr;
            if (hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) return -1;
            std::vector<hipStream_t> one{c};
            const long mm = px.shares(s, one);
            if (mm < 0) return -1;
            if (mm & 1) { rep = c; break; }
            stream_give(device, c);                                 // (another queue's: into the pool, to be classified when somebody takes it)
        }
        if (!rep) return -1;
        qm.reps.push_back(rep);
        return qm.cls[s] = (int) qm.reps.size() - 1;
    };

    struct Cand { hipStream_t s; int cls; bool made; };
    std::vector<Cand> cands;
    auto cleanup = [&](int rc)
    {
        for (Cand &c : cands)
            if (c.made && c.s) stream_give(device, c.s);              // (idle: every experiment ends with its streams synchronized)
        return rc;
    };
    const int anchor_cls = classify(anchor);
    if (anchor_cls < 0) return -1;
    for (int r = 0; r < n; r++)
    {
        const int cls = classify(*roles[r]);
        if (cls == -1) return cleanup(-1);
        cands.push_back({*roles[r], cls, false});
    }
    // pick[r] = candidate of role r, -1 = none; returns the roles served
    auto assign = [&](std::vector<int> &pick) -> int
    {
        std::vector<char> used(qm.reps.size() + 2, 0);
        pick.assign(n, -1);
        int served = 0;
        // first the stream a role has, if it will do; then a made one of a queue nobody has taken
        for (int pass = 0; pass < 2; pass++)
            for (int r = 0; r < n; r++)
            {
                if (pick[r] >= 0) continue;
                for (int c = pass == 0 ? r : n; c < (pass == 0 ? r + 1 : (int) cands.size()); c++)
                {
                    bool taken = false;
                    for (int p : pick) taken = taken || p == c;
                    const int cls = cands[c].cls;
                    if (cls < 0) continue;                          // (unknown, and no experiment allowed now)
                    const bool fits = r == share ? cls == anchor_cls : cls != anchor_cls;
                    if (taken || used[cls] || !fits) continue;
                    pick[r] = c;
                    used[cls] = 1;
                    served++;
                    break;
                }
            }
        return served;
    };
    std::vector<int> pick;
    int served = assign(pick);
    for (int extra = 0; served < n && extra < 16; extra++)
    {
        hipStream_t s = nullptr;
        if (stream_take(device, &s) != hipSuccess) break;
        const int cls = classify(s);
        if (cls == -1)
        {
            (void) hipStreamSynchronize(s);
            stream_give(device, s);
            return cleanup(-1);
        }
        cands.push_back({s, cls, true});
        served = assign(pick);
    }
    int replaced = 0;
    for (int r = 0; r < n; r++)
    {
        if (pick[r] < n) continue;                                  // (served by its own stream, or not served: it keeps what it had)
        Cand &c = cands[pick[r]];
        stream_give(device, *roles[r]);
        *roles[r] = c.s;
        c.made = false;                                             // (taken: not the cleanup's any more)
        replaced++;
    }
    if (std::getenv("HCV_VERBOSE"))
    {
        std::fprintf(stderr, "[hcv] queue probe: %zu queues known, %d experiments%s, main stream on queue %d, %d of %d busy streams placed, %d replaced; classes:",
                     qm.reps.size(), experiments, may_probe ? "" : " (none allowed: the device is streaming)", anchor_cls, served, n, replaced);
        for (int r = 0; r < n; r++) std::fprintf(stderr, " %d", pick[r] >= 0 ? cands[pick[r]].cls : -1);
        std::fprintf(stderr, "\n");
    }
    return cleanup(replaced);
}

}  // namespace hcv
