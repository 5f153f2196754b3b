// C ABI (include/hisstools_amd.h) and the host-side class semantics of the reference API
// (error codes, clamps, silent-pair rules) on top of the device engine.
//
// Each hcv_* object mirrors one reference class; the arithmetic lives in hcv_kernels.hip, the
// device orchestration in hcv_engine.hip.  Nothing here computes audio on the CPU.

#include "../../include/hisstools_amd.h"
#include "hcv_engine.h"
#include "hcv_api_common.h"
#include "hcv_rccl.h"
#include "hcv_shard_pool.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using hcv::Engine;
using hcv::EngineCfg;
using hcv::StageCfg;

// shared with hcv_api_fft.hip (hcv_api_common.h)
thread_local std::string hcv_api::tlsError;
int hcv_api::gDefaultDevice = -1;
using hcv_api::gDefaultDevice;
using hcv_api::set_error;
using hcv_api::tlsError;

// ------------------------------------------------------------------------------------------------ helpers

// PartitionedConvolve::log2 (PartitionedConvolve.cpp:114-129): ceil(log2(v)), 0 for v <= 1
static unsigned part_log2(uintptr_t value)
{
    unsigned count = 0;
    for (uintptr_t v = value; v; v >>= 1) count++;
    if (!count) return 0;
    return (value == (uintptr_t(1) << (count - 1))) ? count - 1 : count;
}

// ------------------------------------------------------------------------------------------------ PartitionedConvolve

struct hcv_partitioned
{
    uintptr_t offset = 0, length = 0, maxImpulseLength = 0;
    unsigned maxLog2 = 0, log2 = 0, engineLog2 = 0;
    intptr_t resetOffset = -1;
    uintptr_t numPartitions = 0;
    std::unique_ptr<Engine> engine;

    bool build()
    {
        EngineCfg cfg;
        cfg.nin = cfg.nout = 1;
        cfg.device = gDefaultDevice;
        StageCfg st;
        st.fft_size = 1u << log2;
        st.offset = offset;
        st.length = length;
        st.capacity = maxImpulseLength;
        cfg.stages.push_back(st);
        std::string err;
        engine.reset(Engine::create(cfg, &err));
        if (!engine) set_error(err);
        engineLog2 = log2;
        return (bool) engine;
    }
};

extern "C" int hcv_partitioned_set_fft_size(hcv_partitioned *h, uintptr_t FFTSize)
{
    unsigned l2 = part_log2(FFTSize);
    int error = HCV_ERR_NONE;
    if (l2 < 5 || l2 > h->maxLog2) return HCV_ERR_FFT_SIZE_OUT_OF_RANGE;
    if (FFTSize != (uintptr_t(1) << l2)) error = HCV_ERR_FFT_SIZE_NON_POWER_OF_TWO;
    if (l2 != h->log2)
    {
        h->numPartitions = 0;
        h->log2 = l2;
    }
    return error;
}

extern "C" int hcv_partitioned_set_length(hcv_partitioned *h, uintptr_t length)
{
    h->length = std::min(length, h->maxImpulseLength);
    return length > h->maxImpulseLength ? HCV_ERR_PARTITION_LENGTH_TOO_LARGE : HCV_ERR_NONE;
}

extern "C" void hcv_partitioned_set_offset(hcv_partitioned *h, uintptr_t offset) { h->offset = offset; }
extern "C" void hcv_partitioned_set_reset_offset(hcv_partitioned *h, intptr_t offset) { h->resetOffset = offset; }

extern "C" hcv_partitioned *hcv_partitioned_create(uintptr_t maxFFTSize, uintptr_t maxLength, uintptr_t offset, uintptr_t length)
{
    std::unique_ptr<hcv_partitioned> h(new hcv_partitioned());
    unsigned ml2 = part_log2(maxFFTSize);                   // setMaxFFTSize, .cpp:26-50 (errors discarded by the ctor)
    if (ml2 > 20) ml2 = 20;
    if (ml2 < 5) ml2 = 5;
    h->maxLog2 = ml2;
    h->maxImpulseLength = maxLength;
    hcv_partitioned_set_fft_size(h.get(), uintptr_t(1) << ml2);
    hcv_partitioned_set_offset(h.get(), offset);
    hcv_partitioned_set_length(h.get(), length);
    const uintptr_t maxh = (uintptr_t(1) << ml2) >> 1;      // .cpp:77-82
    if (h->maxImpulseLength % maxh) h->maxImpulseLength = (h->maxImpulseLength / maxh + 1) * maxh;
    if (!h->build()) return nullptr;
    return h.release();
}

extern "C" void hcv_partitioned_destroy(hcv_partitioned *h) { delete h; }

extern "C" int hcv_partitioned_set(hcv_partitioned *h, const float *input, uintptr_t length)
{
    int error = HCV_ERR_NONE;
    uintptr_t load = (!input || length <= h->offset) ? 0 : length - h->offset;
    load = (h->length && h->length < load) ? h->length : load;
    if (load > h->maxImpulseLength) error = HCV_ERR_MEM_ALLOC_TOO_SMALL;

    if (h->engineLog2 != h->log2 && !h->build())
    {
        h->numPartitions = 0;
        return HCV_ERR_MEM_UNAVAILABLE;
    }
    h->engine->set_stage_window(0, h->offset, h->length);
    if (!h->engine->set_ir(0, 0, input, input ? length : 0, false))
    {
        set_error(h->engine->last_error());
        h->numPartitions = 0;
        return HCV_ERR_MEM_UNAVAILABLE;
    }
    h->numPartitions = h->engine->stage_partitions(0, 0, 0);
    return error;
}

extern "C" void hcv_partitioned_reset(hcv_partitioned *h) { h->engine->reset_pair(0, 0); }

extern "C" int hcv_partitioned_process(hcv_partitioned *h, const float *in, float *out, uintptr_t numSamples)
{
    if (!h->numPartitions) return 0;                        // .cpp:262-263: out untouched
    const float *ins[1] = { in };
    float *outs[1] = { out };
    if (!h->engine->process(ins, outs, 1, 1, numSamples, false))
    {
        set_error(h->engine->last_error());
        return -1;
    }
    return 1;
}

// ------------------------------------------------------------------------------------------------ TimeDomainConvolve

struct hcv_timedomain
{
    uintptr_t offset = 0, length = 0, taps = 0;
    std::unique_ptr<Engine> engine;
};

extern "C" int hcv_timedomain_set_length(hcv_timedomain *h, uintptr_t length)
{
    h->length = std::min(length, uintptr_t(2044));
    return length > 2044 ? HCV_ERR_TIME_LENGTH_OUT_OF_RANGE : HCV_ERR_NONE;
}

extern "C" void hcv_timedomain_set_offset(hcv_timedomain *h, uintptr_t offset) { h->offset = offset; }

extern "C" hcv_timedomain *hcv_timedomain_create(uintptr_t offset, uintptr_t length)
{
    std::unique_ptr<hcv_timedomain> h(new hcv_timedomain());
    hcv_timedomain_set_offset(h.get(), offset);
    hcv_timedomain_set_length(h.get(), length);
    EngineCfg cfg;
    cfg.nin = cfg.nout = 1;
    cfg.has_td = true;
    cfg.td_offset = h->offset;
    cfg.td_length = h->length;
    cfg.device = gDefaultDevice;
    std::string err;
    h->engine.reset(Engine::create(cfg, &err));
    if (!h->engine)
    {
        set_error(err);
        return nullptr;
    }
    return h.release();
}

extern "C" void hcv_timedomain_destroy(hcv_timedomain *h) { delete h; }

extern "C" int hcv_timedomain_set(hcv_timedomain *h, const float *input, uintptr_t length)
{
    h->engine->set_td_window(h->offset, h->length);
    const bool have = input && length > h->offset;
    if (!h->engine->set_ir(0, 0, have ? input : nullptr, have ? length : 0, false)) set_error(h->engine->last_error());
    h->taps = h->engine->td_taps(0, 0);
    // the reference evaluates (length - mOffset) in unsigned arithmetic even when length <= mOffset (.cpp:86)
    return (!h->length && (uintptr_t) (length - h->offset) > 2044) ? HCV_ERR_TIME_IMPULSE_TOO_LONG : HCV_ERR_NONE;
}

extern "C" void hcv_timedomain_reset(hcv_timedomain *h) { h->engine->reset_pair(0, 0); }

extern "C" int hcv_timedomain_process(hcv_timedomain *h, const float *in, float *out, uintptr_t numSamples)
{
    const float *ins[1] = { in };
    float *outs[1] = { out };
    if (!h->engine->process(ins, outs, 1, 1, numSamples, false))     // with no taps this writes zeros, as .cpp:100-125 does
    {
        set_error(h->engine->last_error());
        return -1;
    }
    return h->taps ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ Mono / NToMono / Convolver
//
// One engine holds the whole matrix; this struct keeps the per-pair bookkeeping the reference keeps inside each
// MonoConvolve (mLength, and the MemorySwap'd tail's logical size / presence).

namespace
{
    struct Layout
    {
        std::vector<uint32_t> sizes;
        bool zeroLatency = false;
        uint32_t tailOffset = 0, largest = 0;
        int pivot = -1;                          // extended ladder: index of the reference's own tail stage, in front of the rungs (-1: no rungs)
        std::vector<StageCfg> fixedStages;       // the non-resizable PartitionedConvolves, in order
        std::string error;

        // MonoConvolve::setPartitions (MonoConvolve.cpp:203-258)
        bool build(bool zero, uint32_t A, uint32_t B, uint32_t C, uint32_t D)
        {
            zeroLatency = zero;
            const uint32_t req[4] = { A, B, C, D };
            uint32_t prev = 0;
            for (int i = 0; i < 4; i++)
            {
                const int size = (int) req[i], p = (int) prev;
                if (size >= (1 << 5) && size <= (1 << 20) && size > p)
                    sizes.push_back(req[i]);
                else if (size)
                {
                    error = "invalid FFT size or order";
                    return false;
                }
                prev = req[i];
            }
            if (sizes.empty())
            {
                error = "no valid FFT sizes given";
                return false;
            }
            const size_t ns = sizes.size();
            uint32_t offset = zeroLatency ? sizes[0] >> 1 : 0;
            largest = sizes[ns - 1];
            auto part = [&](uint32_t size, uint32_t next)
            {
                StageCfg st;
                st.fft_size = size;
                st.offset = offset;
                st.length = (next - size) >> 1;
                st.capacity = st.length;
                fixedStages.push_back(st);
                offset += (next - size) >> 1;
            };
            if (ns == 4) part(sizes[0], sizes[1]);
            if (ns > 2) part(sizes[ns - 3], sizes[ns - 2]);
            if (ns > 1) part(sizes[ns - 2], sizes[ns - 1]);
            tailOffset = offset;
            return true;
        }

        // MI355X extension (no reference analogue): continue the non-uniform partitioning ladder past the reference's
        // largest FFT.  A stage of hop H can serve IR samples from offset H - latency onwards, so the reference's tail
        // (hop largest/2) is capped where a `ratio` times larger FFT can take over, and so on up to 2^20.  The sum is
        // the same convolution; the far tail just moves ratio x fewer bytes per sample per rung.  Only rungs the IR can
        // reach (maxLength) are created.
        // boundaryBytes > 0 (the automatic rule): no rung whose hop boundary would have to stream more than that for its first
        // partition alone (8 bytes x bins x pairs, all of it inside ONE process call, beside the rung's forward and inverse transforms)
        void extend_tail(uint64_t maxLength, uint32_t ratio, uint64_t pairs = 0, uint64_t boundaryBytes = 0)
        {
            if (ratio < 2) return;
            const uint64_t latency = zeroLatency ? 0 : sizes[0] >> 1;
            uint64_t cur = largest, curOffset = tailOffset;
            const int ownTail = (int) fixedStages.size();
            // (at most three rungs: the engine holds kMaxStages = 8 stages, the reference's own ladder is up to four of them and the head one —
            // ratio 2 past a long impulse response asked for six and the object could not be made)
            while (cur * ratio <= (uint64_t(1) << 20) && (int) fixedStages.size() - ownTail < 3)
            {
                const uint64_t next = cur * ratio, nextOffset = (next >> 1) - latency;
                if (nextOffset >= maxLength) break;
                if (boundaryBytes && 8 * (next >> 1) * pairs > boundaryBytes) break;
                StageCfg st;
                st.fft_size = (uint32_t) cur;
                st.offset = curOffset;
                st.length = nextOffset - curOffset;
                st.capacity = st.length;
                fixedStages.push_back(st);
                cur = next;
                curOffset = nextOffset;
            }
            if ((int) fixedStages.size() > ownTail) pivot = ownTail;
            largest = (uint32_t) cur;
            tailOffset = (uint32_t) curOffset;
        }

        // capacity of the tail PartitionedConvolve for a MemorySwap size (allocator lambda, :249-252)
        uint64_t tail_capacity(uint64_t size) const
        {
            const uint64_t reach = std::max<uint64_t>(size, largest);
            return reach > tailOffset ? reach - tailOffset : largest >> 1;
        }
    };

    // The extended far-tail ladder for callers that never heard of it (the reference-shaped constructors): HCV_TAIL_RATIO = 2 / 4 / 8
    // always continues the partition ladder past the reference's largest FFT, 0 never does; unset = the rule below.
    constexpr uint32_t kLadderAuto = 0xFFFFFFFFu;           // make_matrix: "what the environment / the rule says"
    uint32_t env_tail_ratio()
    {
        static const int r = std::getenv("HCV_TAIL_RATIO") ? std::atoi(std::getenv("HCV_TAIL_RATIO")) : -1;
        return r < 0 ? kLadderAuto : (r == 2 || r == 4 || r == 8) ? (uint32_t) r : 0u;
    }
    // Unset: ratio 8 where the reference's tail would be HBM-bound — at least 32 partitions of it and at least 1 GiB of tail spectra
    // over the matrix (c5 703 partitions / 11.8 GB: 8.8 x faster on the ladder, 64 x 64 with 10 s IRs 58 / 15.7 GB: 2.5 x; 64 x 64 with
    // 2 s IRs, 11 partitions, and the cache-resident one-output engines, which run as ONE launch per block, are faster as they are).
    constexpr uint64_t kLadderMinParts = 32, kLadderMinBytes = uint64_t(1) << 30;
    // ... and no rung whose hop boundary — its big transforms and its first partition's multiply-accumulate, all in the ONE call that
    // completes the hop — would hold a real-time caller up for more than about a millisecond: 16 x 16 takes both rungs (the 2^20-point
    // rung's boundary call: 0.85 ms once per 5.5 s), 64 x 64 the 131072-point rung only (0.5 - 1.0 ms once per 1.4 s; a 2^20-point
    // rung's boundary would stream 17 GB).  HCV_TAIL_RATIO and the explicit constructors are not bound by it.
    constexpr uint64_t kLadderBoundaryBytes = 2500000000ull;

    struct Matrix
    {
        Layout layout;
        uint32_t nin = 1, nout = 1;
        bool diag = false;
        std::unique_ptr<Engine> engine;
        // what the object was built from, kept for a re-layout (relayout_for)
        struct { bool zero = false; uint32_t A = 0, B = 0, C = 0, D = 0; int device = 0; uint32_t maxBlock = 0; } args;
        uint32_t ladder = 0;                // 0: the layout is the caller's; 2 / 4 / 8: that ratio; kLadderAuto: the rule
        uint64_t laidFor = 0;               // the longest impulse response the current stage list was laid out for
        bool pristineOnly = false;          // (a shard of a sharded object: re-laid only before its first process call)
        std::atomic<bool> everProcessed { false };
        // audio-side calls in flight / a control call replacing the engine of an EMPTY object (the audio side never waits for it:
        // the block of an empty object is silence either way)
        std::atomic<int> users { 0 };
        std::atomic<bool> swapping { false };
        // (a shard of a sharded object: the engine a process call uses is read ONCE per call, first_use, and an engine replaced before the
        // shard's first call is kept until the object goes — a call that raced the replacement holds either one, both empty, and never waits)
        std::atomic<Engine *> live { nullptr };
        std::vector<std::unique_ptr<Engine>> retired;
        std::vector<uint64_t> mLength, part4Size;
        std::vector<uint8_t> part4Alloc;
        intptr_t resetOffset = -1;
        mutable std::mutex stateMutex;      // per-pair bookkeeping is written by set/resize (control thread) and read by process

        size_t pair(uint32_t in, uint32_t out) const { return (size_t) out * (diag ? 1 : nin) + (diag ? 0 : in); }
        size_t tail() const { return layout.fixedStages.size(); }

        bool build(uint32_t numIns, uint32_t numOuts, bool parallel, uint64_t maxLength, int device, uint32_t maxBlock)
        {
            nin = parallel ? numOuts : numIns;
            nout = numOuts;
            diag = parallel;
            args.device = device;
            args.maxBlock = maxBlock;
            laidFor = maxLength;
            engine.reset(make_engine(layout, maxLength));
            if (!engine) return false;
            live.store(engine.get(), std::memory_order_release);
            const size_t pairs = (size_t) nout * (diag ? 1 : nin);
            mLength.assign(pairs, 0);
            part4Size.assign(pairs, maxLength);                          // part4.equal(..., maxLength), :254
            part4Alloc.assign(pairs, maxLength ? 1 : 0);
            return true;
        }

        // (the boundary bound applies where the RULE chose the ladder)
        uint64_t boundary_bound() const { return (ladder == kLadderAuto && env_tail_ratio() == kLadderAuto) ? kLadderBoundaryBytes : 0; }
        uint64_t pair_count() const { return (uint64_t) nout * (diag ? 1 : nin); }

        uint32_t ratio_for(uint64_t length) const
        {
            if (ladder != kLadderAuto) return ladder;
            const uint32_t e = env_tail_ratio();
            if (e != kLadderAuto) return e;
            Layout base;
            if (!base.build(args.zero, args.A, args.B, args.C, args.D) || base.largest != 16384 || length <= base.tailOffset) return 0;
            const uint64_t hop = base.largest >> 1, parts = (length - base.tailOffset + hop - 1) / hop;
            const uint64_t bytes = parts * hop * 8 * (uint64_t) nout * (diag ? 1 : nin);
            if (!(parts >= kLadderMinParts && bytes >= kLadderMinBytes)) return 0u;
            // Which ratio: every stage reads its partitions once per hop of ITS size, so per 8192 samples a ladder moves (sum of its
            // partitions) x pairs x 64 KiB whatever the stages' sizes — a smaller ratio means fewer partitions per rung but more rungs, and
            // a rung is a chain of launches (~15 us per 8192 samples beside the others).  Measured per 8192-sample step, ratios 8 / 4 / 2
            // (explicit constructors: three rungs at most, no boundary bound):
            //   16 x 16, 60 s @ 96 kHz   0.1175 / 0.1232 / 0.333 ms   (26 / 21 / 92 partitions)
            //   64 x 64, 10 s @ 48 kHz   0.690  / 0.511  / 0.616 ms   (15 / 10 / 11 partitions; ratio 4 has a 262144-point rung whose boundary
            //                                                           call streams 4.3 GB, beyond the bound below; ratio 2 stays within it)
            // The RULE keeps ratio 8: the paced real-time behaviour of the three-rung ratio-2 layout (three boundaries instead of one) has
            // not been measured yet, and that is what the bound exists for.  Callers after throughput ask for their ratio
            // (hcv_convolver_create_extended, HCV_TAIL_RATIO): bench.py's extended leg of the 64 x 64 / 10 s shape takes 4.
            return 8u;
        }

        // An EMPTY object (no pair loaded) asked to hold a longer impulse response than its stage list was laid out for: lay the
        // ladder out again for that length and replace the engine.  Nothing audible can change — no pair is loaded, the next set()
        // restarts its pair from silence anyway (MonoConvolve.cpp:139-150) — and a running audio thread never waits: a process call
        // that meets the swap writes the silence the empty object would have produced.  Once pairs are loaded the stage list stays
        // (a longer IR then extends the last stage, as the reference extends its tail).  Caller holds stateMutex.
        void relayout_for(uint64_t length)
        {
            if (!ladder || length <= laidFor) return;
            for (uint64_t l : mLength)
                if (l) return;
            if (pristineOnly && everProcessed.load(std::memory_order_acquire)) return;
            const uint32_t ratio = ratio_for(length);
            Layout nl;
            if (!nl.build(args.zero, args.A, args.B, args.C, args.D)) return;
            nl.extend_tail(length, ratio, pair_count(), boundary_bound());
            laidFor = length;
            if (nl.fixedStages.size() == layout.fixedStages.size()) return;            // the same stage list
            uint64_t cap = length;
            for (uint64_t s : part4Size) cap = std::max(cap, s);
            std::unique_ptr<Engine> fresh(make_engine(nl, cap));                        // (milliseconds: before any flag is raised)
            if (!fresh) return;                                                         // (the old layout stays; the error text is set)
            fresh->set_profiling(engine->profiling());
            swapping.store(true, std::memory_order_seq_cst);
            // A shard's engine pointer is held by the shard pool's jobs across a block, without a count in `users`: its first process
            // call and this swap settle it between them (Dekker: each side writes its flag, then reads the other's, both seq_cst — at
            // least one of them sees the other).  The call raises everProcessed and takes the engine that is live (first_use: the replaced
            // one is kept alive); this side, finding everProcessed, leaves the engine be.
            if (pristineOnly && everProcessed.load(std::memory_order_seq_cst))
            {
                swapping.store(false, std::memory_order_release);
                return;                                                                 // (`fresh` is destroyed: the shard keeps its stage list)
            }
            while (users.load(std::memory_order_seq_cst) != 0) std::this_thread::yield();
            engine.swap(fresh);
            layout = nl;
            live.store(engine.get(), std::memory_order_seq_cst);
            swapping.store(false, std::memory_order_release);
            // (`fresh` now holds the old engine: destroyed here, after whatever it still had in flight — but a shard's is KEPT: its first
            // process call may have read the pointer a moment ago, first_use)
            if (pristineOnly) retired.push_back(std::move(fresh));
        }

        // a shard's process call: the engine this call runs on.  The first call raises everProcessed — from here on the shard keeps its stage
        // list — and takes whichever engine is live at that moment; it never waits for a replacement in progress (round 5 polled `swapping`
        // here): the replaced engine stays alive and is as empty as its successor.
        Engine *first_use()
        {
            if (!everProcessed.load(std::memory_order_relaxed)) everProcessed.store(true, std::memory_order_seq_cst);
            return live.load(std::memory_order_seq_cst);
        }

        Engine *make_engine(const Layout &l, uint64_t maxLength)
        {
            EngineCfg cfg;
            cfg.nin = nin;
            cfg.nout = nout;
            cfg.diag = diag;
            cfg.device = args.device;
            cfg.max_block = args.maxBlock;
            if (l.zeroLatency)
            {
                cfg.has_td = true;
                cfg.td_offset = 0;
                cfg.td_length = l.sizes[0] >> 1;                        // TimeDomainConvolve(0, A/2), :240
                if (cfg.td_length > 2044) cfg.td_length = 2044;         // TimeDomainConvolve::setLength clamp
            }
            cfg.stages = l.fixedStages;
            cfg.pivot = l.pivot;
            StageCfg tl;
            tl.fft_size = l.largest;
            tl.offset = l.tailOffset;
            tl.length = 0;
            tl.capacity = l.tail_capacity(maxLength);
            cfg.stages.push_back(tl);
            std::string err;
            Engine *e = Engine::create(cfg, &err);
            if (!e) set_error(err);
            return e;
        }

        // MemorySwap::equal on the tail (MemorySwap.h:209-229)
        void tail_equal(size_t p, uint64_t length)
        {
            if (length == part4Size[p]) return;
            if (engine->ensure_stage_capacity(tail(), layout.tail_capacity(length)))
            {
                part4Alloc[p] = 1;
                part4Size[p] = length;
            }
            else
            {
                part4Alloc[p] = 0;
                part4Size[p] = 0;
            }
        }

        // MonoConvolve::process, .cpp:181-183: mPart4.attempt() — while a control call holds the pair (set / resize in
        // progress) the audio thread does not wait, the pair is silent for the block
        bool active(size_t p) const
        {
            std::unique_lock<std::mutex> g(stateMutex, std::try_to_lock);
            if (!g.owns_lock()) return false;
            return mLength[p] && mLength[p] <= part4Size[p];
        }

        // MonoConvolve::resize (.cpp:101-110)
        int resize(uint32_t in, uint32_t out, uint64_t length)
        {
            std::lock_guard<std::mutex> g(stateMutex);
            const size_t p = pair(in, out);
            mLength[p] = 0;
            relayout_for(length);
            engine->set_ir(in, out, nullptr, 0, false);                  // the pair is silent until the next set
            tail_equal(p, length);
            return part4Size[p] == length ? HCV_ERR_NONE : HCV_ERR_MEM_UNAVAILABLE;
        }

        // MonoConvolve::set (.cpp:118-140)
        int set(uint32_t in, uint32_t out, const float *ir, uint64_t length, bool requestResize, bool devicePtr)
        {
            std::lock_guard<std::mutex> g(stateMutex);
            const size_t p = pair(in, out);
            // HCV_REFERENCE_QUIRKS=1: the reference holds the pair's memory through the whole of set() — upload, transforms — so the pair is
            // SILENT for the blocks processed meanwhile and what it still had to deliver is dropped (MonoConvolve.cpp:118-140, 181-183).  By
            // default the pair plays its previous IR until the new one is swapped in; on request it is cleared first, as there.
            static const bool mute_during_set = std::getenv("HCV_REFERENCE_QUIRKS") && std::atoi(std::getenv("HCV_REFERENCE_QUIRKS")) != 0;
            if (mute_during_set && mLength[p]) (void) engine->set_ir(in, out, nullptr, 0, false);
            mLength[p] = 0;
            if (requestResize)
            {
                relayout_for(length);
                tail_equal(p, length);
            }
            bool ok = true;
            if (part4Alloc[p])
            {
                // an IR longer than the tail's logical size is "loaded but silent" in the reference (:139,183)
                if (ir && length && length <= part4Size[p])
                    ok = engine->set_ir(in, out, ir, length, devicePtr);
                else
                    ok = engine->set_ir(in, out, nullptr, 0, false);
                mLength[p] = length;
            }
            else
                ok = engine->set_ir(in, out, nullptr, 0, false);
            if (!ok) set_error(engine->last_error());
            return (length && !part4Alloc[p]) ? HCV_ERR_MEM_UNAVAILABLE : (length > part4Size[p]) ? HCV_ERR_MEM_ALLOC_TOO_SMALL : HCV_ERR_NONE;
        }
    };

    // An audio-side call's hold on the matrix's engine (see Matrix::relayout_for): `ok` false = a control call is replacing the
    // engine of this (empty) object right now; the caller delivers the silence it would have computed.
    struct EngineUse
    {
        Matrix &m;
        bool ok;
        explicit EngineUse(Matrix &mm) : m(mm)
        {
            m.users.fetch_add(1, std::memory_order_seq_cst);
            ok = !m.swapping.load(std::memory_order_seq_cst);
            if (!ok) m.users.fetch_sub(1, std::memory_order_seq_cst);
            else if (!m.everProcessed.load(std::memory_order_relaxed)) m.everProcessed.store(true, std::memory_order_release);
        }
        ~EngineUse() { if (ok) m.users.fetch_sub(1, std::memory_order_release); }
        EngineUse(const EngineUse &) = delete;
        EngineUse &operator=(const EngineUse &) = delete;
    };

    Matrix *make_matrix(uint32_t numIns, uint32_t numOuts, bool parallel, uint64_t maxLength, bool zeroLatency, uint32_t A, uint32_t B, uint32_t C,
                        uint32_t D, int device, uint32_t maxBlock, std::string *err, uint32_t tailRatio = 0)
    {
        std::unique_ptr<Matrix> m(new Matrix());
        if (!m->layout.build(zeroLatency, A, B, C, D))
        {
            if (err) *err = m->layout.error;
            set_error(m->layout.error);
            return nullptr;
        }
        m->args.zero = zeroLatency; m->args.A = A; m->args.B = B; m->args.C = C; m->args.D = D;
        m->ladder = tailRatio;
        m->nin = parallel ? numOuts : numIns;
        m->nout = numOuts;
        m->diag = parallel;
        m->layout.extend_tail(maxLength, m->ratio_for(maxLength), m->pair_count(), m->boundary_bound());
        if (!m->build(numIns, numOuts, parallel, maxLength, device, maxBlock))
        {
            if (err) *err = tlsError;
            return nullptr;
        }
        return m.release();
    }

    void latency_sizes(int latency, bool &zero, uint32_t &A, uint32_t &B, uint32_t &C, uint32_t &D)
    {
        // MonoConvolve.cpp:26-31
        switch (latency)
        {
            case HCV_LATENCY_ZERO: zero = true; A = 256; B = 1024; C = 4096; D = 16384; break;
            case HCV_LATENCY_SHORT: zero = false; A = 256; B = 1024; C = 4096; D = 16384; break;
            default: zero = false; A = 1024; B = 4096; C = 16384; D = 0; break;
        }
    }
}

struct hcv_mono { std::unique_ptr<Matrix> m; bool quirks = false; };
struct hcv_ntomono { std::unique_ptr<Matrix> m; };
// One block of a sharded Convolver: its own Matrix (engine) on its own device, owning outputs [out_lo, out_hi) of inputs
// [in_lo, in_hi) of the caller's matrix
struct hcv_shard
{
    std::unique_ptr<Matrix> m;
    uint32_t in_lo = 0, in_hi = 0, out_lo = 0, out_hi = 0;
    int row = 0, col = 0, device = 0;
    // input-split layouts: this shard's partial output block [nout_local][maxBlock] on its device, twice — block b of the object
    // goes to part[b & 1], so that the row root's sum of block b - 1 never stands between this shard and its next block
    float *part[2] = { nullptr, nullptr };
    hipEvent_t evPart = nullptr;        // ... of the current block is complete
    hipEvent_t evDone[2] = { nullptr, nullptr };    // row root: the sum of the last block of that parity has read every partial block of the row group
    Engine *cur = nullptr;              // the engine the process call in progress (or the last one) runs on (Matrix::first_use)
};

struct hcv_shards
{
    std::vector<hcv_shard> s;           // rank order: row-major, the root (col 0) of a row group first
    int go = 1, gi = 1, home = 0;
    uint32_t nin = 1, nout = 1, maxBlock = 0;
    bool diag = false;
    bool peer_ok = true;                // every device in play maps every other's memory (needed by the device-pointer entry points only)
    uint64_t blocks = 0;                // device-pointer blocks so far (parity of the partial-block buffers)
    std::unique_ptr<hcv::ShardPool> pool;   // one enqueue thread per shard after the first (null: the calling thread does them all)

    hcv_shard *owner(uint32_t in, uint32_t out)
    {
        for (hcv_shard &x : s)
            if (out >= x.out_lo && out < x.out_hi && (diag || (in >= x.in_lo && in < x.in_hi))) return &x;
        return nullptr;
    }
    ~hcv_shards()
    {
        pool.reset();
        for (hcv_shard &x : s)
        {
            (void) hipSetDevice(x.device);
            if (x.m && x.m->engine) x.m->engine->synchronize();
            for (int q = 0; q < 2; q++)
            {
                if (x.part[q]) (void) hipFree(x.part[q]);
                if (x.evDone[q]) (void) hipEventDestroy(x.evDone[q]);
            }
            if (x.evPart) (void) hipEventDestroy(x.evPart);
        }
    }
};

struct hcv_convolver
{
    std::unique_ptr<Matrix> m;                 // one engine on one GPU ...
    std::unique_ptr<hcv_shards> sh;            // ... or one per device of a sharded object (then m is empty)
    std::vector<float> tmpIn, tmpOut;          // float staging of the double overloads
    hcv::RcclComm *comm = nullptr;             // one-process-per-GPU deployments: this rank's communicator of its row group
    ~hcv_convolver() { hcv::rccl_comm_destroy(comm); }
    Engine *engine0() { return sh ? sh->s[0].m->engine.get() : m->engine.get(); }
};

// ---- MonoConvolve

static hcv_mono *mono_create(uintptr_t maxLength, int zeroLatency, uint32_t A, uint32_t B, uint32_t C, uint32_t D, char *err, size_t errlen, uint32_t ladder)
{
    std::string e;
    Matrix *m = make_matrix(1, 1, false, maxLength, zeroLatency != 0, A, B, C, D, gDefaultDevice, 0, &e, ladder);
    if (!m)
    {
        if (err && errlen)
        {
            std::strncpy(err, e.c_str(), errlen - 1);
            err[errlen - 1] = 0;
        }
        return nullptr;
    }
    hcv_mono *h = new hcv_mono();
    h->m.reset(m);
    h->quirks = std::getenv("HCV_REFERENCE_QUIRKS") && std::atoi(std::getenv("HCV_REFERENCE_QUIRKS")) != 0;
    return h;
}

extern "C" hcv_mono *hcv_mono_create_custom(uintptr_t maxLength, int zeroLatency, uint32_t A, uint32_t B, uint32_t C, uint32_t D, char *err, size_t errlen)
{
    return mono_create(maxLength, zeroLatency, A, B, C, D, err, errlen, 0);        // (the caller's own partitioning: never extended)
}

extern "C" hcv_mono *hcv_mono_create(uintptr_t maxLength, int latency)
{
    bool zero;
    uint32_t A, B, C, D;
    latency_sizes(latency, zero, A, B, C, D);
    return mono_create(maxLength, zero, A, B, C, D, nullptr, 0, kLadderAuto);
}

extern "C" void hcv_mono_destroy(hcv_mono *h) { delete h; }
extern "C" void hcv_mono_set_reset_offset(hcv_mono *h, intptr_t offset) { h->m->resetOffset = offset; }
extern "C" int hcv_mono_resize(hcv_mono *h, uintptr_t length) { return h->m->resize(0, 0, length); }
extern "C" int hcv_mono_set(hcv_mono *h, const float *input, uintptr_t length, int requestResize)
{
    return h->m->set(0, 0, input, length, requestResize != 0, false);
}
extern "C" int hcv_mono_reset(hcv_mono *h)
{
    h->m->engine->reset_pair(0, 0);
    return HCV_ERR_NONE;
}

extern "C" int hcv_mono_process(hcv_mono *h, const float *in, float *temp, float *out, uintptr_t numSamples, int accumulate)
{
    (void) temp;                                            // stage summation happens on the device
    if (!h->m->active(0)) return 0;                         // MonoConvolve.cpp:183: out untouched
    EngineUse use(*h->m);
    if (!use.ok) return 0;                                  // (the engine of an empty object is being replaced: nothing loaded, out untouched)
    const float *ins[1] = { in };
    float *outs[1] = { out };
    // HCV_REFERENCE_QUIRKS=1 (read when the object is made): MonoConvolve.cpp:195-197 as the reference has it.  In a zero-latency layout of
    // fewer than four FFT sizes mPart1 is null, so the first FFT stage is processed with `accumulate || mPart1` = the caller's flag: called
    // without accumulate it WRITES `out` and the time-domain head's output, written a line before, is lost — provided that stage holds
    // partitions (a PartitionedConvolve without an IR returns false and touches nothing).  The default sums every stage.
    if (h->quirks)
        h->m->engine->set_drop_head(!accumulate && h->m->layout.zeroLatency && h->m->layout.sizes.size() < 4 && h->m->engine->stage_partitions(0, 0, 0) > 0);
    if (!h->m->engine->process(ins, outs, 1, 1, numSamples, accumulate != 0))
    {
        set_error(h->m->engine->last_error());
        return -1;
    }
    return 1;
}

// ---- NToMonoConvolve

extern "C" hcv_ntomono *hcv_ntomono_create(uint32_t inChans, uintptr_t maxLength, int latency)
{
    bool zero;
    uint32_t A, B, C, D;
    latency_sizes(latency, zero, A, B, C, D);
    Matrix *m = make_matrix(inChans, 1, false, maxLength, zero, A, B, C, D, gDefaultDevice, 0, nullptr, kLadderAuto);
    if (!m) return nullptr;
    hcv_ntomono *h = new hcv_ntomono();
    h->m.reset(m);
    return h;
}

extern "C" void hcv_ntomono_destroy(hcv_ntomono *h) { delete h; }

extern "C" int hcv_ntomono_resize(hcv_ntomono *h, uint32_t inChan, uintptr_t length)
{
    return inChan < h->m->nin ? h->m->resize(inChan, 0, length) : HCV_ERR_IN_CHAN_OUT_OF_RANGE;
}

extern "C" int hcv_ntomono_set(hcv_ntomono *h, uint32_t inChan, const float *input, uintptr_t length, int resize)
{
    return inChan < h->m->nin ? h->m->set(inChan, 0, input, length, resize != 0, false) : HCV_ERR_IN_CHAN_OUT_OF_RANGE;
}

extern "C" int hcv_ntomono_reset(hcv_ntomono *h, uint32_t inChan)
{
    if (inChan >= h->m->nin) return HCV_ERR_IN_CHAN_OUT_OF_RANGE;
    h->m->engine->reset_pair(inChan, 0);
    return HCV_ERR_NONE;
}

extern "C" int hcv_ntomono_process(hcv_ntomono *h, const float *const *ins, float *out, float *temp, size_t numSamples, size_t activeInChans)
{
    (void) temp;
    float *outs[1] = { out };
    const uint32_t act = (uint32_t) std::min<size_t>(activeInChans, h->m->nin);
    EngineUse use(*h->m);
    if (!use.ok)
    {
        std::memset(out, 0, sizeof(float) * numSamples);    // (an empty object whose engine is being replaced: the sum of no inputs, .cpp:39)
        return 0;
    }
    if (!h->m->engine->process(ins, outs, act, 1, numSamples, false))   // zero + accumulate == overwrite with the sum (.cpp:39-42)
    {
        set_error(h->m->engine->last_error());
        return -1;
    }
    return 0;
}

// ---- Convolver

static hcv_convolver *wrap(Matrix *m)
{
    if (!m) return nullptr;
    hcv_convolver *h = new hcv_convolver();
    h->m.reset(m);
    return h;
}

// ---- sharded Convolver: ONE host object, one engine per listed device.  Output rows are split over the devices
// first (no exchange); with fewer rows than devices the inputs are split too and a row group's partial blocks are summed on
// the group's first device — the per-output sum of NToMonoConvolve.cpp:39-42 taken across GPUs.

static void split_range(uint32_t n, uint32_t parts, uint32_t index, uint32_t &lo, uint32_t &hi)
{
    const uint32_t base = n / parts, rem = n % parts;
    lo = index * base + std::min(index, rem);
    hi = lo + base + (index < rem ? 1 : 0);
}

static hcv_convolver *make_sharded(uint32_t numIns, uint32_t numOuts, bool parallel, uint64_t maxLength, bool zeroLatency, uint32_t A, uint32_t B,
                                   uint32_t C, uint32_t D, const int *devices, int n, uint32_t maxBlock, uint32_t ladder = 0)
{
    if (!devices || n < 1 || n > 64 || !numOuts)
    {
        set_error("sharded Convolver: needs 1 .. 64 devices and at least one output");
        return nullptr;
    }
    const int count = hcv_device_count();
    for (int k = 0; k < n; k++)
        if (devices[k] < 0 || devices[k] >= count)
        {
            set_error("sharded Convolver: device index out of range");
            return nullptr;
        }
    numIns = parallel ? numOuts : (numIns < 1 ? 1 : numIns);
    // rows first; what is left of the devices splits the inputs (never more column groups than inputs, none in parallel mode)
    uint32_t go = std::min<uint32_t>((uint32_t) n, numOuts);
    uint32_t gi = parallel ? 1 : std::min<uint32_t>(std::max<uint32_t>(1, (uint32_t) n / go), numIns);
    if (gi > (uint32_t) hcv::kMaxParts) gi = hcv::kMaxParts;
    std::unique_ptr<hcv_convolver> h(new hcv_convolver());
    h->sh.reset(new hcv_shards());
    hcv_shards &sh = *h->sh;
    sh.go = (int) go;
    sh.gi = (int) gi;
    sh.nin = numIns;
    sh.nout = numOuts;
    sh.diag = parallel;
    sh.home = devices[0];
    int prev = -1;
    (void) hipGetDevice(&prev);
    bool ok = true;
    for (uint32_t r = 0; r < go && ok; r++)
        for (uint32_t c = 0; c < gi && ok; c++)
        {
            sh.s.emplace_back();
            hcv_shard &x = sh.s.back();
            x.row = (int) r;
            x.col = (int) c;
            x.device = devices[r * gi + c];
            split_range(numOuts, go, r, x.out_lo, x.out_hi);
            if (parallel) { x.in_lo = x.out_lo; x.in_hi = x.out_hi; }
            else split_range(numIns, gi, c, x.in_lo, x.in_hi);
            x.m.reset(make_matrix(x.in_hi - x.in_lo, x.out_hi - x.out_lo, parallel, maxLength, zeroLatency, A, B, C, D, x.device, maxBlock, nullptr, ladder));
            if (!x.m) { ok = false; break; }
            x.m->pristineOnly = true;        // (the shard pool's jobs hold engine pointers across a block: re-laid before the first block only)
            sh.maxBlock = x.m->engine->max_block();
            ok = hipSetDevice(x.device) == hipSuccess;
            // every device reads the caller's buffers (on the home device) and the row root reads its group's partial blocks
            // directly, over xGMI: peer access both ways between the devices in play
            // (the host-pointer path stages through each engine's own pinned buffers and needs none of this: a box without
            // peer-to-peer keeps the object, and only the device-pointer entry points refuse)
            for (int k = 0; k < n && ok; k++)
                if (devices[k] != x.device)
                {
                    const hipError_t e = hipDeviceEnablePeerAccess(devices[k], 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) sh.peer_ok = false;
                    (void) hipGetLastError();
                }
            if (ok && gi > 1)
            {
                const size_t bytes = sizeof(float) * (size_t) (x.out_hi - x.out_lo) * sh.maxBlock;
                ok = hipMalloc(&x.part[0], bytes) == hipSuccess && hipMalloc(&x.part[1], bytes) == hipSuccess &&
                     hipEventCreateWithFlags(&x.evPart, hipEventDisableTiming) == hipSuccess;
                for (int q = 0; q < 2 && ok; q++)
                    ok = hipEventCreateWithFlags(&x.evDone[q], hipEventDisableTiming) == hipSuccess &&
                         hipEventRecord(x.evDone[q], x.m->engine->main_stream()) == hipSuccess;
            }
            if (!ok) set_error("sharded Convolver: device set-up failed (peer access / allocation)");
        }
    // Enqueue threads (HCV_SHARD_THREADS = 0 / 1 forces the choice): by default when the shards really are on several devices —
    // each device has its own queues and the HIP calls of the shards run side by side; engines sharing ONE device (the test
    // box's stand-in for a node) contend for that device's runtime locks and gain little.
    if (ok && sh.s.size() > 1 && sh.s.size() <= 64)
    {
        std::vector<int> devs;
        bool distinct = false;
        for (const hcv_shard &x : sh.s)
        {
            devs.push_back(x.device);
            distinct = distinct || x.device != sh.s[0].device;
        }
        const char *env = std::getenv("HCV_SHARD_THREADS");
        if (env ? std::atoi(env) != 0 : distinct) sh.pool.reset(new hcv::ShardPool((int) devs.size(), devs.data()));
    }
    if (prev >= 0) (void) hipSetDevice(prev);
    if (!ok) return nullptr;
    return h.release();
}

extern "C" hcv_convolver *hcv_convolver_create_sharded(uint32_t numIns, uint32_t numOuts, int parallel, uintptr_t maxLength, int zeroLatency, uint32_t A,
                                                       uint32_t B, uint32_t C, uint32_t D, const int *devices, int numDevices, uint32_t maxBlock)
{
    return make_sharded(numIns, numOuts, parallel != 0, maxLength, zeroLatency != 0, A, B, C, D, devices, numDevices, maxBlock);
}

// HCV_DEVICES="0,1,2,3": objects made by the reference-shaped constructors (HISSTools::Convolver, hcv_convolver_create / _parallel)
// are sharded over these devices, so a C++ caller that recompiles unchanged gets every GPU it lists
static bool env_devices(std::vector<int> &out)
{
    const char *env = std::getenv("HCV_DEVICES");
    if (!env || !*env) return false;
    for (const char *p = env; *p;)
    {
        char *end = nullptr;
        const long v = std::strtol(p, &end, 10);
        if (end == p) break;
        out.push_back((int) v);
        p = (*end == ',') ? end + 1 : end;
    }
    return out.size() > 1;
}

extern "C" hcv_convolver *hcv_convolver_create_on(uint32_t numIns, uint32_t numOuts, int latency, int device, uint32_t maxBlock)
{
    bool zero;
    uint32_t A, B, C, D;
    latency_sizes(latency, zero, A, B, C, D);
    numIns = numIns < 1 ? 1 : numIns;                       // Convolver.cpp:8
    std::vector<int> devs;
    if (device < 0 && env_devices(devs))
    {
        if (hcv_convolver *h = make_sharded(numIns, numOuts, false, 16384, zero, A, B, C, D, devs.data(), (int) devs.size(), maxBlock, kLadderAuto)) return h;
        std::fprintf(stderr, "hisstools_amd: HCV_DEVICES could not be honoured (%s); using one device\n", tlsError.c_str());
    }
    return wrap(make_matrix(numIns, numOuts, false, 16384, zero, A, B, C, D, device < 0 ? gDefaultDevice : device, maxBlock, nullptr, kLadderAuto));
}

extern "C" hcv_convolver *hcv_convolver_create(uint32_t numIns, uint32_t numOuts, int latency)
{
    return hcv_convolver_create_on(numIns, numOuts, latency, -1, 0);
}

extern "C" hcv_convolver *hcv_convolver_create_parallel(uint32_t numIO, int latency)
{
    bool zero;
    uint32_t A, B, C, D;
    latency_sizes(latency, zero, A, B, C, D);
    numIO = numIO < 1 ? 1 : numIO;                          // Convolver.cpp:27
    std::vector<int> devs;
    if (env_devices(devs))
    {
        if (hcv_convolver *h = make_sharded(numIO, numIO, true, 16384, zero, A, B, C, D, devs.data(), (int) devs.size(), 0, kLadderAuto)) return h;
        std::fprintf(stderr, "hisstools_amd: HCV_DEVICES could not be honoured (%s); using one device\n", tlsError.c_str());
    }
    return wrap(make_matrix(numIO, numIO, true, 16384, zero, A, B, C, D, gDefaultDevice, 0, nullptr, kLadderAuto));
}

extern "C" hcv_convolver *hcv_convolver_create_custom(uint32_t numIns, uint32_t numOuts, int parallel, uintptr_t maxLength, int zeroLatency, uint32_t A,
                                                      uint32_t B, uint32_t C, uint32_t D, int device, uint32_t maxBlock)
{
    numIns = numIns < 1 ? 1 : numIns;
    return wrap(make_matrix(numIns, numOuts, parallel != 0, maxLength, zeroLatency != 0, A, B, C, D, device < 0 ? gDefaultDevice : device, maxBlock,
                            nullptr));
}

extern "C" hcv_convolver *hcv_convolver_create_extended(uint32_t numIns, uint32_t numOuts, int parallel, uintptr_t maxLength, int zeroLatency, uint32_t A,
                                                        uint32_t B, uint32_t C, uint32_t D, int device, uint32_t maxBlock, uint32_t tailRatio)
{
    numIns = numIns < 1 ? 1 : numIns;
    if (tailRatio != 0 && tailRatio != 2 && tailRatio != 4 && tailRatio != 8)
    {
        set_error("tailRatio must be 0, 2, 4 or 8");
        return nullptr;
    }
    return wrap(make_matrix(numIns, numOuts, parallel != 0, maxLength, zeroLatency != 0, A, B, C, D, device < 0 ? gDefaultDevice : device, maxBlock,
                            nullptr, tailRatio));
}

extern "C" void hcv_convolver_destroy(hcv_convolver *h) { delete h; }

// parallel mode: "inChan -= outChan" in unsigned arithmetic, then the 1-input NToMonoConvolve range check (Convolver.cpp:92,106,118)
static bool conv_in_ok(bool diag, uint32_t nin, uint32_t &inChan, uint32_t outChan)
{
    if (diag)
    {
        inChan -= outChan;
        return inChan < 1;
    }
    return inChan < nin;
}
static bool conv_in_ok(const Matrix &m, uint32_t &inChan, uint32_t outChan) { return conv_in_ok(m.diag, m.nin, inChan, outChan); }

// the pair's place in a sharded object: range checks on the caller's matrix (same codes as the single-device object), then the
// owning shard and the pair's indices inside it.  Returns an error code, or -1 with `x` set.
static int shard_pair(hcv_convolver *h, uint32_t &inChan, uint32_t &outChan, int outRangeCode, hcv_shard *&x)
{
    hcv_shards &sh = *h->sh;
    const bool inOk = conv_in_ok(sh.diag, sh.nin, inChan, outChan);
    if (outChan >= sh.nout) return outRangeCode;
    if (!inOk) return HCV_ERR_IN_CHAN_OUT_OF_RANGE;
    if (sh.diag) inChan = outChan;
    x = sh.owner(inChan, outChan);
    if (!x) return HCV_ERR_IN_CHAN_OUT_OF_RANGE;
    outChan -= x->out_lo;
    inChan = sh.diag ? outChan : inChan - x->in_lo;
    return -1;
}

extern "C" int hcv_convolver_set_f32(hcv_convolver *h, uint32_t inChan, uint32_t outChan, const float *input, uintptr_t length, int resize)
{
    if (h->sh)
    {
        hcv_shard *x = nullptr;
        const int rc = shard_pair(h, inChan, outChan, HCV_ERR_OUT_CHAN_OUT_OF_RANGE, x);
        return rc >= 0 ? rc : x->m->set(inChan, outChan, input, length, resize != 0, false);
    }
    Matrix &m = *h->m;
    const bool inOk = conv_in_ok(m, inChan, outChan);
    if (outChan >= m.nout) return HCV_ERR_OUT_CHAN_OUT_OF_RANGE;
    if (!inOk) return HCV_ERR_IN_CHAN_OUT_OF_RANGE;
    return m.set(m.diag ? outChan : inChan, outChan, input, length, resize != 0, false);
}

extern "C" int hcv_convolver_set_f32_dev(hcv_convolver *h, uint32_t inChan, uint32_t outChan, const float *input_dev, uintptr_t length, int resize)
{
    if (h->sh)
    {
        // (the IR lies on the home device; a shard elsewhere reads it over the peer mapping)
        hcv_shard *x = nullptr;
        const int rc = shard_pair(h, inChan, outChan, HCV_ERR_OUT_CHAN_OUT_OF_RANGE, x);
        if (rc < 0 && !h->sh->peer_ok && x->device != h->sh->home)
        {
            set_error("sharded Convolver: no peer access between the devices; load this pair through host pointers (hcv_convolver_set_f32)");
            return HCV_ERR_MEM_UNAVAILABLE;
        }
        return rc >= 0 ? rc : x->m->set(inChan, outChan, input_dev, length, resize != 0, true);
    }
    Matrix &m = *h->m;
    const bool inOk = conv_in_ok(m, inChan, outChan);
    if (outChan >= m.nout) return HCV_ERR_OUT_CHAN_OUT_OF_RANGE;
    if (!inOk) return HCV_ERR_IN_CHAN_OUT_OF_RANGE;
    return m.set(m.diag ? outChan : inChan, outChan, input_dev, length, resize != 0, true);
}

extern "C" int hcv_convolver_set_f64(hcv_convolver *h, uint32_t inChan, uint32_t outChan, const double *input, uintptr_t length, int resize)
{
    std::vector<float> f(length);
    for (uintptr_t i = 0; i < length; i++) f[i] = static_cast<float>(input[i]);
    return hcv_convolver_set_f32(h, inChan, outChan, length ? f.data() : nullptr, length, resize);
}

extern "C" void hcv_convolver_clear_chan(hcv_convolver *h, uint32_t inChan, uint32_t outChan, int resize)
{
    hcv_convolver_set_f32(h, inChan, outChan, nullptr, 0, resize);
}

extern "C" void hcv_convolver_clear(hcv_convolver *h, int resize)
{
    struct { uint32_t nin, nout; bool diag; } m = { h->sh ? h->sh->nin : h->m->nin, h->sh ? h->sh->nout : h->m->nout, h->sh ? h->sh->diag : h->m->diag };
    for (uint32_t o = 0; o < m.nout; o++)
    {
        if (!m.diag)
            for (uint32_t i = 0; i < m.nin; i++) hcv_convolver_clear_chan(h, i, o, resize);
        else
            hcv_convolver_clear_chan(h, o, o, resize);
    }
}

extern "C" int hcv_convolver_reset_chan(hcv_convolver *h, uint32_t inChan, uint32_t outChan)
{
    if (h->sh)
    {
        hcv_shard *x = nullptr;
        const int rc = shard_pair(h, inChan, outChan, HCV_ERR_OUT_CHAN_OUT_OF_RANGE, x);
        if (rc >= 0) return rc;
        x->m->engine->reset_pair(inChan, outChan);
        return HCV_ERR_NONE;
    }
    Matrix &m = *h->m;
    const bool inOk = conv_in_ok(m, inChan, outChan);
    if (outChan >= m.nout) return HCV_ERR_OUT_CHAN_OUT_OF_RANGE;
    if (!inOk) return HCV_ERR_IN_CHAN_OUT_OF_RANGE;
    // (an empty object whose engine is being replaced, Matrix::relayout_for, has nothing to restart: the hold says so)
    EngineUse use(m);
    if (use.ok) m.engine->reset_pair(m.diag ? outChan : inChan, outChan);
    return HCV_ERR_NONE;
}

extern "C" void hcv_convolver_reset(hcv_convolver *h)
{
    if (h->sh)
        for (hcv_shard &x : h->sh->s) x.m->engine->reset_all();
    else
    {
        EngineUse use(*h->m);
        if (use.ok) h->m->engine->reset_all();
    }
}

extern "C" int hcv_convolver_resize(hcv_convolver *h, uint32_t inChan, uint32_t outChan, uintptr_t length)
{
    if (h->sh)
    {
        hcv_shard *x = nullptr;
        const int rc = shard_pair(h, inChan, outChan, HCV_ERR_IN_CHAN_OUT_OF_RANGE /* sic, as below */, x);
        return rc >= 0 ? rc : x->m->resize(inChan, outChan, length);
    }
    Matrix &m = *h->m;
    const bool inOk = conv_in_ok(m, inChan, outChan);
    if (outChan >= m.nout) return HCV_ERR_IN_CHAN_OUT_OF_RANGE;          // sic: Convolver.cpp:108-111 returns the IN code here
    if (!inOk) return HCV_ERR_IN_CHAN_OUT_OF_RANGE;
    return m.resize(m.diag ? outChan : inChan, outChan, length);
}

// One call of a sharded object, host pointers: every shard's block is begun (inputs staged, kernels and download enqueued on
// its own device) before the first is waited for, so the devices run side by side; an input-split row group's partial blocks
// are added up as they are delivered (root first, which overwrites).
namespace
{
    // one block of a sharded object's host-pointer call, as the shard workers see it
    struct HostJob
    {
        hcv_shards *sh;
        const float *const *ins;
        float *const *outs;
        size_t pos;
        uint32_t B, ni, no;
        std::string err[64];
    };

    void shard_active(const hcv_shards &sh, const hcv_shard &x, uint32_t ni, uint32_t no, uint32_t &a_ni, uint32_t &a_no)
    {
        a_no = no > x.out_lo ? std::min(no, x.out_hi) - x.out_lo : 0;
        a_ni = sh.diag ? a_no : (ni > x.in_lo ? std::min(ni, x.in_hi) - x.in_lo : 0);
    }

    // phase 1: stage the shard's inputs, enqueue its block and the download on its own device
    bool host_begin_job(void *ctx, int k)
    {
        HostJob &j = *static_cast<HostJob *>(ctx);
        hcv_shard &x = j.sh->s[(size_t) k];
        uint32_t a_ni, a_no;
        shard_active(*j.sh, x, j.ni, j.no, a_ni, a_no);
        if (!a_no || (x.col > 0 && !a_ni)) return true;
        const float *ip[64];
        std::vector<const float *> big;
        const float **rows = ip;
        if (a_ni > 64) { big.resize(a_ni); rows = big.data(); }
        for (uint32_t i = 0; i < a_ni; i++) rows[i] = j.ins[x.in_lo + i] + j.pos;
        if (x.cur->process_begin(rows, a_ni, a_no, j.B)) return true;
        if (k < 64) j.err[k] = x.cur->last_error();
        return false;
    }

    // phase 2, run for the ROOT of every row group: wait for and deliver the group's blocks in column order (the root's block
    // overwrites the caller's rows, the others are added: NToMonoConvolve.cpp:39-42 across the shards)
    bool host_end_job(void *ctx, int k)
    {
        HostJob &j = *static_cast<HostJob *>(ctx);
        hcv_shards &sh = *j.sh;
        for (int c = 0; c < sh.gi; c++)
        {
            hcv_shard &x = sh.s[(size_t) k + (size_t) c];
            uint32_t a_ni, a_no;
            shard_active(sh, x, j.ni, j.no, a_ni, a_no);
            if (!a_no || (x.col > 0 && !a_ni)) continue;
            float *op[64];
            std::vector<float *> big;
            float **rows = op;
            if (a_no > 64) { big.resize(a_no); rows = big.data(); }
            for (uint32_t o = 0; o < a_no; o++) rows[o] = j.outs[x.out_lo + o] + j.pos;
            if (!x.cur->process_end(rows, a_no, j.B, /* accumulate */ x.col > 0))
            {
                if (k < 64) j.err[k] = x.cur->last_error();
                return false;
            }
        }
        return true;
    }

    bool run_shards(hcv_shards &sh, hcv::ShardPool::Fn fn, void *ctx, uint64_t mask)
    {
        if (sh.pool) return sh.pool->run(fn, ctx, mask);
        bool ok = true;
        for (size_t k = 0; k < sh.s.size() && ok; k++)
            if (mask >> k & 1) ok = fn(ctx, (int) k);
        return ok;
    }
}

// One call of a sharded object, host pointers: every shard's block is begun (inputs staged, kernels and download enqueued on
// its own device) before the first is waited for, so the devices run side by side; an input-split row group's partial blocks
// are added up as they are delivered (root first, which overwrites).  With enqueue threads (hcv_shard_pool.h) the shards' begin
// halves run side by side on the host too, and so do the row groups' end halves.
static int sharded_process_host(hcv_convolver *h, const float *const *ins, float *const *outs, size_t numIns, size_t numOuts, size_t numSamples)
{
    hcv_shards &sh = *h->sh;
    for (hcv_shard &x : sh.s) x.cur = x.m->first_use();         // (from here on the shards keep their stage lists)
    HostJob j;
    j.sh = &sh;
    j.ins = ins;
    j.outs = outs;
    j.no = (uint32_t) std::min<size_t>(numOuts, sh.nout);
    j.ni = sh.diag ? j.no : (uint32_t) std::min<size_t>(numIns, sh.nin);
    uint64_t all = 0, roots = 0;
    for (size_t k = 0; k < sh.s.size() && k < 64; k++)
    {
        all |= uint64_t(1) << k;
        if (sh.s[k].col == 0) roots |= uint64_t(1) << k;
    }
    for (size_t pos = 0; pos < numSamples; pos += sh.maxBlock)
    {
        j.pos = pos;
        j.B = (uint32_t) std::min<size_t>(sh.maxBlock, numSamples - pos);
        if (!run_shards(sh, host_begin_job, &j, all) || !run_shards(sh, host_end_job, &j, roots))
        {
            for (const std::string &e : j.err)
                if (!e.empty()) { set_error(e); break; }
            return -1;
        }
    }
    return 0;
}

// ---- registered host memory: callers that keep their channel buffers for a while (plug-in hosts) can pin and map them once;
// process() calls whose rows all lie in registered memory, evenly spaced, then run on the caller's memory in place — the
// kernels read the inputs and write the outputs over PCIe — instead of going through the staging copies (two host memcpys of
// the whole block and two DMA transfers per call: 0.55 of the 1.1 ms a 64x64 engine's 8192-sample host call takes).

namespace
{
    struct HostRegion { const char *base; size_t bytes; char *dev; };
    std::mutex gRegMutex;
    std::vector<HostRegion> gRegions;
    std::atomic<int> gRegionCount { 0 };

    // the device address of a [rows][n] block given as row pointers, if it is one evenly spaced block inside a registered region
    bool mapped_block(const float *const *rows, size_t nrows, size_t n, const float **dev, int64_t *stride)
    {
        if (!nrows || !n || !rows[0]) return false;
        const ptrdiff_t st = nrows > 1 ? rows[1] - rows[0] : (ptrdiff_t) n;
        if (st < (ptrdiff_t) n) return false;                           // rows overlap or run backwards: not a plain block
        for (size_t r = 2; r < nrows; r++)
            if (rows[r] - rows[r - 1] != st) return false;
        const char *lo = reinterpret_cast<const char *>(rows[0]);
        const char *hi = reinterpret_cast<const char *>(rows[nrows - 1] + n);
        std::lock_guard<std::mutex> g(gRegMutex);
        for (const HostRegion &r : gRegions)
            if (lo >= r.base && hi <= r.base + r.bytes)
            {
                *dev = reinterpret_cast<const float *>(r.dev + (lo - r.base));
                *stride = (int64_t) st;
                return true;
            }
        return false;
    }
}

extern "C" int hcv_host_register(void *ptr, size_t bytes)
{
    if (!ptr || !bytes) return -1;
    if (gDefaultDevice >= 0) (void) hipSetDevice(gDefaultDevice);
    if (hipHostRegister(ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess)
    {
        set_error(std::string("hipHostRegister: ") + hipGetErrorString(hipGetLastError()));
        return -1;
    }
    void *dev = nullptr;
    if (hipHostGetDevicePointer(&dev, ptr, 0) != hipSuccess || !dev)
    {
        (void) hipGetLastError();
        (void) hipHostUnregister(ptr);
        set_error("hipHostGetDevicePointer failed for the registered block");
        return -1;
    }
    std::lock_guard<std::mutex> g(gRegMutex);
    gRegions.push_back({ static_cast<const char *>(ptr), bytes, static_cast<char *>(dev) });
    gRegionCount = (int) gRegions.size();
    return 0;
}

extern "C" int hcv_host_unregister(void *ptr)
{
    {
        std::lock_guard<std::mutex> g(gRegMutex);
        auto it = std::find_if(gRegions.begin(), gRegions.end(), [&](const HostRegion &r) { return r.base == ptr; });
        if (it == gRegions.end()) return -1;
        gRegions.erase(it);
        gRegionCount = (int) gRegions.size();
    }
    if (hipHostUnregister(ptr) != hipSuccess)
    {
        (void) hipGetLastError();
        return -1;
    }
    return 0;
}

extern "C" int hcv_convolver_process_f32(hcv_convolver *h, const float *const *ins, float **outs, size_t numIns, size_t numOuts, size_t numSamples)
{
    if (h->sh) return sharded_process_host(h, ins, outs, numIns, numOuts, numSamples);
    Matrix &m = *h->m;
    const uint32_t no = (uint32_t) std::min<size_t>(numOuts, m.nout);
    const uint32_t ni = m.diag ? no : (uint32_t) std::min<size_t>(numIns, m.nin);
    EngineUse use(m);
    if (!use.ok)
    {
        // (an empty object whose engine is being replaced, Matrix::relayout_for: the silence it would have computed)
        for (uint32_t o = 0; o < no; o++) std::memset(outs[o], 0, sizeof(float) * numSamples);
        return 0;
    }
    if (gRegionCount.load(std::memory_order_relaxed) > 0 && no && numSamples)
    {
        const float *din = nullptr, *dout = nullptr;
        int64_t is = 0, os = 0;
        const uint32_t rows_in = m.diag ? no : ni;
        if ((rows_in == 0 || mapped_block(ins, rows_in, numSamples, &din, &is)) && mapped_block(outs, no, numSamples, &dout, &os))
        {
            if (!m.engine->process_pinned(rows_in ? ins[0] : nullptr, din, rows_in ? is : (int64_t) numSamples, outs[0], const_cast<float *>(dout), os, ni, no,
                                          numSamples))
            {
                set_error(m.engine->last_error());
                return -1;
            }
            return 0;
        }
    }
    if (!m.engine->process(ins, outs, ni, no, numSamples, false))
    {
        set_error(m.engine->last_error());
        return -1;
    }
    return 0;
}

extern "C" int hcv_convolver_process_f64(hcv_convolver *h, const double *const *ins, double **outs, size_t numIns, size_t numOuts, size_t numSamples)
{
    struct { uint32_t nin, nout; bool diag; } m = { h->sh ? h->sh->nin : h->m->nin, h->sh ? h->sh->nout : h->m->nout, h->sh ? h->sh->diag : h->m->diag };
    const uint32_t no = (uint32_t) std::min<size_t>(numOuts, m.nout);
    const uint32_t ni = m.diag ? no : (uint32_t) std::min<size_t>(numIns, m.nin);
    h->tmpIn.resize((size_t) std::max<uint32_t>(ni, 1) * numSamples);
    h->tmpOut.resize((size_t) std::max<uint32_t>(no, 1) * numSamples);
    std::vector<const float *> ip(std::max<uint32_t>(ni, 1));
    std::vector<float *> op(std::max<uint32_t>(no, 1));
    for (uint32_t i = 0; i < ni; i++)
    {
        float *dst = h->tmpIn.data() + (size_t) i * numSamples;
        for (size_t j = 0; j < numSamples; j++) dst[j] = static_cast<float>(ins[i][j]);
        ip[i] = dst;
    }
    for (uint32_t o = 0; o < no; o++) op[o] = h->tmpOut.data() + (size_t) o * numSamples;
    if (h->sh)
    {
        if (sharded_process_host(h, ip.data(), op.data(), ni, no, numSamples) != 0) return -1;
    }
    else
    {
        EngineUse use(*h->m);
        if (!use.ok)
            std::fill(h->tmpOut.begin(), h->tmpOut.end(), 0.f);        // (an empty object whose engine is being replaced: silence)
        else if (!h->m->engine->process(ip.data(), op.data(), ni, no, numSamples, false))
        {
            set_error(h->m->engine->last_error());
            return -1;
        }
    }
    for (uint32_t o = 0; o < no; o++)
        for (size_t j = 0; j < numSamples; j++) outs[o][j] = op[o][j];
    return 0;
}

// One call of a sharded object, device pointers (both buffers on the home device).  Every shard's kernels read their input rows
// and write their output rows in the caller's buffers directly (peer access over xGMI for the shards on other devices): no
// staging copies.  Input-split layouts: each shard of a row group emits its partial block into a buffer of its own, the
// group's root waits for them (events across devices) and one kernel on the root's stream writes their sum to the caller's rows.
namespace
{
    struct DevJob
    {
        hcv_shards *sh;
        const float *ins;
        float *outs;
        size_t in_stride, out_stride, pos, B;
        uint32_t ni, no;
        int par;                         // parity of the partial-block buffers for this block
    };

    // phase 1: the shard's block on its own device — into the caller's rows (row sharding), or into its partial block
    bool dev_block_job(void *ctx, int k)
    {
        DevJob &j = *static_cast<DevJob *>(ctx);
        hcv_shards &sh = *j.sh;
        hcv_shard &x = sh.s[(size_t) k];
        uint32_t a_ni, a_no;
        shard_active(sh, x, j.ni, j.no, a_ni, a_no);
        if (!a_no) return true;
        Engine &e = *x.cur;
        const float *src = j.ins + (size_t) x.in_lo * j.in_stride + j.pos;
        if (sh.gi == 1)
            return e.process_dev(src, (int64_t) j.in_stride, j.outs + (size_t) x.out_lo * j.out_stride + j.pos, (int64_t) j.out_stride, a_ni, a_no, j.B, false);
        hcv_shard &root = sh.s[(size_t) k - (size_t) x.col];
        // the root's sum of the block two back read this buffer: every writer of this block follows it (`after`: a streamed
        // whole-hop block writes from its last stage's stream, not from the main stream)
        if (!e.process_dev(src, (int64_t) j.in_stride, x.part[j.par], (int64_t) sh.maxBlock, a_ni, a_no, j.B, false, root.evDone[j.par])) return false;
        int prev = -1;
        (void) hipGetDevice(&prev);
        bool ok = (prev == x.device || hipSetDevice(x.device) == hipSuccess) && hipEventRecord(x.evPart, e.main_stream()) == hipSuccess;
        if (prev >= 0 && prev != x.device) (void) hipSetDevice(prev);
        return ok;
    }

    // phase 2, for the root of every row group: wait for the group's partial blocks (events across devices) and write their sum
    // to the caller's rows with one kernel on the root's stream
    bool dev_sum_job(void *ctx, int k)
    {
        DevJob &j = *static_cast<DevJob *>(ctx);
        hcv_shards &sh = *j.sh;
        hcv_shard &root = sh.s[(size_t) k];
        uint32_t a_ni, a_no;
        shard_active(sh, root, j.ni, j.no, a_ni, a_no);
        if (!a_no) return true;
        hipStream_t rs = root.cur->main_stream();
        int prev = -1;
        (void) hipGetDevice(&prev);
        bool ok = prev == root.device || hipSetDevice(root.device) == hipSuccess;
        hcv::PartSources ps;
        ps.count = sh.gi;
        for (int c = 0; c < sh.gi && ok; c++)
        {
            ps.part[c] = sh.s[(size_t) k + (size_t) c].part[j.par];
            if (c) ok = hipStreamWaitEvent(rs, sh.s[(size_t) k + (size_t) c].evPart, 0) == hipSuccess;
        }
        ok = ok && hcv::launch_sum_parts(ps, (long long) sh.maxBlock, (int) j.B, (int) a_no, j.outs + (size_t) root.out_lo * j.out_stride + j.pos,
                                         (long long) j.out_stride, rs) == hipSuccess;
        ok = ok && hipEventRecord(root.evDone[j.par], rs) == hipSuccess;
        if (prev >= 0 && prev != root.device) (void) hipSetDevice(prev);
        return ok;
    }
}

// One call of a sharded object, device pointers (both buffers on the home device).  Every shard's kernels read their input rows
// and write their output rows in the caller's buffers directly (peer access over xGMI for the shards on other devices): no
// staging copies.  Input-split layouts: each shard of a row group emits its partial block into a buffer of its own, the
// group's root waits for them (events across devices) and one kernel on the root's stream writes their sum to the caller's rows.
static int sharded_process_dev(hcv_convolver *h, const float *ins_dev, size_t in_stride, float *outs_dev, size_t out_stride, size_t numIns, size_t numOuts,
                               size_t numSamples, int sync)
{
    hcv_shards &sh = *h->sh;
    if (!sh.peer_ok)
    {
        set_error("sharded Convolver: no peer access between the devices; use the host-pointer entry point (hcv_convolver_process_f32)");
        return -1;
    }
    for (hcv_shard &x : sh.s) x.cur = x.m->first_use();
    DevJob j;
    j.sh = &sh;
    j.ins = ins_dev;
    j.outs = outs_dev;
    j.in_stride = in_stride;
    j.out_stride = out_stride;
    j.no = (uint32_t) std::min<size_t>(numOuts, sh.nout);
    j.ni = sh.diag ? j.no : (uint32_t) std::min<size_t>(numIns, sh.nin);
    uint64_t all = 0, roots = 0;
    for (size_t k = 0; k < sh.s.size() && k < 64; k++)
    {
        all |= uint64_t(1) << k;
        if (sh.s[k].col == 0) roots |= uint64_t(1) << k;
    }
    int prev = -1;
    (void) hipGetDevice(&prev);
    bool ok = true;
    for (size_t pos = 0; pos < numSamples && ok; pos += sh.maxBlock)
    {
        j.pos = pos;
        j.B = std::min<size_t>(sh.maxBlock, numSamples - pos);
        j.par = (int) (sh.blocks++ & 1);
        ok = run_shards(sh, dev_block_job, &j, all);
        // (the sums wait for events the block jobs recorded: phase 2 starts when every shard's phase 1 has returned)
        if (ok && sh.gi > 1) ok = run_shards(sh, dev_sum_job, &j, roots);
    }
    if (prev >= 0) (void) hipSetDevice(prev);
    if (!ok)
    {
        (void) hipGetLastError();
        set_error("sharded Convolver: a device call failed in process_dev");
        return -1;
    }
    return sync ? hcv_convolver_synchronize(h) : 0;
}

extern "C" int hcv_convolver_process_f32_dev(hcv_convolver *h, const float *ins_dev, size_t in_stride, float *outs_dev, size_t out_stride, size_t numIns,
                                             size_t numOuts, size_t numSamples, int sync)
{
    if (h->sh) return sharded_process_dev(h, ins_dev, in_stride, outs_dev, out_stride, numIns, numOuts, numSamples, sync);
    Matrix &m = *h->m;
    const uint32_t no = (uint32_t) std::min<size_t>(numOuts, m.nout);
    const uint32_t ni = m.diag ? no : (uint32_t) std::min<size_t>(numIns, m.nin);
    EngineUse use(m);
    if (!use.ok)
    {
        // (an empty object whose engine is being replaced, Matrix::relayout_for: the silence it would have computed — on the null
        // stream, which everything the caller enqueues later is ordered behind)
        if (no && numSamples && hipMemset2DAsync(outs_dev, sizeof(float) * out_stride, 0, sizeof(float) * numSamples, no, nullptr) != hipSuccess)
        {
            (void) hipGetLastError();
            set_error("process_f32_dev: could not clear the output block");
            return -1;
        }
        return 0;
    }
    if (!m.engine->process_dev(ins_dev, (int64_t) in_stride, outs_dev, (int64_t) out_stride, ni, no, numSamples, sync != 0))
    {
        set_error(m.engine->last_error());
        return -1;
    }
    return 0;
}

extern "C" int hcv_convolver_synchronize(hcv_convolver *h)
{
    if (h->sh)
    {
        for (hcv_shard &x : h->sh->s)
        {
            // (what the last process call ran on; an object that has not processed yet: its live engine)
            Engine *e = x.cur ? x.cur : x.m->live.load(std::memory_order_acquire);
            if (e && !e->synchronize())
            {
                set_error(e->last_error());
                return -1;
            }
        }
        return 0;
    }
    EngineUse use(*h->m);
    if (!use.ok) return 0;          // (an empty object whose engine is being replaced: nothing of the caller's is in flight on it)
    if (!h->m->engine->synchronize())
    {
        set_error(h->m->engine->last_error());
        return -1;
    }
    return 0;
}

extern "C" int hcv_convolver_device(hcv_convolver *h) { return h->sh ? h->sh->home : h->m->engine->device(); }
extern "C" int hcv_convolver_num_shards(hcv_convolver *h) { return h->sh ? (int) h->sh->s.size() : 1; }
// (the statistics of a sharded object are those of its first shard)
// The diagnostic entry points take the matrix's state mutex: the engine of an empty object is only ever replaced under it
// (Matrix::relayout_for, from set / resize), so none of them can look at an engine that is going away.  (They may wait out a
// set() in progress — milliseconds; they are not for the audio thread.)
namespace
{
    template <class F> void each_engine(hcv_convolver *h, F f)
    {
        if (h->sh)
        {
            for (hcv_shard &x : h->sh->s)
            {
                std::lock_guard<std::mutex> g(x.m->stateMutex);
                f(*x.m->engine);
            }
        }
        else
        {
            std::lock_guard<std::mutex> g(h->m->stateMutex);
            f(*h->m->engine);
        }
    }
    template <class F> auto first_engine(hcv_convolver *h, F f)
    {
        Matrix &m = h->sh ? *h->sh->s[0].m : *h->m;
        std::lock_guard<std::mutex> g(m.stateMutex);
        return f(*m.engine);
    }
}
extern "C" void hcv_convolver_set_profiling(hcv_convolver *h, int on) { each_engine(h, [&](Engine &e) { e.set_profiling(on != 0); }); }
extern "C" int hcv_convolver_num_stages(hcv_convolver *h) { return first_engine(h, [](Engine &e) { return (int) e.num_stages(); }); }
extern "C" void hcv_convolver_clear_stats(hcv_convolver *h)
{
    each_engine(h, [](Engine &e)
    {
        e.clear_stats();
        e.clear_rt_stats();
    });
}

// (the audio-thread counters are atomics of the engine: read without the state mutex — a set() in progress holds that for its whole
// length, posted section included — through the same hold on the engine the process calls take)
extern "C" int hcv_convolver_rt_stats(hcv_convolver *h, hcv_rt_stats *out)
{
    if (!out) return -1;
    out->start_collisions = out->mailbox_runs = out->mailbox_ns_max = out->mailbox_ns_total = out->ctl_sections = out->start_waits = out->arena_misses = 0;
    auto add = [&](Engine &e)
    {
        const Engine::RtStats r = e.rt_stats();
        out->start_collisions += r.start_collisions;
        out->mailbox_runs += r.mailbox_runs;
        out->mailbox_ns_max = std::max<uint64_t>(out->mailbox_ns_max, r.mailbox_ns_max);
        out->mailbox_ns_total += r.mailbox_ns_total;
        out->ctl_sections += r.ctl_sections;
        out->arena_misses += r.arena_misses;
        out->start_waits += r.start_waits;
    };
    auto visit = [&](Matrix &m)
    {
        EngineUse use(m);
        if (use.ok) add(*m.engine);
    };
    if (h->sh)
        for (hcv_shard &x : h->sh->s) visit(*x.m);
    else
        visit(*h->m);
    return 0;
}

// ---- one process per GPU (torch.distributed, MPI, ...): the sum over an input-split row group as ONE RCCL all-reduce on the
// engine's own stream, behind the block's emit — no host synchronisation between the convolution and the collective

extern "C" int hcv_rccl_unique_id(void *out128)
{
    std::string err;
    if (!hcv::rccl_unique_id(out128, &err))
    {
        set_error(err);
        return -1;
    }
    return 0;
}

extern "C" int hcv_convolver_comm_init(hcv_convolver *h, const void *unique_id128, int rank, int nranks)
{
    if (h->sh)
    {
        set_error("hcv_convolver_comm_init: a sharded object sums inside the process; communicators are for one-object-per-GPU deployments");
        return -1;
    }
    std::string err;
    hcv::RcclComm *c = hcv::rccl_comm_create(unique_id128, rank, nranks, h->m->engine->device(), &err);
    if (!c)
    {
        set_error(err);
        return -1;
    }
    hcv::rccl_comm_destroy(h->comm);
    h->comm = c;
    return 0;
}

extern "C" int hcv_convolver_process_f32_dev_allreduce(hcv_convolver *h, const float *ins_dev, size_t in_stride, float *outs_dev, size_t out_stride,
                                                       size_t numIns, size_t numOuts, size_t numSamples, int sync)
{
    if (!h->comm || h->sh)
    {
        set_error("hcv_convolver_process_f32_dev_allreduce: no communicator (hcv_convolver_comm_init)");
        return -1;
    }
    if (hcv_convolver_process_f32_dev(h, ins_dev, in_stride, outs_dev, out_stride, numIns, numOuts, numSamples, 0) != 0) return -1;
    const size_t no = std::min<size_t>(numOuts, h->m->nout);
    std::string err;
    int prev = -1;
    (void) hipGetDevice(&prev);
    (void) hipSetDevice(h->m->engine->device());
    const bool ok = hcv::rccl_all_reduce_sum(h->comm, outs_dev, no, numSamples, out_stride, h->m->engine->main_stream(), &err);
    if (prev >= 0) (void) hipSetDevice(prev);
    if (!ok)
    {
        set_error(err);
        return -1;
    }
    return sync ? hcv_convolver_synchronize(h) : 0;
}

extern "C" int hcv_convolver_stage_stats(hcv_convolver *h, int stage, hcv_stage_stats *out)
{
    hcv::StageStats s;
    if (stage < 0 || !out || !first_engine(h, [&](Engine &e) { return e.stage_stats((size_t) stage, &s); })) return -1;
    out->fft_size = s.fft_size;
    out->partitions = s.partitions;
    out->num_ins = s.nin;
    out->num_outs = s.nout;
    out->mac_launches = s.mac_launches;
    out->mac_hops = s.mac_hops;
    out->mac_ms = s.mac_ms;
    out->ksplit = s.ksplit;
    out->out_tile = s.out_tile;
    out->mac_steady_launches = s.mac_steady_launches;
    out->hop_tile = s.hop_tile;
    out->launch_partitions = s.launch_partitions;
    out->fused_launches = s.fused_launches;
    out->fused_stood_down = s.fused_stood_down;
    out->host_pre_launches = s.host_pre_launches;
    return 0;
}

// ------------------------------------------------------------------------------------------------ library / device

extern "C" const char *hcv_version(void) { return "hisstools_amd 0.1 (gfx950)"; }

extern "C" int hcv_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
    {
        (void) hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int hcv_set_default_device(int device)
{
    if (device >= hcv_device_count()) return -1;
    gDefaultDevice = device;
    return 0;
}

extern "C" int hcv_get_default_device(void) { return gDefaultDevice; }
extern "C" int hcv_ctl_reserve(int device, size_t bytes)
{
    if (device < 0 || device >= hcv_device_count()) return -1;
    return hcv::ctl_arena_reserve(device, bytes) ? 0 : -1;
}
extern "C" size_t hcv_ctl_reserved(int device) { return hcv::ctl_arena_size(device); }
extern "C" long long hcv_order_check_violations(void) { return hcv::order_violations(); }

#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <unistd.h>
namespace
{
struct sigaction gPrevAct[3];
int gCrashFd = 2;           // HCV_NATIVE_BACKTRACE=<path with a slash>: a file of its own (pytest captures descriptor 2 while a test runs)
const int gCrashSigs[3] = {SIGSEGV, SIGABRT, SIGBUS};
void crash_backtrace(int sig, siginfo_t *info, void *ctx)
{
    static const char head[] = "\n[hcv] native call stack of the thread that took the signal:\n";
    (void) !write(gCrashFd, head, sizeof(head) - 1);
    void *frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, gCrashFd);
    for (int k = 0; k < 3; k++)
        if (gCrashSigs[k] == sig)
        {
            sigaction(sig, &gPrevAct[k], nullptr);      // (the previous owner — Python's faulthandler — gets the re-raised signal)
            break;
        }
    (void) info;
    (void) ctx;
    raise(sig);
}
}  // namespace
extern "C" void hcv_debug_native_backtrace_on_crash(void)
{
    if (const char *p = std::getenv("HCV_NATIVE_BACKTRACE"))
        if (std::strchr(p, '/'))
        {
            const int fd = open(p, O_WRONLY | O_CREAT | O_APPEND, 0644);
            if (fd >= 0) gCrashFd = fd;
        }
    for (int k = 0; k < 3; k++)
    {
        struct sigaction sa = {};
        sa.sa_sigaction = crash_backtrace;
        sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_ONSTACK;
        sigemptyset(&sa.sa_mask);
        sigaction(gCrashSigs[k], &sa, &gPrevAct[k]);
    }
}
extern "C" const char *hcv_last_error(void) { return tlsError.c_str(); }
