// C ABI (include/hisstools_amd.h) and the host-side class semantics of the reference API
// (error codes, clamps, silent-pair rules) on top of the device engine.
//
// Each hcv_* object mirrors one reference class; the arithmetic lives in hcv_kernels.hip, the
// device orchestration in hcv_engine.hip.  Nothing here computes audio on the CPU.

#include "../../include/hisstools_amd.h"
#include "hcv_engine.h"
#include "hcv_api_common.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
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
        void extend_tail(uint64_t maxLength, uint32_t ratio)
        {
            if (ratio < 2) return;
            const uint64_t latency = zeroLatency ? 0 : sizes[0] >> 1;
            uint64_t cur = largest, curOffset = tailOffset;
            while (cur * ratio <= (uint64_t(1) << 20))
            {
                const uint64_t next = cur * ratio, nextOffset = (next >> 1) - latency;
                if (nextOffset >= maxLength) break;
                StageCfg st;
                st.fft_size = (uint32_t) cur;
                st.offset = curOffset;
                st.length = nextOffset - curOffset;
                st.capacity = st.length;
                fixedStages.push_back(st);
                cur = next;
                curOffset = nextOffset;
            }
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

    struct Matrix
    {
        Layout layout;
        uint32_t nin = 1, nout = 1;
        bool diag = false;
        std::unique_ptr<Engine> engine;
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
            EngineCfg cfg;
            cfg.nin = nin;
            cfg.nout = nout;
            cfg.diag = diag;
            cfg.device = device;
            cfg.max_block = maxBlock;
            if (layout.zeroLatency)
            {
                cfg.has_td = true;
                cfg.td_offset = 0;
                cfg.td_length = layout.sizes[0] >> 1;                   // TimeDomainConvolve(0, A/2), :240
                if (cfg.td_length > 2044) cfg.td_length = 2044;         // TimeDomainConvolve::setLength clamp
            }
            cfg.stages = layout.fixedStages;
            StageCfg tl;
            tl.fft_size = layout.largest;
            tl.offset = layout.tailOffset;
            tl.length = 0;
            tl.capacity = layout.tail_capacity(maxLength);
            cfg.stages.push_back(tl);
            std::string err;
            engine.reset(Engine::create(cfg, &err));
            if (!engine)
            {
                set_error(err);
                return false;
            }
            const size_t pairs = (size_t) nout * (diag ? 1 : nin);
            mLength.assign(pairs, 0);
            part4Size.assign(pairs, maxLength);                          // part4.equal(..., maxLength), :254
            part4Alloc.assign(pairs, maxLength ? 1 : 0);
            return true;
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

        bool active(size_t p) const
        {
            std::lock_guard<std::mutex> g(stateMutex);
            return mLength[p] && mLength[p] <= part4Size[p];
        }

        // MonoConvolve::resize (.cpp:101-110)
        int resize(uint32_t in, uint32_t out, uint64_t length)
        {
            std::lock_guard<std::mutex> g(stateMutex);
            const size_t p = pair(in, out);
            mLength[p] = 0;
            engine->set_ir(in, out, nullptr, 0, false);                  // the pair is silent until the next set
            tail_equal(p, length);
            return part4Size[p] == length ? HCV_ERR_NONE : HCV_ERR_MEM_UNAVAILABLE;
        }

        // MonoConvolve::set (.cpp:118-140)
        int set(uint32_t in, uint32_t out, const float *ir, uint64_t length, bool requestResize, bool devicePtr)
        {
            std::lock_guard<std::mutex> g(stateMutex);
            const size_t p = pair(in, out);
            mLength[p] = 0;
            if (requestResize) tail_equal(p, length);
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
        m->layout.extend_tail(maxLength, tailRatio);
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

struct hcv_mono { std::unique_ptr<Matrix> m; };
struct hcv_ntomono { std::unique_ptr<Matrix> m; };
struct hcv_convolver
{
    std::unique_ptr<Matrix> m;
    std::vector<float> tmpIn, tmpOut;          // float staging of the double overloads
};

// ---- MonoConvolve

extern "C" hcv_mono *hcv_mono_create_custom(uintptr_t maxLength, int zeroLatency, uint32_t A, uint32_t B, uint32_t C, uint32_t D, char *err, size_t errlen)
{
    std::string e;
    Matrix *m = make_matrix(1, 1, false, maxLength, zeroLatency != 0, A, B, C, D, gDefaultDevice, 0, &e);
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
    return h;
}

extern "C" hcv_mono *hcv_mono_create(uintptr_t maxLength, int latency)
{
    bool zero;
    uint32_t A, B, C, D;
    latency_sizes(latency, zero, A, B, C, D);
    return hcv_mono_create_custom(maxLength, zero, A, B, C, D, nullptr, 0);
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
    const float *ins[1] = { in };
    float *outs[1] = { out };
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
    Matrix *m = make_matrix(inChans, 1, false, maxLength, zero, A, B, C, D, gDefaultDevice, 0, nullptr);
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

extern "C" hcv_convolver *hcv_convolver_create_on(uint32_t numIns, uint32_t numOuts, int latency, int device, uint32_t maxBlock)
{
    bool zero;
    uint32_t A, B, C, D;
    latency_sizes(latency, zero, A, B, C, D);
    numIns = numIns < 1 ? 1 : numIns;                       // Convolver.cpp:8
    return wrap(make_matrix(numIns, numOuts, false, 16384, zero, A, B, C, D, device < 0 ? gDefaultDevice : device, maxBlock, nullptr));
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
    return wrap(make_matrix(numIO, numIO, true, 16384, zero, A, B, C, D, gDefaultDevice, 0, nullptr));
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
static bool conv_in_ok(const Matrix &m, uint32_t &inChan, uint32_t outChan)
{
    if (m.diag)
    {
        inChan -= outChan;
        return inChan < 1;
    }
    return inChan < m.nin;
}

extern "C" int hcv_convolver_set_f32(hcv_convolver *h, uint32_t inChan, uint32_t outChan, const float *input, uintptr_t length, int resize)
{
    Matrix &m = *h->m;
    const bool inOk = conv_in_ok(m, inChan, outChan);
    if (outChan >= m.nout) return HCV_ERR_OUT_CHAN_OUT_OF_RANGE;
    if (!inOk) return HCV_ERR_IN_CHAN_OUT_OF_RANGE;
    return m.set(m.diag ? outChan : inChan, outChan, input, length, resize != 0, false);
}

extern "C" int hcv_convolver_set_f32_dev(hcv_convolver *h, uint32_t inChan, uint32_t outChan, const float *input_dev, uintptr_t length, int resize)
{
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
    Matrix &m = *h->m;
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
    Matrix &m = *h->m;
    const bool inOk = conv_in_ok(m, inChan, outChan);
    if (outChan >= m.nout) return HCV_ERR_OUT_CHAN_OUT_OF_RANGE;
    if (!inOk) return HCV_ERR_IN_CHAN_OUT_OF_RANGE;
    m.engine->reset_pair(m.diag ? outChan : inChan, outChan);
    return HCV_ERR_NONE;
}

extern "C" void hcv_convolver_reset(hcv_convolver *h) { h->m->engine->reset_all(); }

extern "C" int hcv_convolver_resize(hcv_convolver *h, uint32_t inChan, uint32_t outChan, uintptr_t length)
{
    Matrix &m = *h->m;
    const bool inOk = conv_in_ok(m, inChan, outChan);
    if (outChan >= m.nout) return HCV_ERR_IN_CHAN_OUT_OF_RANGE;          // sic: Convolver.cpp:108-111 returns the IN code here
    if (!inOk) return HCV_ERR_IN_CHAN_OUT_OF_RANGE;
    return m.resize(m.diag ? outChan : inChan, outChan, length);
}

extern "C" int hcv_convolver_process_f32(hcv_convolver *h, const float *const *ins, float **outs, size_t numIns, size_t numOuts, size_t numSamples)
{
    Matrix &m = *h->m;
    const uint32_t no = (uint32_t) std::min<size_t>(numOuts, m.nout);
    const uint32_t ni = m.diag ? no : (uint32_t) std::min<size_t>(numIns, m.nin);
    if (!m.engine->process(ins, outs, ni, no, numSamples, false))
    {
        set_error(m.engine->last_error());
        return -1;
    }
    return 0;
}

extern "C" int hcv_convolver_process_f64(hcv_convolver *h, const double *const *ins, double **outs, size_t numIns, size_t numOuts, size_t numSamples)
{
    Matrix &m = *h->m;
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
    if (!m.engine->process(ip.data(), op.data(), ni, no, numSamples, false))
    {
        set_error(m.engine->last_error());
        return -1;
    }
    for (uint32_t o = 0; o < no; o++)
        for (size_t j = 0; j < numSamples; j++) outs[o][j] = op[o][j];
    return 0;
}

extern "C" int hcv_convolver_process_f32_dev(hcv_convolver *h, const float *ins_dev, size_t in_stride, float *outs_dev, size_t out_stride, size_t numIns,
                                             size_t numOuts, size_t numSamples, int sync)
{
    Matrix &m = *h->m;
    const uint32_t no = (uint32_t) std::min<size_t>(numOuts, m.nout);
    const uint32_t ni = m.diag ? no : (uint32_t) std::min<size_t>(numIns, m.nin);
    if (!m.engine->process_dev(ins_dev, (int64_t) in_stride, outs_dev, (int64_t) out_stride, ni, no, numSamples, sync != 0))
    {
        set_error(m.engine->last_error());
        return -1;
    }
    return 0;
}

extern "C" int hcv_convolver_synchronize(hcv_convolver *h)
{
    if (!h->m->engine->synchronize())
    {
        set_error(h->m->engine->last_error());
        return -1;
    }
    return 0;
}

extern "C" int hcv_convolver_device(hcv_convolver *h) { return h->m->engine->device(); }
extern "C" void hcv_convolver_set_profiling(hcv_convolver *h, int on) { h->m->engine->set_profiling(on != 0); }
extern "C" int hcv_convolver_num_stages(hcv_convolver *h) { return (int) h->m->engine->num_stages(); }
extern "C" void hcv_convolver_clear_stats(hcv_convolver *h) { h->m->engine->clear_stats(); }

extern "C" int hcv_convolver_stage_stats(hcv_convolver *h, int stage, hcv_stage_stats *out)
{
    hcv::StageStats s;
    if (stage < 0 || !out || !h->m->engine->stage_stats((size_t) stage, &s)) return -1;
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
    out->reserved = 0;
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
extern "C" const char *hcv_last_error(void) { return tlsError.c_str(); }
