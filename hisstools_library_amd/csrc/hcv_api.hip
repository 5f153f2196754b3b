// C ABI (include/hisstools_amd.h) and the host-side class semantics of the reference API
// (error codes, clamps, silent-pair rules) on top of the device engine.
//
// Each hcv_* object mirrors one reference class; the arithmetic lives in hcv_kernels.hip, the
// device orchestration in hcv_engine.hip.  Nothing here computes audio on the CPU.

#include "../../include/hisstools_amd.h"
#include "hcv_engine.h"
#include "hcv_fftx.h"
#include "hcv_irx.h"

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

static thread_local std::string tlsError;
static int gDefaultDevice = -1;

static void set_error(const std::string &s) { tlsError = s; }

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

// ------------------------------------------------------------------------------------------------ FFT plumbing

static bool fft_size_ok(unsigned log2n)
{
    if (log2n < 5 || log2n > (unsigned) hcv::kMaxFFTLog2)
    {
        set_error("hcv_rfft/rifft: log2n must be in [5, 20]");
        return false;
    }
    return true;
}

// scratch + sub-transform tables for one-off transforms above the LDS limit
struct ScopedBigWork
{
    hcv::BigFFTWork w;
    bool ok = true;
    ScopedBigWork(int dev, unsigned log2n, size_t batch)
    {
        if (!hcv::is_big_fft((int) log2n)) return;
        int l1, l2;
        hcv::big_fft_split((int) log2n, l1, l2);
        std::string err;
        w.tw1 = hcv::twiddles(dev, l1 + 1, &err);
        w.tw2 = hcv::twiddles(dev, l2 + 1, &err);
        w.elems = (size_t(1) << (log2n - 1)) * std::min<size_t>(batch, 16);
        ok = w.tw1 && w.tw2 && hipMalloc(&w.a, sizeof(float2) * w.elems) == hipSuccess && hipMalloc(&w.b, sizeof(float2) * w.elems) == hipSuccess;
        if (!ok) set_error(err.empty() ? "big FFT workspace allocation failed" : err);
    }
    ~ScopedBigWork()
    {
        if (w.a) (void) hipFree(w.a);
        if (w.b) (void) hipFree(w.b);
    }
};

#define HCV_API_TRY(expr)                                                                                              \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess)                                                                                          \
        {                                                                                                              \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                                              \
            ok = false;                                                                                                \
        }                                                                                                              \
    } while (0)

extern "C" int hcv_rfft_f32(const float *in, size_t in_length, size_t in_stride, size_t batch, unsigned log2n, float *realp, float *imagp)
{
    if (!fft_size_ok(log2n) || !batch) return batch ? -1 : 0;
    int dev = 0;
    if (hcv_device_count() <= 0)
    {
        set_error("no HIP device available");
        return -1;
    }
    if (gDefaultDevice >= 0) (void) hipSetDevice(gDefaultDevice);
    (void) hipGetDevice(&dev);
    std::string err;
    const float2 *tw = hcv::twiddles(dev, (int) log2n, &err);
    if (!tw)
    {
        set_error(err);
        return -1;
    }
    const size_t half = size_t(1) << (log2n - 1), n = half * 2;
    const size_t take = std::min(in_length, n);
    bool ok = true;
    float *din = nullptr;
    float2 *dout = nullptr;
    std::vector<float> packed(batch * take);
    for (size_t b = 0; b < batch; b++) std::memcpy(packed.data() + b * take, in + b * in_stride, sizeof(float) * take);
    std::vector<float2> spec(batch * half);
    HCV_API_TRY(hipMalloc(&din, sizeof(float) * std::max<size_t>(1, batch * take)));
    if (ok) HCV_API_TRY(hipMalloc(&dout, sizeof(float2) * batch * half));
    if (ok && take) HCV_API_TRY(hipMemcpy(din, packed.data(), sizeof(float) * batch * take, hipMemcpyHostToDevice));
    ScopedBigWork big(dev, log2n, batch);
    ok = ok && big.ok;
    if (ok) HCV_API_TRY(hcv::launch_rfft_rows((int) log2n, din, (long long) take, (long long) take, (int) batch, dout, tw, &big.w, nullptr));
    if (ok) HCV_API_TRY(hipMemcpy(spec.data(), dout, sizeof(float2) * batch * half, hipMemcpyDeviceToHost));
    if (din) (void) hipFree(din);
    if (dout) (void) hipFree(dout);
    if (!ok) return -1;
    for (size_t e = 0; e < batch * half; e++)
    {
        realp[e] = spec[e].x;
        imagp[e] = spec[e].y;
    }
    return 0;
}

extern "C" int hcv_rifft_f32(const float *realp, const float *imagp, size_t batch, unsigned log2n, float *out)
{
    if (!fft_size_ok(log2n) || !batch) return batch ? -1 : 0;
    int dev = 0;
    if (hcv_device_count() <= 0)
    {
        set_error("no HIP device available");
        return -1;
    }
    if (gDefaultDevice >= 0) (void) hipSetDevice(gDefaultDevice);
    (void) hipGetDevice(&dev);
    std::string err;
    const float2 *tw = hcv::twiddles(dev, (int) log2n, &err);
    if (!tw)
    {
        set_error(err);
        return -1;
    }
    const size_t half = size_t(1) << (log2n - 1), n = half * 2;
    std::vector<float2> spec(batch * half);
    for (size_t e = 0; e < batch * half; e++) spec[e] = make_float2(realp[e], imagp[e]);
    bool ok = true;
    float2 *din = nullptr;
    float *dout = nullptr;
    HCV_API_TRY(hipMalloc(&din, sizeof(float2) * batch * half));
    if (ok) HCV_API_TRY(hipMalloc(&dout, sizeof(float) * batch * n));
    if (ok) HCV_API_TRY(hipMemcpy(din, spec.data(), sizeof(float2) * batch * half, hipMemcpyHostToDevice));
    ScopedBigWork big(dev, log2n, batch);
    ok = ok && big.ok;
    if (ok) HCV_API_TRY(hcv::launch_rifft_rows((int) log2n, din, (int) batch, dout, tw, &big.w, nullptr));
    if (ok) HCV_API_TRY(hipMemcpy(out, dout, sizeof(float) * batch * n, hipMemcpyDeviceToHost));
    if (din) (void) hipFree(din);
    if (dout) (void) hipFree(dout);
    return ok ? 0 : -1;
}

// ------------------------------------------------------------------------------------------------ spectral_processor (next row)
//
// spectral_processor<float>::convolve / correlate, real overloads (SpectralProcessor.hpp:173-184): both inputs are
// transformed with the real FFT kernels, multiplied bin-wise on the device, inverted and arranged per edge mode.
// Host code only sizes the problem and lays the inputs out (zero padding, mirrored edges); it does no arithmetic.

namespace
{
    enum { EDGE_LINEAR = 0, EDGE_WRAP = 1, EDGE_WRAP_CENTRE = 2, EDGE_FOLD = 3, EDGE_FOLD_REPEAT = 4 };

    struct OpSizes                                              // op_sizes, SpectralProcessor.hpp:318-357
    {
        int mode;
        bool fold;
        size_t size1, size2, mn, mx, linear, fold_copy, fft;
        unsigned fft_log2;
    };

    unsigned spectral_log2(size_t size)                         // calc_fft_size_log2, :231-243
    {
        unsigned count = 0;
        while (count < 8 * sizeof(size_t) && (size >> count)) count++;
        if (count && size == (size_t(1) << (count - 1))) return count - 1;
        return count;
    }

    OpSizes op_sizes(size_t n1, size_t n2, int mode)
    {
        OpSizes s;
        s.mode = mode;
        s.fold = mode == EDGE_FOLD || mode == EDGE_FOLD_REPEAT;
        s.size1 = n1;
        s.size2 = n2;
        s.mn = std::min(n1, n2);
        s.mx = std::max(n1, n2);
        s.linear = n1 + n2 - 1;
        s.fold_copy = s.mx + ((s.mn >> 1) << 1);
        s.fft_log2 = spectral_log2(s.fold ? s.fold_copy + (s.mn - 1) : s.linear);
        s.fft = size_t(1) << s.fft_log2;
        return s;
    }

    // device scratch of one spectral_binary call: the folded operand, both spectra, the circular result, four-step work
    struct SpectralWork
    {
        float *fold = nullptr, *t = nullptr;
        float2 *spec = nullptr;
        hcv::BigFFTWork big;
        size_t fold_elems = 0, t_elems = 0, spec_elems = 0, big_elems = 0;
        void release()
        {
            if (fold) (void) hipFree(fold);
            if (t) (void) hipFree(t);
            if (spec) (void) hipFree(spec);
            if (big.a) (void) hipFree(big.a);
            if (big.b) (void) hipFree(big.b);
            *this = SpectralWork();
        }
        // grow-only; false on allocation failure
        bool reserve(int dev, unsigned log2n, std::string *err)
        {
            const size_t fft = size_t(1) << log2n, half = fft >> 1;
            auto grow = [](auto *&p, size_t &have, size_t want)
            {
                if (have >= want) return true;
                if (p) (void) hipFree(p);
                p = nullptr;
                have = 0;
                if (hipMalloc(&p, sizeof(*p) * want) != hipSuccess) return false;
                have = want;
                return true;
            };
            if (!grow(fold, fold_elems, fft) || !grow(t, t_elems, fft) || !grow(spec, spec_elems, 2 * half)) return false;
            if (hcv::is_big_fft((int) log2n))
            {
                int l1, l2;
                hcv::big_fft_split((int) log2n, l1, l2);
                big.tw1 = hcv::twiddles(dev, l1 + 1, err);
                big.tw2 = hcv::twiddles(dev, l2 + 1, err);
                if (!big.tw1 || !big.tw2) return false;
                if (big_elems < half)
                {
                    if (big.a) (void) hipFree(big.a);
                    if (big.b) (void) hipFree(big.b);
                    big.a = big.b = nullptr;
                    big_elems = 0;
                    if (hipMalloc(&big.a, sizeof(float2) * half) != hipSuccess || hipMalloc(&big.b, sizeof(float2) * half) != hipSuccess) return false;
                    big_elems = half;
                }
                big.elems = big_elems;
            }
            return true;
        }
    };
}

extern "C" size_t hcv_spectral_size(size_t size1, size_t size2, int mode)                      // calc_conv_corr_size, :549-560
{
    if (!size1 || !size2 || mode < 0 || mode > EDGE_FOLD_REPEAT) return 0;
    const OpSizes s = op_sizes(size1, size2, mode);
    if (s.fft_log2 > (unsigned) hcv::kMaxFFTLog2) return 0;     // our "max_fft_size" is 2^20
    return mode != EDGE_LINEAR ? s.mx : s.linear;
}

// Both operands and the result are device-resident; everything is enqueued on `st`.
static bool spectral_core(int dev, const float *d1, size_t n1, const float *d2, size_t n2, int mode, bool correlate, float *dout, SpectralWork &w,
                          hipStream_t st)
{
    const OpSizes s = op_sizes(n1, n2, mode);
    // the device FFTs start at 32 points; a larger circular size is equivalent as long as every index below uses it
    const unsigned log2n = std::max(s.fft_log2, 5u);
    const size_t fft = size_t(1) << log2n, half = fft >> 1;
    std::string err;
    const float2 *tw = hcv::twiddles(dev, (int) log2n, &err);
    if (!tw || !w.reserve(dev, log2n, &err))
    {
        set_error(err.empty() ? "spectral_processor: device allocation failed" : err);
        return false;
    }
    bool ok = true;
    // operands: the longer one is mirrored at both ends in the fold modes (copy_fold, :361-377); the FFT loader zero-pads
    const size_t fold_size = s.mn >> 1;
    const int fold_off = mode == EDGE_FOLD_REPEAT ? 0 : 1;
    const float *row1 = d1, *row2 = d2;
    size_t len1 = n1, len2 = n2;
    if (s.fold)
    {
        const bool first = n1 >= n2;
        HCV_API_TRY(hcv::launch_fold_copy(w.fold, first ? d1 : d2, (long long) (first ? n1 : n2), (long long) fold_size, fold_off, st));
        (first ? row1 : row2) = w.fold;
        (first ? len1 : len2) += 2 * fold_size;
    }
    if (ok) HCV_API_TRY(hcv::launch_rfft_rows((int) log2n, row1, (long long) len1, (long long) len1, 1, w.spec, tw, &w.big, st));
    if (ok) HCV_API_TRY(hcv::launch_rfft_rows((int) log2n, row2, (long long) len2, (long long) len2, 1, w.spec + half, tw, &w.big, st));
    if (ok) HCV_API_TRY(hcv::launch_spectral_pointwise(w.spec, w.spec + half, (int) half, 0.25f / (float) fft, correlate ? 1 : 0, st));
    if (ok) HCV_API_TRY(hcv::launch_rifft_rows((int) log2n, w.spec, 1, w.t, tw, &w.big, st));

    auto seg = [&](size_t o_off, size_t off, size_t n, int op)
    {
        if (ok) HCV_API_TRY(hcv::launch_segment_op(dout, w.t, (long long) o_off, (long long) off, (long long) n, op, st));
    };
    auto copy = [&](size_t o_off, size_t off, size_t n) { seg(o_off, off, n, 0); };
    auto wrap = [&](size_t o_off, size_t last, size_t n) { seg(o_off, last - n, n, 1); };     // adds t[last-n .. last)
    auto zero = [&](size_t a, size_t b) { if (b > a) seg(a, 0, b - a, 2); };

    if (n1 == 1 && n2 == 1)
        copy(0, 0, 1);                                          // circular product of two 1-sample signals
    else if (!correlate)
    {
        const size_t min_m1 = s.mn - 1;                         // arrange_convolve, :448-486
        switch (mode)
        {
            case EDGE_LINEAR: copy(0, 0, s.linear); break;
            case EDGE_WRAP: copy(0, 0, s.mx); wrap(0, s.linear, min_m1); break;
            case EDGE_WRAP_CENTRE:
            {
                const size_t wrapped = min_m1 >> 1;
                copy(0, wrapped, s.mx);
                wrap(0, s.linear, min_m1 - wrapped);
                wrap(s.mx - wrapped, wrapped, wrapped);
                break;
            }
            default: copy(0, min_m1, s.mx); break;
        }
    }
    else
    {
        const size_t size2_m1 = s.size2 - 1;                    // arrange_correlate, :488-545 (fft = the size actually used)
        switch (mode)
        {
            case EDGE_LINEAR: copy(0, 0, s.size1); copy(s.size1, fft - size2_m1, size2_m1); break;
            case EDGE_WRAP:
                copy(0, 0, s.size1);
                zero(s.size1, s.size2);
                wrap(s.mx - size2_m1, fft, size2_m1);
                break;
            case EDGE_WRAP_CENTRE:
            {
                const size_t w1 = (s.mn - 1) >> 1;
                const size_t w2 = std::min(size2_m1, s.mx - w1);
                const size_t w3 = size2_m1 - w2;
                const size_t offset = w3 ? 0 : s.mx - (size2_m1 + w1);
                zero(0, s.mx);
                copy(0, w1, s.size1 - w1);
                copy(s.mx - w1, 0, w1);
                wrap(offset, fft, w2);
                wrap(s.mx - w3, fft - w2, w3);
                break;
            }
            default:
                if (s.size1 >= s.size2)
                    copy(0, 0, s.mx);
                else
                {
                    const size_t cs = s.mx - 1;
                    copy(0, 0, 1);
                    copy(1, fft - cs, cs);
                }
                break;
        }
    }
    return ok;
}

static bool spectral_ready(size_t n1, size_t n2, int mode, int &dev, size_t &result)
{
    result = hcv_spectral_size(n1, n2, mode);
    if (!result) return false;                                  // the reference returns without touching `out` (:651-652)
    if (hcv_device_count() <= 0)
    {
        set_error("no HIP device available (no CPU fallback)");
        result = (size_t) -1;
        return false;
    }
    if (gDefaultDevice >= 0) (void) hipSetDevice(gDefaultDevice);
    (void) hipGetDevice(&dev);
    return true;
}

static int spectral_binary(const float *in1, size_t n1, const float *in2, size_t n2, int mode, bool correlate, float *out)
{
    int dev = 0;
    size_t result = 0;
    if (!spectral_ready(n1, n2, mode, dev, result)) return result == (size_t) -1 ? -1 : 0;
    bool ok = true;
    float *d1 = nullptr, *d2 = nullptr, *dout = nullptr;
    SpectralWork w;
    HCV_API_TRY(hipMalloc(&d1, sizeof(float) * n1));
    if (ok) HCV_API_TRY(hipMalloc(&d2, sizeof(float) * n2));
    if (ok) HCV_API_TRY(hipMalloc(&dout, sizeof(float) * result));
    if (ok) HCV_API_TRY(hipMemcpy(d1, in1, sizeof(float) * n1, hipMemcpyHostToDevice));
    if (ok) HCV_API_TRY(hipMemcpy(d2, in2, sizeof(float) * n2, hipMemcpyHostToDevice));
    ok = ok && spectral_core(dev, d1, n1, d2, n2, mode, correlate, dout, w, nullptr);
    if (ok) HCV_API_TRY(hipMemcpy(out, dout, sizeof(float) * result, hipMemcpyDeviceToHost));
    if (d1) (void) hipFree(d1);
    if (d2) (void) hipFree(d2);
    if (dout) (void) hipFree(dout);
    w.release();
    return ok ? 0 : -1;
}

// HBM-resident operands: scratch is cached per device (grow-only) and the calls of one device are serialised by a mutex
// while they enqueue; calls on different streams that overlap in time must be ordered by the caller.
static int spectral_binary_dev(const float *d1, size_t n1, const float *d2, size_t n2, int mode, bool correlate, float *dout, void *stream, int sync)
{
    int dev = 0;
    size_t result = 0;
    if (!spectral_ready(n1, n2, mode, dev, result)) return result == (size_t) -1 ? -1 : 0;
    static std::mutex mutex;
    static std::map<int, SpectralWork> cache;
    std::lock_guard<std::mutex> g(mutex);
    hipStream_t st = static_cast<hipStream_t>(stream);
    bool ok = spectral_core(dev, d1, n1, d2, n2, mode, correlate, dout, cache[dev], st);
    if (ok && sync) HCV_API_TRY(hipStreamSynchronize(st));
    return ok ? 0 : -1;
}

extern "C" int hcv_spectral_convolve_f32(const float *in1, size_t size1, const float *in2, size_t size2, int mode, float *out)
{
    return spectral_binary(in1, size1, in2, size2, mode, false, out);
}

extern "C" int hcv_spectral_correlate_f32(const float *in1, size_t size1, const float *in2, size_t size2, int mode, float *out)
{
    return spectral_binary(in1, size1, in2, size2, mode, true, out);
}

extern "C" int hcv_spectral_convolve_f32_dev(const float *in1, size_t size1, const float *in2, size_t size2, int mode, float *out, void *stream, int sync)
{
    return spectral_binary_dev(in1, size1, in2, size2, mode, false, out, stream, sync);
}

extern "C" int hcv_spectral_correlate_f32_dev(const float *in1, size_t size1, const float *in2, size_t size2, int mode, float *out, void *stream, int sync)
{
    return spectral_binary_dev(in1, size1, in2, size2, mode, true, out, stream, sync);
}

// ------------------------------------------------------------------------------------------------ the full FFT surface (next row 2)
//
// HISSTools_FFT.h:87-369 as one batched entry point; the kernels are in hcv_fftx.hip.  The host variant only moves
// bytes: upload the source extent, run, download the destination extent.

namespace
{
    struct FftOperand
    {
        size_t len = 0, elem = 0;       // elements per transform, bytes per element
        bool two = false;               // split (a and b) or samples (a only)
    };

    bool use_default_device(int &dev)
    {
        if (hcv_device_count() <= 0)
        {
            set_error("no HIP device available");
            return false;
        }
        if (gDefaultDevice >= 0) (void) hipSetDevice(gDefaultDevice);
        return hipGetDevice(&dev) == hipSuccess;
    }

    hcv::FxCall to_fx(const hcv_fft_call &c)
    {
        hcv::FxCall f;
        f.op = c.op; f.precision = c.precision; f.log2n = c.log2n; f.batch = c.batch;
        f.src_a = c.src_a; f.src_b = c.src_b; f.dst_a = c.dst_a; f.dst_b = c.dst_b;
        f.src_stride = c.src_stride; f.dst_stride = c.dst_stride; f.in_length = c.in_length;
        return f;
    }

    // shapes of the two operands, and default (dense) strides
    void fft_operands(hcv::FxCall &f, FftOperand &src, FftOperand &dst)
    {
        const size_t n = size_t(1) << f.log2n, half = n >> 1;
        const size_t real_bytes = f.precision == hcv::FX_F32 ? 4 : 8;
        const bool complex_op = f.op == hcv::FX_FFT || f.op == hcv::FX_IFFT;
        const size_t split_len = complex_op ? n : half;
        src.elem = dst.elem = real_bytes;
        switch (f.op)
        {
            case hcv::FX_RFFT_ZIP:
            case hcv::FX_UNZIP:
                f.in_length = std::min(f.in_length, n);
                src.len = f.in_length; src.two = false;
                src.elem = f.precision == hcv::FX_F64 ? 8 : 4;
                dst.len = half; dst.two = true;
                break;
            case hcv::FX_RIFFT_ZIP:
            case hcv::FX_ZIP:
                src.len = half; src.two = true;
                dst.len = half ? n : 0; dst.two = false;
                break;
            default:
                src.len = dst.len = split_len;
                src.two = dst.two = true;
        }
        if (!f.src_stride) f.src_stride = src.len;
        if (!f.dst_stride) f.dst_stride = dst.len;
    }
}

extern "C" int hcv_fft_exec_dev(const hcv_fft_call *call, void *stream, int sync)
{
    if (!call)
    {
        set_error("hcv_fft_exec_dev: null descriptor");
        return -1;
    }
    int dev = 0;
    if (!use_default_device(dev)) return -1;
    hcv::FxCall f = to_fx(*call);
    std::string err;
    if (!hcv::fftx_valid(f, &err))
    {
        set_error(err);
        return -1;
    }
    FftOperand src, dst;
    fft_operands(f, src, dst);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hcv::fftx_exec(dev, f, st, &err);
    if (e == hipSuccess && sync) e = hipStreamSynchronize(st);
    if (e != hipSuccess)
    {
        set_error(err.empty() ? std::string("hcv_fft_exec_dev: ") + hipGetErrorString(e) : err);
        return -1;
    }
    return 0;
}

extern "C" int hcv_fft_exec(const hcv_fft_call *call)
{
    if (!call)
    {
        set_error("hcv_fft_exec: null descriptor");
        return -1;
    }
    int dev = 0;
    if (!use_default_device(dev)) return -1;
    hcv::FxCall f = to_fx(*call);
    std::string err;
    if (!hcv::fftx_valid(f, &err))
    {
        set_error(err);
        return -1;
    }
    if (!f.batch) return 0;
    FftOperand src, dst;
    fft_operands(f, src, dst);
    const bool in_place = call->src_a == call->dst_a;
    if (in_place && (src.two != dst.two || src.elem != dst.elem || (src.two && call->src_b != call->dst_b) || f.src_stride != f.dst_stride))
    {
        set_error("hcv_fft_exec: operands may alias only exactly (same layout, both arrays)");
        return -1;
    }
    const size_t src_extent = src.len ? (f.batch - 1) * f.src_stride + src.len : 0;
    const size_t dst_extent = dst.len ? (f.batch - 1) * f.dst_stride + dst.len : 0;
    if (!dst_extent) return 0;

    bool ok = true;
    void *d[4] = { nullptr, nullptr, nullptr, nullptr };            // src a, src b, dst a, dst b
    auto up = [&](void *&p, const void *host, size_t elems, size_t elem_bytes, bool copy)
    {
        if (!ok) return;
        HCV_API_TRY(hipMalloc(&p, std::max<size_t>(16, elems * elem_bytes)));
        if (ok && copy && elems) HCV_API_TRY(hipMemcpy(p, host, elems * elem_bytes, hipMemcpyHostToDevice));
    };
    up(d[0], call->src_a, src_extent, src.elem, true);
    if (src.two) up(d[1], call->src_b, src_extent, src.elem, true);
    if (in_place)
    {
        d[2] = d[0];
        d[3] = d[1];
    }
    else
    {
        // gaps between strided destination rows keep the caller's bytes
        const bool gaps = f.dst_stride != dst.len && f.batch > 1;
        up(d[2], call->dst_a, dst_extent, dst.elem, gaps);
        if (dst.two) up(d[3], call->dst_b, dst_extent, dst.elem, gaps);
    }
    if (ok)
    {
        f.src_a = d[0]; f.src_b = d[1]; f.dst_a = d[2]; f.dst_b = d[3];
        hipError_t e = hcv::fftx_exec(dev, f, nullptr, &err);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess)
        {
            set_error(err.empty() ? std::string("hcv_fft_exec: ") + hipGetErrorString(e) : err);
            ok = false;
        }
    }
    if (ok) HCV_API_TRY(hipMemcpy(call->dst_a, d[2], dst_extent * dst.elem, hipMemcpyDeviceToHost));
    if (ok && dst.two) HCV_API_TRY(hipMemcpy(call->dst_b, d[3], dst_extent * dst.elem, hipMemcpyDeviceToHost));
    if (!in_place)
    {
        if (d[2]) (void) hipFree(d[2]);
        if (d[3]) (void) hipFree(d[3]);
    }
    if (d[0]) (void) hipFree(d[0]);
    if (d[1]) (void) hipFree(d[1]);
    return ok ? 0 : -1;
}

// ------------------------------------------------------------------------------------------------ spectral IR functions (next row 4)
//
// SpectralFunctions.hpp:365-413 on batches of packed half spectra; kernels in hcv_irx.hip.  The host variants only move bytes.

namespace
{
    hcv::IrCall to_ir(const hcv_ir_call &c)
    {
        hcv::IrCall r;
        r.op = c.op; r.precision = c.precision; r.log2n = c.log2n; r.batch = c.batch;
        r.src_re = c.src_re; r.src_im = c.src_im; r.dst_re = c.dst_re; r.dst_im = c.dst_im;
        r.src_stride = c.src_stride; r.dst_stride = c.dst_stride; r.value = c.value; r.zero_center = c.zero_center;
        return r;
    }

    // device buffer helper for the host-pointer entries
    struct DevBuf
    {
        void *p = nullptr;
        ~DevBuf() { if (p) (void) hipFree(p); }
        bool alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(16, bytes)) == hipSuccess; }
    };
}

extern "C" int hcv_ir_exec_dev(const hcv_ir_call *call, void *stream, int sync)
{
    if (!call)
    {
        set_error("hcv_ir_exec_dev: null descriptor");
        return -1;
    }
    int dev = 0;
    if (!use_default_device(dev)) return -1;
    std::string err;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hcv::irx_exec(dev, to_ir(*call), st, &err);
    if (e == hipSuccess && sync) e = hipStreamSynchronize(st);
    if (e != hipSuccess)
    {
        set_error(err.empty() ? std::string("hcv_ir_exec_dev: ") + hipGetErrorString(e) : err);
        return -1;
    }
    return 0;
}

extern "C" int hcv_ir_exec(const hcv_ir_call *call)
{
    if (!call)
    {
        set_error("hcv_ir_exec: null descriptor");
        return -1;
    }
    int dev = 0;
    if (!use_default_device(dev)) return -1;
    hcv::IrCall c = to_ir(*call);
    std::string err;
    if (!hcv::irx_valid(c, &err))
    {
        set_error(err);
        return -1;
    }
    if (!c.batch) return 0;
    const size_t half = (size_t(1) << c.log2n) >> 1, elem = c.precision == hcv::FX_F32 ? 4 : 8;
    if (!c.src_stride) c.src_stride = half;
    if (!c.dst_stride) c.dst_stride = half;
    const size_t src_extent = (c.batch - 1) * c.src_stride + half, dst_extent = (c.batch - 1) * c.dst_stride + half;
    const bool has_src = c.op != hcv::IR_SPIKE;
    const bool in_place = has_src && call->src_re == call->dst_re && call->src_im == call->dst_im && c.src_stride == c.dst_stride;
    DevBuf sr, si, dr, di;
    bool ok = true;
    if (has_src)
    {
        ok = sr.alloc(src_extent * elem) && si.alloc(src_extent * elem);
        if (ok) HCV_API_TRY(hipMemcpy(sr.p, call->src_re, src_extent * elem, hipMemcpyHostToDevice));
        if (ok) HCV_API_TRY(hipMemcpy(si.p, call->src_im, src_extent * elem, hipMemcpyHostToDevice));
    }
    if (ok && !in_place)
    {
        ok = dr.alloc(dst_extent * elem) && di.alloc(dst_extent * elem);
        const bool gaps = c.dst_stride != half && c.batch > 1;
        if (ok && gaps) HCV_API_TRY(hipMemcpy(dr.p, call->dst_re, dst_extent * elem, hipMemcpyHostToDevice));
        if (ok && gaps) HCV_API_TRY(hipMemcpy(di.p, call->dst_im, dst_extent * elem, hipMemcpyHostToDevice));
    }
    if (!ok)
    {
        if (tlsError.empty()) set_error("hcv_ir_exec: device allocation failed");
        return -1;
    }
    c.src_re = sr.p; c.src_im = si.p;
    c.dst_re = in_place ? sr.p : dr.p;
    c.dst_im = in_place ? si.p : di.p;
    hipError_t e = hcv::irx_exec(dev, c, nullptr, &err);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess)
    {
        set_error(err.empty() ? std::string("hcv_ir_exec: ") + hipGetErrorString(e) : err);
        return -1;
    }
    HCV_API_TRY(hipMemcpy(call->dst_re, c.dst_re, dst_extent * elem, hipMemcpyDeviceToHost));
    if (ok) HCV_API_TRY(hipMemcpy(call->dst_im, c.dst_im, dst_extent * elem, hipMemcpyDeviceToHost));
    return ok ? 0 : -1;
}

// spectral_processor::calc_fft_size_log2 (SpectralProcessor.hpp:231-242) of round(size * time_multiplier)
static unsigned phase_fft_log2(size_t size, double time_multiplier)
{
    const size_t want = (size_t) std::llround((double) size * time_multiplier);
    unsigned count = 0;
    while (count < 63 && (want >> count)) count++;
    if (count && want == (size_t(1) << (count - 1))) return count - 1;
    return count;
}

extern "C" size_t hcv_spectral_phase_size(size_t size, double time_multiplier)
{
    if (size == 1) return 1;
    const unsigned l2 = phase_fft_log2(size, time_multiplier);
    return l2 > (unsigned) hcv::kFxMaxComplexLog2 + 1 ? 0 : size_t(1) << l2;
}

template <class T> static int change_phase(const T *in, size_t size, double phase, double time_multiplier, T *out)
{
    if (!in || !out || !size)
    {
        set_error("hcv_spectral_change_phase: null or empty input");
        return -1;
    }
    if (size == 1)                                             // SpectralProcessor.hpp:195-199
    {
        out[0] = in[0];
        return 0;
    }
    const unsigned log2n = phase_fft_log2(size, time_multiplier);
    if (log2n < 3 || log2n > (unsigned) hcv::kFxMaxComplexLog2 + 1)
    {
        set_error("hcv_spectral_change_phase: fft size out of range (8 .. 2^23)");
        return -1;
    }
    int dev = 0;
    if (!use_default_device(dev)) return -1;
    const size_t n = size_t(1) << log2n, half = n >> 1, take = std::min(size, n);
    const int prec = sizeof(T) == 4 ? hcv::FX_F32 : hcv::FX_F64;
    DevBuf x, re, im, y;
    bool ok = x.alloc(sizeof(T) * n) && re.alloc(sizeof(T) * half) && im.alloc(sizeof(T) * half) && y.alloc(sizeof(T) * n);
    if (!ok)
    {
        set_error("hcv_spectral_change_phase: device allocation failed");
        return -1;
    }
    HCV_API_TRY(hipMemcpy(x.p, in, sizeof(T) * take, hipMemcpyHostToDevice));
    std::string err;
    hipError_t e = hipSuccess;
    if (ok)
    {
        hcv::FxCall f;
        f.precision = prec; f.log2n = log2n; f.batch = 1;
        f.op = hcv::FX_RFFT_ZIP; f.src_a = x.p; f.dst_a = re.p; f.dst_b = im.p; f.in_length = take; f.src_stride = n; f.dst_stride = half;
        e = hcv::fftx_exec(dev, f, nullptr, &err);
        hcv::IrCall c;
        c.op = hcv::IR_PHASE; c.precision = prec; c.log2n = log2n; c.batch = 1; c.value = phase; c.zero_center = 0;
        c.src_re = c.dst_re = re.p; c.src_im = c.dst_im = im.p;
        if (e == hipSuccess) e = hcv::irx_exec(dev, c, nullptr, &err);
        f.op = hcv::FX_RIFFT_ZIP; f.src_a = re.p; f.src_b = im.p; f.dst_a = y.p; f.dst_b = nullptr; f.src_stride = half; f.dst_stride = n;
        if (e == hipSuccess) e = hcv::fftx_exec(dev, f, nullptr, &err);
        if (e == hipSuccess) e = hcv::launch_scale(static_cast<T *>(y.p), (long long) n, (T) 0.5 / (T) n, nullptr);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess)
        {
            set_error(err.empty() ? std::string("hcv_spectral_change_phase: ") + hipGetErrorString(e) : err);
            return -1;
        }
    }
    if (ok) HCV_API_TRY(hipMemcpy(out, y.p, sizeof(T) * n, hipMemcpyDeviceToHost));
    return ok ? 0 : -1;
}

extern "C" int hcv_spectral_change_phase_f32(const float *in, size_t size, double phase, double time_multiplier, float *out)
{
    return change_phase<float>(in, size, phase, time_multiplier, out);
}

extern "C" int hcv_spectral_change_phase_f64(const double *in, size_t size, double phase, double time_multiplier, double *out)
{
    return change_phase<double>(in, size, phase, time_multiplier, out);
}
