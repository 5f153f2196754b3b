// Device-resident N x M multi-stage convolution engine: host-side orchestration of the gfx950 kernels.
// See hcv_engine.h for the HBM layout.

#include "hcv_engine_impl.h"

#include <chrono>
#include <cstdio>
#include <map>
#include <thread>

#include <dlfcn.h>

namespace hcv
{

namespace
{
    struct CtlArena;
    CtlArena *arena_of(int device, bool create);       // the device's control arena (below, with ctl_alloc)
    void arena_reap_all(int device);                   // every parked block of the device's arena back to its free list, waiting for their events
    void arena_engine(int device, long long want);     // an engine of the device comes (its want of arena bytes) or goes (minus it)
}

static int ilog2(uint64_t v)
{
    int l = 0;
    while ((uint64_t(1) << l) < v) l++;
    return l;
}

// ------------------------------------------------------------------------------------------------
// twiddle tables: N-th roots of unity exp(-2 pi i m / N), m < N/2, computed in double and rounded once
// (the reference builds its tables the same way, HISSTools_FFT_Core.h:414-448); one table per (device, N)
// shared by every engine instead of one setup per PartitionedConvolve (PartitionedConvolve.cpp:101).
// ------------------------------------------------------------------------------------------------

static std::mutex gTwMutex;
static std::map<std::pair<int, int>, float2 *> gTwTables;

// A single-stage engine without a time-domain head (a plain PartitionedConvolve) has nothing to overlap within a block:
// its kernels form one dependency chain, and every cross-stream hop of that chain costs microseconds.  It runs every
// kernel on one stream (measured on config 2: 0.070 -> 0.052 ms per 8192-sample block; the four-stage config 3 loses
// 30 % without the overlap, so multi-stage engines keep their streams).
static bool one_stream_mode(const EngineCfg &cfg)
{
    return cfg.stages.size() == 1 && !cfg.has_td;
}

const float2 *twiddles(int device, int log2n, std::string *err)
{
    std::lock_guard<std::mutex> g(gTwMutex);
    auto key = std::make_pair(device, log2n);
    auto it = gTwTables.find(key);
    if (it != gTwTables.end()) return it->second;

    const size_t half = size_t(1) << (log2n - 1);
    std::vector<float2> host(half);
    const double pi = 3.14159265358979323846264338327950288;
    for (size_t m = 0; m < half; m++)
    {
        double angle = -(double) m * pi / (double) half;
        host[m] = make_float2((float) std::cos(angle), (float) std::sin(angle));
    }
    float2 *dev = nullptr;
    hipError_t e = hipMalloc(&dev, half * sizeof(float2));
    if (e == hipSuccess) e = hipMemcpy(dev, host.data(), half * sizeof(float2), hipMemcpyHostToDevice);
    if (e != hipSuccess)
    {
        if (err) *err = std::string("twiddle table upload failed: ") + hipGetErrorString(e);
        if (dev) (void) hipFree(dev);
        return nullptr;
    }
    gTwTables[key] = dev;
    return dev;
}

// ------------------------------------------------------------------------------------------------

// shape of one spectral_mac launch of a stage (hcv_kernels.h: MacShape)
MacShape Engine::mac_shape(const Stage &st, int P, int Pcap, int nin, int nin_alloc, int nout, int diag, int T, int max_ksplit)
{
    MacShape s;
    s.M = (int) st.M;
    s.R = (int) st.R;
    s.P = P;
    s.Pcap = Pcap;
    s.nin = nin;
    s.nin_alloc = nin_alloc;
    s.nout = nout;
    s.diag = diag;
    s.T = T;
    s.max_ksplit = max_ksplit;
    s.target_blocks = 0;
    s.ot_cap = 0;
    return s;
}

bool Engine::fail(const char *what, hipError_t e)
{
    mErr = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}

namespace
{
    std::mutex gStreamPoolMutex;
    std::map<int, std::vector<hipStream_t>> gStreamPool;             // per device: idle streams
}

// HCV_STREAM_POOL = 0 (round 6, VERDICT r5 item 6: the abort inside hipStreamDestroy was side-stepped by the pool, not explained): streams are
// created and DESTROYED again as before round 5 — for tools/micro/crash_loop.sh and the runtime-only repro beside it, never for production.
static bool stream_pool_on()
{
    static const bool on = !(std::getenv("HCV_STREAM_POOL") && std::atoi(std::getenv("HCV_STREAM_POOL")) == 0);
    return on;
}

hipError_t stream_take(int device, hipStream_t *s)
{
    if (stream_pool_on())
    {
        std::lock_guard<std::mutex> g(gStreamPoolMutex);
        std::vector<hipStream_t> &pool = gStreamPool[device];
        if (!pool.empty())
        {
            *s = pool.back();
            pool.pop_back();
            return hipSuccess;
        }
    }
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

// (the caller has synchronized the stream, or never used it)
void stream_give(int device, hipStream_t s)
{
    if (!s) return;
    if (!stream_pool_on())
    {
        (void) hipStreamDestroy(s);
        return;
    }
    std::lock_guard<std::mutex> g(gStreamPoolMutex);
    gStreamPool[device].push_back(s);
}

const RoctxApi *roctx_api()
{
    static const RoctxApi *api = []() -> const RoctxApi *
    {
        if (!(std::getenv("HCV_ROCTX") && std::atoi(std::getenv("HCV_ROCTX")))) return nullptr;
        static RoctxApi a;
        for (const char *lib : { "librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4" })
            if (void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL))
            {
                a.push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
                a.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (a.push && a.pop) return &a;
            }
        return nullptr;
    }();
    return api;
}

long long order_violations() { return order_mode() > 0 ? order_registry().violations.load() : -1; }

Engine *Engine::create(const EngineCfg &cfg, std::string *err)
{
    Engine *e = new Engine();
    if (!e->init(cfg))
    {
        if (err) *err = e->mErr;
        delete e;
        return nullptr;
    }
    return e;
}

bool Engine::init(const EngineCfg &cfg)
{
    mCfg = cfg;
    {
        static std::atomic<unsigned> engines { 0 };
        mPinXcd = (int) (engines.fetch_add(1, std::memory_order_relaxed) & 7u);
    }
    if (mCfg.nin < 1) mCfg.nin = 1;
    if (mCfg.nout < 1)
    {
        mErr = "engine needs at least one output";
        return false;
    }
    if (mCfg.diag) mCfg.nin = mCfg.nout;
    mNinAlloc = mCfg.diag ? 1 : mCfg.nin;

    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
    {
        mErr = "no HIP device available (the convolution engine has no CPU fallback)";
        if (e != hipSuccess) mErr += std::string(": ") + hipGetErrorString(e);
        return false;
    }
    if (cfg.device >= 0)
        mDevice = cfg.device;
    else if (const char *env = std::getenv("HCV_DEVICE"))
        mDevice = std::atoi(env);
    else
        HCV_TRY(hipGetDevice(&mDevice));
    if (mDevice < 0 || mDevice >= count)
    {
        mErr = "HIP device index out of range";
        return false;
    }
    DeviceGuard dg(mDevice);
    {
        // every kernel the engine may launch from the audio thread is resident on this device before the first block (hcv_kernels.h: preload_*)
        static std::atomic<bool> loaded[64];
        if (mDevice < 64 && !loaded[mDevice].exchange(true, std::memory_order_acq_rel))
        {
            preload_kernels();
            preload_mac();
            preload_mac_tiled();
            preload_mac_mfma();
            preload_ghost();
            preload_bigfft();
            preload_fft_split();
            preload_fused_nxm();
        }
    }

    mMaxBlock = cfg.max_block;
    if (!mMaxBlock)
    {
        const char *env = std::getenv("HCV_MAX_BLOCK");
        mMaxBlock = env ? (uint32_t) std::atoi(env) : 32768u;
    }
    if (mMaxBlock < 16) mMaxBlock = 16;

    if (mCfg.stages.size() > (size_t) kMaxStages)
    {
        mErr = "too many FFT stages";
        return false;
    }
    uint32_t nmax = 0;
    for (const StageCfg &sc : mCfg.stages)
    {
        int l2 = ilog2(sc.fft_size);
        if ((uint64_t(1) << l2) != sc.fft_size || l2 < kMinFFTLog2 || l2 > kMaxFFTLog2)
        {
            mErr = "invalid FFT size";
            return false;
        }
        nmax = std::max(nmax, sc.fft_size);
    }

    HCV_TRY(stream_take(mDevice, &mStream));
    mOneStream = one_stream_mode(mCfg);
    if (mOneStream)
        mTdStream = mInStream = mStream;
    else
    {
        HCV_TRY(stream_take(mDevice, &mTdStream));
        HCV_TRY(stream_take(mDevice, &mInStream));
    }
    for (int k = 0; k < 2; k++)
    {
        HCV_TRY(hipEventCreateWithFlags(&mEvInput[k], hipEventDisableTiming));
        HCV_TRY(hipEventCreateWithFlags(&mEvTd[k], hipEventDisableTiming));
        HCV_TRY(hipEventCreateWithFlags(&mEvEmit[k], hipEventDisableTiming));
    }
    HCV_TRY(hipEventCreateWithFlags(&mEvCtl, hipEventDisableTiming));
    HCV_TRY(hipEventCreateWithFlags(&mEvSerial, hipEventDisableTiming));
    // control work that must not delay the audio thread (IR upload + FFTs into staging, capacity growth) has its own stream
    HCV_TRY(stream_take(mDevice, &mCtlStream));
    HCV_TRY(hipEventCreateWithFlags(&mEvSwapDone, hipEventDisableTiming));
    HCV_TRY(hipEventCreateWithFlags(&mEvSnap, hipEventDisableTiming));
    HCV_TRY(hipEventCreateWithFlags(&mEvHostDone, hipEventDisableTiming));
    HCV_TRY(stream_take(mDevice, &mPipeStream));
    for (int k = 0; k < 2; k++) HCV_TRY(hipEventCreateWithFlags(&mEvPipe[k], hipEventDisableTiming));
    for (int k = 0; k < 4; k++) HCV_TRY(hipEventCreateWithFlags(&mEvPipeEnd[k], hipEventDisableTiming));
    HCV_TRY(hipEventCreateWithFlags(&mEvFwd, hipEventDisableTiming));
    for (int k = 0; k < 4; k++) HCV_TRY(hipEventCreateWithFlags(&mEvNxmEnd[k], hipEventDisableTiming));
    HCV_TRY(hipEventRecord(mEvSwapDone, mStream));

    // three blocks deep: block k+1 is scattered while block k-1's readers may still be running
    mHistLen = pow2ceil(3LL * mMaxBlock + std::max<long long>(nmax, 4096));
    const size_t pairs = (size_t) mCfg.nout * mNinAlloc;

    HCV_TRY(hipMalloc(&mHist, sizeof(float) * mCfg.nin * mHistLen));
    HCV_TRY(hipMemset(mHist, 0, sizeof(float) * mCfg.nin * mHistLen));
    // (retire_pair's frame, allocated here so that no swap section in an owner section ever allocates)
    if (nmax) HCV_TRY(hipMalloc(&mRetireTmp, sizeof(float) * nmax));
    HCV_TRY(hipMalloc(&mDevIn, sizeof(float) * mCfg.nin * mMaxBlock));
    HCV_TRY(hipMalloc(&mDevOut, sizeof(float) * mCfg.nout * mMaxBlock));
    HCV_TRY(hipHostMalloc(&mPinIn, sizeof(float) * mCfg.nin * mMaxBlock, hipHostMallocMapped));
    HCV_TRY(hipHostMalloc(&mPinOut, sizeof(float) * mCfg.nout * mMaxBlock, hipHostMallocMapped));
    if (hipHostGetDevicePointer((void **) &mPinInDev, mPinIn, 0) != hipSuccess || hipHostGetDevicePointer((void **) &mPinOutDev, mPinOut, 0) != hipSuccess)
    {
        (void) hipGetLastError();
        mPinInDev = mPinOutDev = nullptr;                       // no mapping: every block takes the copy path
    }
    if (mCfg.has_td)
    {
        for (int k = 0; k < 2; k++) HCV_TRY(hipMalloc(&mTdOut[k], sizeof(float) * mCfg.nout * mMaxBlock));
        HCV_TRY(hipMalloc(&mTaps, sizeof(float) * pairs * 2048));
        HCV_TRY(hipMemset(mTaps, 0, sizeof(float) * pairs * 2048));
        HCV_TRY(hipMalloc(&mTdValid, sizeof(long long) * pairs));
        HCV_TRY(hipMemset(mTdValid, 0, sizeof(long long) * pairs));
        HCV_TRY(hipMalloc(&mStageTaps, sizeof(float) * 2048));
        mTdCount.assign(pairs, 0);
    }
    mPending.assign(pairs, 0);
    mLoaded.assign(pairs, 0);
    mRetired.assign(pairs, 0);
    mGhostOf.assign(pairs, nullptr);

    // Whole-hop mode (see enqueue_chunk): needs the zero-latency ladder — head at [0, a), every shorter stage continuing
    // where the previous coverage ends, the last stage starting exactly one of its hops in.  The last stage then keeps the
    // spectrum of IR[0 : its hop) in a lead slot in front of every pair's partitions (Stage::lead).
    // With far-tail rungs behind it (EngineCfg::pivot, the extended ladder) the stage in front of them takes that role: whole-hop blocks
    // are whole hops of THAT stage, and the rungs keep their own schedule beside it (enqueue_chunk).
    mPivot = mCfg.stages.empty() ? 0 : mCfg.stages.size() - 1;
    if (mCfg.pivot >= 0 && (size_t) mCfg.pivot < mCfg.stages.size()) mPivot = (size_t) mCfg.pivot;
    if (mCfg.stages.size() >= 2 || (mCfg.has_td && !mCfg.stages.empty()))
    {
        static const bool allow = !(std::getenv("HCV_TAIL_HEAD") && std::atoi(std::getenv("HCV_TAIL_HEAD")) == 0);
        uint64_t end = 0;
        bool ok = allow;
        if (mCfg.has_td)
        {
            ok = ok && mCfg.td_offset == 0 && mCfg.td_length > 0;
            end = mCfg.td_length;
        }
        for (size_t k = 0; ok && k < mPivot; k++)
        {
            const StageCfg &sc = mCfg.stages[k];
            ok = sc.offset == end && sc.length > 0;
            end = sc.offset + sc.length;
        }
        const StageCfg &tl = mCfg.stages[mPivot];
        ok = ok && tl.offset == end && tl.offset == tl.fft_size / 2 && !is_big_fft(ilog2(tl.fft_size));
        mTailHead = mLeadSlot = ok;
        if (!ok) mPivot = mCfg.stages.size() - 1;
    }
    else if (mCfg.stages.size() == 1 && !mCfg.has_td && !is_big_fft(ilog2(mCfg.stages[0].fft_size)))
    {
        // a lone FFT stage (a PartitionedConvolve): the same block structure without a lead slot — hop-aligned blocks compute
        // exactly the hops they emit (the partitions' one hop of latency is taken by evaluating hop h - 1 in block h)
        static const bool allow = !(std::getenv("HCV_TAIL_HEAD") && std::atoi(std::getenv("HCV_TAIL_HEAD")) == 0);
        mTailHead = allow;
    }

    for (const StageCfg &sc : mCfg.stages)
    {
        Stage *st = new Stage();
        st->cfg = sc;
        st->lead = (mLeadSlot && &sc == &mCfg.stages[mPivot]) ? 1 : 0;
        st->log2n = ilog2(sc.fft_size);
        st->N = sc.fft_size;
        st->M = sc.fft_size / 2;
        st->pact.assign(pairs, 0);
        mStages.push_back(st);
        st->tw = twiddles(mDevice, st->log2n, &mErr);
        if (!st->tw) return false;
        fft_split_prepare(st->log2n);               // (tables of the residue-split transforms, hcv_fft_split.hip)
        if (!alloc_stage(*st)) return false;
    }
    if (mLeadSlot) HCV_TRY(hipMalloc(&mStageTailHead, sizeof(float2) * mStages[mPivot]->M));
    // The streams that carry a step side by side, on hardware queues chosen for them (hcv_queue_probe.hip; nothing is enqueued on them yet).
    // An extended ladder has five busy streams for the process's four queues: the pivot stage's two lanes and the first rung get queues of
    // their own, the LAST rung the main stream's — whose emit launches then wait behind that rung's long multiply-accumulate slices now and
    // then.  Measured, c5 on the ladder over 128 steps, ms per step: that pairing 0.116 - 0.119 (what a process's second engine got by
    // itself); emit with the first lane — what a process's first engine got — 0.124 - 0.139: emit (k), waiting for the other lane's block,
    // holds the first lane's block k + 1 back; emit alone (eight queues, or a stream of high priority: another pool of queues) 0.130 - 0.150.
    // An engine whose one-hop block is the n x m pair of launches: the forward launch's stream not with the main one.
    if (!mOneStream && !mStages.empty())
    {
        Stage &pv = *mStages[mPivot];
        std::vector<hipStream_t *> roles;
        int share = -1;
        if (pv.stream2)
        {
            roles.push_back(&pv.stream);
            roles.push_back(&pv.stream2);
            for (size_t k = mPivot + 1; k < mStages.size() && roles.size() < 4; k++) roles.push_back(&mStages[k]->stream);
            share = (int) roles.size() - 1;
        }
        else if (pv.nxm_helped)
            roles.push_back(&mPipeStream);
        // (round 6: a streamed tail — a GB-class store of spectra — whose hop-sized HOST-pointer calls start the old partitions' multiply-accumulate
        // ahead of the upload, host_pre_mac: the upload is enqueued on the main stream, and in a hardware queue shared with the stage's stream
        // it would sit behind that launch instead of beside it)
        if (roles.empty() && pv.lead && mCfg.nout > 1 && !mCfg.diag &&
            sizeof(float2) * (double) mCfg.nout * mNinAlloc * pv.Pcap * pv.M >= 512.0 * 1048576.0)
            roles.push_back(&pv.stream);
        if (!roles.empty()) mStreamsSpread = spread_streams(mStream, roles.data(), (int) roles.size(), share);
    }
    if (order_mode() > 0)
    {
        // (HCV_ORDER_CHECK: this engine's streams by name; from here on every record / wait / declared access on them is followed)
        mOrd = new OrderCheck();
        order_register(mOrd, mStream, "main");
        order_register(mOrd, mTdStream, "head");
        order_register(mOrd, mInStream, "input");
        order_register(mOrd, mCtlStream, "control");
        order_register(mOrd, mPipeStream, "pipe");
        static const char *stage_names[] = {"stage 0", "stage 1", "stage 2", "stage 3", "stage 4", "stage 5", "stage 6", "stage 7"};
        for (size_t k = 0; k < mStages.size() && k < 8; k++)
        {
            order_register(mOrd, mStages[k]->stream, stage_names[k]);
            if (mStages[k]->stream2) order_register(mOrd, mStages[k]->stream2, "pivot lane 2");
        }
    }
    // Head through the FFT: the head's taps (<= one hop of the first FFT stage, MonoConvolve.cpp:235-240) form one extra,
    // zero-latency partition of that stage, whose input spectra exist anyway.  Used for hop-aligned blocks of larger
    // matrices, where the direct-form FIR would cost more than all FFT-stage MACs together; ragged blocks and small
    // engines keep the direct form (fir_head_kernel).
    if (mCfg.has_td && !mStages.empty())
    {
        const Stage &s0 = *mStages[0];
        const uint64_t lim = mCfg.td_length ? mCfg.td_length : 2044;
        if (lim <= s0.M && !is_big_fft(s0.log2n) && (size_t) mCfg.nout * mNinAlloc >= 16)
        {
            mHeadFFT = true;
            HCV_TRY(hipMalloc(&mHeadSpec, sizeof(float2) * pairs * s0.M));
            HCV_TRY(hipMemset(mHeadSpec, 0, sizeof(float2) * pairs * s0.M));
            for (int k = 0; k < 2; k++) HCV_TRY(hipMalloc(&mHeadYq[k], sizeof(float2) * (size_t) s0.Tmax * mCfg.nout * s0.M));
            HCV_TRY(hipMalloc(&mStageHead, sizeof(float2) * s0.M));
        }
    }
    {
        // what this engine wants held in the device's control arena (see CtlArena): its largest stage regrown by half — new spectra and
        // ring beside the old ones — plus a pair's staging and IR upload; 16 MiB .. 1 GiB.  hcv_ctl_reserve / HCV_CTL_RESERVE_MB for more.
        size_t most = 0;
        for (const Stage *st : mStages)
            most = std::max(most, sizeof(float2) * ((size_t) mCfg.nout * mNinAlloc * (st->Pcap + st->lead) + (size_t) mCfg.nin * st->R + 3 * (size_t) st->Pcap) * st->M);
        mArenaWant = (long long) std::min<size_t>(size_t(1) << 30, std::max<size_t>(size_t(16) << 20, most + most / 2));
        arena_engine(mDevice, mArenaWant);
    }
    HCV_TRY(hipDeviceSynchronize());
    return true;
}

bool Engine::alloc_stage(Stage &st)
{
    const size_t pairs = (size_t) mCfg.nout * mNinAlloc;
    uint64_t cap = st.cfg.capacity ? st.cfg.capacity : st.M;
    st.Pcap = (uint32_t) std::max<uint64_t>(1, (cap + st.M - 1) / st.M);
    st.Tmax = mMaxBlock / st.M + 1;
    // (the n x m block's forward launches run ahead of the multiply-accumulates on a stream of their own, held back by an event only
    // every few blocks: four more ring slots let that be every sixth block instead of every third — enqueue_chunk)
    st.ring_extra = (mCfg.nout > 1 && !mCfg.diag && st.log2n == 14 && st.lead) ? 4 : 0;
    st.R = st.Pcap + 2 * st.Tmax + st.ring_extra;      // FFT of block k+1 may write while the MAC of block k still reads
    const size_t hs_elems = pairs * st.hstride();
    const size_t x_elems = (size_t) mCfg.nin * st.R * st.M;
    // room for split-K partials: up to 64 slices for short spectra, fewer as the bin axis alone fills the chip
    const uint32_t split_cap = std::max<uint32_t>(1, std::min<uint32_t>(64, 2048u / std::max<uint32_t>(1, st.M / 512)));
    st.y_elems = (size_t) std::max<uint32_t>(st.Tmax, split_cap) * mCfg.nout * st.M;
    HCV_TRY(hipMalloc(&st.Hs, sizeof(float2) * hs_elems));
    HCV_TRY(hipMemset(st.Hs, 0, sizeof(float2) * hs_elems));
    HCV_TRY(hipMalloc(&st.X, sizeof(float2) * x_elems));
    HCV_TRY(hipMemset(st.X, 0, sizeof(float2) * x_elems));
    for (int k = 0; k < 2; k++) HCV_TRY(hipMalloc(&st.Yq[k], sizeof(float2) * st.y_elems));
    st.Y = st.Yq[0];
    HCV_TRY(hipMalloc(&st.hv, sizeof(long long) * pairs));
    HCV_TRY(hipMemset(st.hv, 0, sizeof(long long) * pairs));
    // hand-over state of the fused blocks: one-output engines (hcv_fft_split.hip), and the lead-slot stage of a matrix with several
    // outputs (hcv_fused_nxm.hip: more multiply-accumulate tasks)
    const bool coop_one = mCfg.nout == 1 && (st.log2n == 14 || st.log2n == 12);
    const bool coop_nxm = mCfg.nout > 1 && !mCfg.diag && st.log2n == 14 && st.lead;
    if (coop_one || coop_nxm)
    {
        const size_t marks = (size_t) (coop_nxm ? 0 : kFusedMacTasks) + kFusedFwdTasks;
        const size_t counters = coop_nxm ? (size_t) kFusedShards * kFusedShardStride : 2;           // (the n x m block's counter is sharded)
        HCV_TRY(hipMalloc(&st.coop_bar, sizeof(unsigned) * counters));
        HCV_TRY(hipMemset(st.coop_bar, 0, sizeof(unsigned) * counters));
        HCV_TRY(hipMalloc(&st.coop_flags, sizeof(unsigned long long) * marks));
        HCV_TRY(hipMemset(st.coop_flags, 0, sizeof(unsigned long long) * marks));
        if (coop_nxm)
        {
            HCV_TRY(hipHostMalloc((void **) &st.nxm_helped, sizeof(unsigned), hipHostMallocMapped));
            *st.nxm_helped = 0;
            if (hipHostGetDevicePointer((void **) &st.nxm_helped_dev, st.nxm_helped, 0) != hipSuccess)
            {
                (void) hipGetLastError();
                st.nxm_helped_dev = nullptr;
            }
        }
    }
    HCV_TRY(hipMalloc(&st.Ypre, sizeof(float2) * (size_t) (kBgSlices + kBoundarySlices) * mCfg.nout * st.M));
    HCV_TRY(hipEventCreateWithFlags(&st.bg_done, hipEventDisableTiming));
    if (is_big_fft(st.log2n))
    {
        int l1, l2;
        big_fft_split(st.log2n, l1, l2);
        st.big.tw1 = twiddles(mDevice, l1 + 1, &mErr);
        st.big.tw2 = twiddles(mDevice, l2 + 1, &mErr);
        if (!st.big.tw1 || !st.big.tw2) return false;
        const size_t batch = std::min<size_t>(32, (size_t) st.Tmax * std::max(mCfg.nin, mCfg.nout));
        st.big.elems = batch * st.M;
        HCV_TRY(hipMalloc(&st.big.a, sizeof(float2) * st.big.elems));
        HCV_TRY(hipMalloc(&st.big.b, sizeof(float2) * st.big.elems));
        st.big_ctl = st.big;                                    // the control stream transforms IRs beside the audio streams
        HCV_TRY(hipMalloc(&st.big_ctl.a, sizeof(float2) * st.big.elems));
        HCV_TRY(hipMalloc(&st.big_ctl.b, sizeof(float2) * st.big.elems));
    }
    st.tl_len = pow2ceil(2LL * mMaxBlock + st.M);       // two blocks deep (block k+1 adds while block k is emitted)
    HCV_TRY(hipMalloc(&st.timeline, sizeof(float) * mCfg.nout * st.tl_len));
    HCV_TRY(hipMemset(st.timeline, 0, sizeof(float) * mCfg.nout * st.tl_len));
    if (mOneStream)
        st.stream = mStream;
    else
    {
        HCV_TRY(stream_take(mDevice, &st.stream));
        // (the second lane of an extended ladder's pivot stage: created where the layout has rungs behind this stage)
        static const bool two_lanes = !(std::getenv("HCV_PIVOT_LANES") && std::atoi(std::getenv("HCV_PIVOT_LANES")) == 1);
        if (two_lanes && st.lead && mCfg.pivot >= 0 && (size_t) mCfg.pivot + 1 < mCfg.stages.size()) HCV_TRY(stream_take(mDevice, &st.stream2));
    }
    for (int k = 0; k < 2; k++)
    {
        HCV_TRY(hipEventCreateWithFlags(&st.mac_done[k], hipEventDisableTiming));
    }
    for (int k = 0; k < 2; k++) HCV_TRY(hipEventCreateWithFlags(&st.done[k], hipEventDisableTiming));
    return true;
}

void Engine::free_stage(Stage &st)
{
    // (what the control path allocated goes back the way it came: stream-ordered memory is not mixed with hipFree)
    if (st.hs_ctl)
    {
        ctl_free(st.Hs);
        ctl_free(st.X);
    }
    else
    {
        if (st.Hs) (void) hipFree(st.Hs);
        if (st.X) (void) hipFree(st.X);
    }
    for (int k = 0; k < 2; k++)
    {
        if (st.Yq[k]) (void) hipFree(st.Yq[k]);
        if (st.mac_done[k]) (void) hipEventDestroy(st.mac_done[k]);
        st.Yq[k] = nullptr;
        st.mac_done[k] = nullptr;
    }
    if (st.hv) (void) hipFree(st.hv);
    if (st.coop_bar) (void) hipFree(st.coop_bar);
    if (st.coop_flags) (void) hipFree(st.coop_flags);
    if (st.nxm_helped) (void) hipHostFree(st.nxm_helped);
    st.nxm_helped = st.nxm_helped_dev = nullptr;
    if (st.gh_start) (void) hipFree(st.gh_start);
    if (st.gh_ent) (void) hipFree(st.gh_ent);
    st.gh_start = nullptr;
    st.gh_ent = nullptr;
    if (st.timeline) (void) hipFree(st.timeline);
    if (st.Ypre) (void) hipFree(st.Ypre);
    if (st.bg_done) (void) hipEventDestroy(st.bg_done);
    st.Ypre = nullptr;
    st.bg_done = nullptr;
    if (st.big.a) (void) hipFree(st.big.a);
    if (st.big.b) (void) hipFree(st.big.b);
    if (st.big_ctl.a) (void) hipFree(st.big_ctl.a);
    if (st.big_ctl.b) (void) hipFree(st.big_ctl.b);
    if (st.stage_spec) ctl_free(st.stage_spec);
    st.stage_spec = nullptr;
    st.big.a = st.big.b = st.big_ctl.a = st.big_ctl.b = nullptr;
    for (int k = 0; k < 2; k++)
        if (st.done[k]) (void) hipEventDestroy(st.done[k]);
    if (st.stream && st.stream != mStream) stream_give(mDevice, st.stream);
    if (st.stream2) stream_give(mDevice, st.stream2);
    st.stream2 = nullptr;
    st.Hs = st.X = st.Y = nullptr;
    st.stream = nullptr;
    st.hv = nullptr;
    st.timeline = nullptr;
    st.done[0] = st.done[1] = nullptr;
    st.stream = nullptr;
}

Engine::~Engine()
{
    DeviceGuard dg(mDevice);
    if (mOrd)
    {
        order_unregister(mOrd);
        delete mOrd;
        mOrd = nullptr;
    }
    if (mCtlStream) (void) hipStreamSynchronize(mCtlStream);
    if (mPipeStream) (void) hipStreamSynchronize(mPipeStream);
    if (mInStream) (void) hipStreamSynchronize(mInStream);
    if (mTdStream) (void) hipStreamSynchronize(mTdStream);
    for (Stage *st : mStages)
    {
        if (st->stream) (void) hipStreamSynchronize(st->stream);
        if (st->stream2) (void) hipStreamSynchronize(st->stream2);
    }
    if (mStream) (void) hipStreamSynchronize(mStream);
    for (void *p : mParked) (void) hipFree(p);
    drop_ghosts();
    for (auto &blk : mGhostPool) (void) hipFree(blk.second);
    for (Stage *st : mStages)
    {
        free_stage(*st);
        delete st;
    }
    for (EventPair *ev : mEvents)
    {
        if (ev->a) (void) hipEventDestroy(ev->a);
        if (ev->b) (void) hipEventDestroy(ev->b);
        delete ev;
    }
    if (mHist) (void) hipFree(mHist);
    for (int k = 0; k < 2; k++)
        if (mTdOut[k]) (void) hipFree(mTdOut[k]);
    if (mDevIn) (void) hipFree(mDevIn);
    if (mDevOut) (void) hipFree(mDevOut);
    if (mPinIn) (void) hipHostFree(mPinIn);
    if (mPinOut) (void) hipHostFree(mPinOut);
    if (mIrBuf) ctl_free(mIrBuf);
    if (mCtlStream) (void) hipStreamSynchronize(mCtlStream);        // the stream-ordered frees above have run
    // (... and the arena blocks parked behind events recorded on that stream go back NOW: an event must not outlive the stream it was
    // recorded on — a later query reaches into the destroyed stream's signal pool.  Suspected in two full-suite runs that died inside
    // the create / destroy cycles of tests/test_gpu_parity.py, one with an abort in a destructor, one with a fault on a runtime thread.)
    arena_reap_all(mDevice);
    if (mArenaWant)
    {
        arena_engine(mDevice, -mArenaWant);      // (its want goes with it; chunks nothing lives in and nobody wants go back to the driver)
        mArenaWant = 0;
    }
    if (mTaps) (void) hipFree(mTaps);
    if (mHeadSpec) (void) hipFree(mHeadSpec);
    for (int k = 0; k < 2; k++)
        if (mHeadYq[k]) (void) hipFree(mHeadYq[k]);
    if (mTdValid) (void) hipFree(mTdValid);
    for (int k = 0; k < 2; k++)
    {
        if (mEvInput[k]) (void) hipEventDestroy(mEvInput[k]);
        if (mEvTd[k]) (void) hipEventDestroy(mEvTd[k]);
        if (mEvEmit[k]) (void) hipEventDestroy(mEvEmit[k]);
    }
    if (mEvCtl) (void) hipEventDestroy(mEvCtl);
    if (mEvSerial) (void) hipEventDestroy(mEvSerial);
    if (mEvSwapDone) (void) hipEventDestroy(mEvSwapDone);
    if (mEvSnap) (void) hipEventDestroy(mEvSnap);
    if (mEvHostDone) (void) hipEventDestroy(mEvHostDone);
    if (mEvFwd) (void) hipEventDestroy(mEvFwd);
    for (int k = 0; k < 4; k++)
        if (mEvNxmEnd[k]) (void) hipEventDestroy(mEvNxmEnd[k]);
    if (mStageTaps) (void) hipFree(mStageTaps);
    if (mStageHead) (void) hipFree(mStageHead);
    if (mStageTailHead) (void) hipFree(mStageTailHead);
    if (mCtlStream) stream_give(mDevice, mCtlStream);
    if (mPipeStream) stream_give(mDevice, mPipeStream);
    for (int k = 0; k < 2; k++)
    {
        if (mEvPipe[k]) (void) hipEventDestroy(mEvPipe[k]);
    }
    for (int k = 0; k < 4; k++)
        if (mEvPipeEnd[k]) (void) hipEventDestroy(mEvPipeEnd[k]);
    if (mGhostHist) (void) hipFree(mGhostHist);
    if (mRetireTmp) (void) hipFree(mRetireTmp);
    if (mGhostPin) (void) hipHostFree(mGhostPin);
    if (mGhostUploaded) (void) hipEventDestroy(mGhostUploaded);
    if (mInStream && mInStream != mStream) stream_give(mDevice, mInStream);
    if (mTdStream && mTdStream != mStream) stream_give(mDevice, mTdStream);
    if (mStream) stream_give(mDevice, mStream);
}

uint64_t Engine::stage_capacity(size_t s) const
{
    return s < mStages.size() ? (uint64_t) mStages[s]->Pcap * mStages[s]->M : 0;
}

uint32_t Engine::stage_partitions(size_t s, uint32_t in, uint32_t out) const
{
    if (s >= mStages.size() || out >= mCfg.nout || (!mCfg.diag && in >= mCfg.nin)) return 0;
    return mStages[s]->pact[pair_index(in, out)];
}

uint32_t Engine::td_taps(uint32_t in, uint32_t out) const
{
    if (!mCfg.has_td || out >= mCfg.nout || (!mCfg.diag && in >= mCfg.nin)) return 0;
    return mTdCount[pair_index(in, out)];
}

void Engine::set_stage_window(size_t s, uint64_t offset, uint64_t length)
{
    std::lock_guard<std::mutex> g(mSetMutex);          // (the windows are read by set_ir only)
    if (s < mStages.size())
    {
        mStages[s]->cfg.offset = offset;
        mStages[s]->cfg.length = length;
    }
}

void Engine::set_td_window(uint64_t offset, uint64_t length)
{
    std::lock_guard<std::mutex> g(mSetMutex);
    mCfg.td_offset = offset;
    mCfg.td_length = length;
}

// Control work (IR loads, resets, regrow) changes what a background accumulation reads or means: order it after any
// background MAC still in flight and drop the pre-accumulated spectra — unless the caller keeps the plan and corrects them
// itself (a restart of single pairs: retire_pair takes the pair out of the slices already accumulated).  Caller holds mMutex.
// The boundary chains of a small block run on past its emit (enqueue_stage): whoever puts work on the main stream next — the next
// block, control work — orders the main stream behind them first.  Caller holds mMutex.
bool Engine::fence_chains(bool keep_forward)
{
    for (Stage *st : mStages)
        if (st->chain_pending >= 0)
        {
            HCV_TRY(hipStreamWaitEvent(mStream, st->done[st->chain_pending], 0));
            st->chain_pending = -1;
        }
    if (!keep_forward && !join_forward_stream()) return false;
    return true;
}

// The forward launches of n x m fused blocks (hcv_fused_nxm.hip) sit on the pipe stream with no event towards the main stream: the
// multiply-accumulate launch waits for their arrival counter, or does their work itself.  Physically they are through when that
// launch is, unless the device scheduled one late and the helping path stood in for it — such a launch would still write its spectrum
// and the ring's hop (the same values) whenever it runs.  So before anything else reads or writes the rings on the main stream — a
// block of another kind, control work, a reset's clearing — the main stream goes behind the pipe stream, formally.  Caller holds mMutex.
bool Engine::join_forward_stream()
{
    if (!mFwdPending) return true;
    HCV_TRY(hipEventRecord(mEvFwd, mPipeStream));
    HCV_TRY(hipStreamWaitEvent(mStream, mEvFwd, 0));
    mFwdPending = false;
    mPrevNxm = false;
    return true;
}

bool Engine::fence_background(bool keep_plan)
{
    if (!fence_chains()) return false;
    for (Stage *st : mStages)
    {
        if (st->bg_pending)
        {
            HCV_TRY(hipStreamWaitEvent(mStream, st->bg_done, 0));
            st->bg_pending = false;
        }
        if (!keep_plan) st->pre_hop = -1;
    }
    return true;
}

// Capacity growth of a stage (MonoConvolve::resize / set(..., requestResize) reallocate the tail partition,
// MonoConvolve.cpp:101-110,123; here all pairs of a stage share one allocation, so growing re-strides it).
//
// The MemorySwap idea (MemorySwap.h:187-229) applied to the whole stage: the new buffers are allocated and filled on the
// control stream while the audio thread keeps processing on the old ones; only the pointer swap — plus a catch-up copy of
// the few ring slots written meanwhile — happens in an owner section.
bool Engine::ensure_stage_capacity(size_t s, uint64_t capacity)
{
    std::lock_guard<std::mutex> gs(mSetMutex);
    if (s >= mStages.size()) return false;
    DeviceGuard dg(mDevice);
    Stage &st = *mStages[s];
    RoctxRange range("hcv:resize");
    uint32_t newP = (uint32_t) std::max<uint64_t>(1, (capacity + st.M - 1) / st.M);
    if (newP <= st.Pcap) return true;
    // grow by at least half: obtaining NEW device memory from the driver stalls every HIP call of the process while it maps
    // (one 19 ms `process` call observed beside a regrow to 0.5 GB) — the pool keeps what it has, so make such growth rare
    newP = std::max<uint32_t>(newP, st.Pcap + st.Pcap / 2);

    // ---- outside any owner section: allocate, clear, re-stride what exists
    const size_t pairs = (size_t) mCfg.nout * mNinAlloc;
    const uint32_t newR = newP + 2 * st.Tmax + st.ring_extra;
    float2 *nHs = nullptr, *nX = nullptr;
    const size_t hs_bytes = sizeof(float2) * pairs * (newP + st.lead) * st.M;
    const size_t x_bytes = sizeof(float2) * (size_t) mCfg.nin * newR * st.M;
    if (ctl_alloc((void **) &nHs, hs_bytes) != hipSuccess) return false;
    if (ctl_alloc((void **) &nX, x_bytes) != hipSuccess)
    {
        ctl_free(nHs);
        return false;
    }
    long long h_snap = 0;
    uint32_t live_P = 0;
    // a snapshot of the hop clock; everything the blocks so far have enqueued is ordered before the copies below
    (void) run_exclusive([&]()
    {
        h_snap = mN / st.M;
        live_P = st.P;
        // (the main stream no longer waits for a small block's boundary chains in the call that enqueues them — enqueue_stage, `late` —
        // and the mailbox runs this section in front of the next call's fence: put the main stream behind them first, or the copy
        // below would read the ring slot X[h_snap - 1] the chain's forward transform is still writing)
        if (!fence_chains()) { (void) hipGetLastError(); }
        if (hipEventRecord(mEvSnap, mStream) != hipSuccess) { (void) hipGetLastError(); }
        return true;
    });
    bool ok = hipStreamWaitEvent(mCtlStream, mEvSnap, 0) == hipSuccess && hipStreamWaitEvent(mCtlStream, mEvSwapDone, 0) == hipSuccess;
    ok = ok && hipMemsetAsync(nHs, 0, hs_bytes, mCtlStream) == hipSuccess && hipMemsetAsync(nX, 0, x_bytes, mCtlStream) == hipSuccess;
    if (ok && (live_P > 0 || st.lead))
    {
        // (the spectra are only ever written by control calls, which mSetMutex serialises with this one)
        ok = launch_regrow_spectra(st.Hs, nHs, (long long) pairs, st.hparts(), (int) (newP + st.lead), (int) st.M, mCtlStream) == hipSuccess;
        const int live = (int) std::min<long long>(live_P, h_snap);
        if (ok && live > 0)
            ok = launch_regrow_ring(st.X, nX, (int) mCfg.nin, (int) st.R, (int) newR, (int) st.M, h_snap - 1, live, mCtlStream) == hipSuccess;
    }
    ok = ok && hipStreamSynchronize(mCtlStream) == hipSuccess;
    if (!ok)
    {
        (void) hipGetLastError();
        ctl_free(nHs);
        ctl_free(nX);
        mErr = "stage regrow failed";
        return false;
    }

    // ---- in an owner section: the hops that arrived since the snapshot, then the pointer swap (no allocation, no wait)
    float2 *oHs = nullptr, *oX = nullptr;
    const bool old_ctl = st.hs_ctl;
    (void) run_exclusive([&]()
    {
        if (!fence_background()) ok = false;
        const long long h_now = mN / st.M;
        if (ok && st.P > 0 && h_now < h_snap)
        {
            // a global reset restarted the hop clock meanwhile: the ring copy is stale, the new clock's hops are what counts
            ok = hipMemsetAsync(nX, 0, x_bytes, mStream) == hipSuccess;
            h_snap = 0;
        }
        if (ok && st.P > 0 && h_now > h_snap)
        {
            const int late = (int) std::min<long long>(h_now - h_snap, (long long) st.R);
            ok = launch_regrow_ring(st.X, nX, (int) mCfg.nin, (int) st.R, (int) newR, (int) st.M, h_now - 1, late, mStream) == hipSuccess;
        }
        if (ok)
        {
            oHs = st.Hs;
            oX = st.X;
            st.Hs = nHs;
            st.X = nX;
            st.Pcap = newP;
            st.R = newR;
            mCtlDirty = true;
            ok = hipEventRecord(mEvSnap, mStream) == hipSuccess;       // the old buffers' last readers are behind this
        }
        return true;
    });
    if (!ok)
    {
        (void) hipGetLastError();
        ctl_free(nHs);
        ctl_free(nX);
        mErr = "stage regrow failed";
        return false;
    }
    // the old buffers' last readers are behind mEvSnap: the frees are ordered after it on the control stream (the first
    // buffers of a stage came from hipMalloc in init and are parked until the engine goes: hipFree would stall the device)
    (void) hipStreamWaitEvent(mCtlStream, mEvSnap, 0);
    if (old_ctl)
    {
        ctl_free(oHs);
        ctl_free(oX);
    }
    else
    {
        mParked.push_back(oHs);
        mParked.push_back(oX);
    }
    st.hs_ctl = true;
    return true;
}

// One private memory pool per device for the control path, shared by the engines on it: freed blocks stay in the pool (release
// threshold = everything) instead of going back to the driver at the next synchronisation, and the device's DEFAULT pool — other
// hipFreeAsync users of the process, PyTorch among them — keeps its own policy.
static hipMemPool_t ctl_pool(int device)
{
    static std::mutex mtx;
    static std::map<int, hipMemPool_t> pools;
    std::lock_guard<std::mutex> g(mtx);
    auto it = pools.find(device);
    if (it != pools.end()) return it->second;
    hipMemPool_t pool = nullptr;
    hipMemPoolProps props;
    std::memset(&props, 0, sizeof(props));
    props.allocType = hipMemAllocationTypePinned;
    props.handleTypes = hipMemHandleTypeNone;
    props.location.type = hipMemLocationTypeDevice;
    props.location.id = device;
    if (hipMemPoolCreate(&pool, &props) == hipSuccess && pool)
    {
        uint64_t keep = ~uint64_t(0);
        (void) hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    }
    else
        pool = nullptr;             // (allocation then falls back to the device's default pool, policy untouched)
    (void) hipGetLastError();
    pools[device] = pool;
    return pool;
}

// The control ARENA: device memory for the control path that is mapped BEFORE the stream it serves runs.  Obtaining new device memory
// from the driver — a regrown stage's spectra: up to a gigabyte in the contract test — stalls every HIP call of the process while the
// driver maps it: the audio thread's next launch stood behind that for 19 - 49 ms, whichever allocator asked (hipMalloc, or the stream-
// ordered pool when it held nothing that large yet).  So each device keeps an arena the control path's buffers — regrown spectra and
// rings, staging buffers, the IR upload buffer — are carved out of: first fit over coalescing free lists, host-only.
//
// Round 6: the arena is sized by the engines that use it, not 4 GiB for everyone.  Every engine states at its creation what
// it wants held for it — by default one regrow of its largest stage by half plus a pair's staging, 16 MiB .. 1 GiB (Engine::init) — and the
// device's arena grows by a chunk, THEN (a creation maps memory anyway), whenever the engines' wants together exceed what it holds; an
// engine's going takes its want back and releases the chunks nothing lives in and nobody wants; the last engine's going releases all.
// A host that will load longer impulse responses than its objects were created for reserves explicitly: hcv_ctl_reserve(device, bytes) /
// HCV_CTL_RESERVE_MB raise the floor of what the device holds while it has an engine (0 = the engines' wants only).
// A freed block goes back behind an event recorded on the freeing engine's control stream (ctl_free is stream-ordered like
// hipFreeAsync: the block's last users are ordered in front of that point) and is handed out again only once that event has
// completed, so any engine of the device may take it.  Requests the arena cannot serve fall back to the stream-ordered pool (and may
// stall every stream of the process while the driver maps them: what the reserve is for).
namespace
{
    struct CtlArena
    {
        struct Chunk
        {
            char *base = nullptr;
            size_t size = 0;
            std::map<size_t, size_t> free, live;                        // offset -> length
            void release(size_t off, size_t len)
            {
                auto it = free.emplace(off, len).first;
                auto nx = std::next(it);
                if (nx != free.end() && it->first + it->second == nx->first)
                {
                    it->second += nx->second;
                    free.erase(nx);
                }
                if (it != free.begin())
                {
                    auto pv = std::prev(it);
                    if (pv->first + pv->second == it->first)
                    {
                        pv->second += it->second;
                        free.erase(it);
                    }
                }
            }
            void *take(size_t bytes)
            {
                for (auto it = free.begin(); it != free.end(); ++it)
                    if (it->second >= bytes)
                    {
                        const size_t off = it->first, len = it->second;
                        free.erase(it);
                        if (len > bytes) free.emplace(off + bytes, len - bytes);
                        live.emplace(off, bytes);
                        return base + off;
                    }
                return nullptr;
            }
            bool holds(const void *p) const { return static_cast<const char *>(p) >= base && static_cast<const char *>(p) < base + size; }
        };
        std::mutex mtx;
        std::vector<Chunk *> chunks;
        struct Pending { void *p; hipEvent_t ev; };
        std::vector<Pending> pending;
        size_t wanted = 0;                                              // sum of the live engines' wants
        int engines = 0;

        size_t capacity() const
        {
            size_t c = 0;
            for (const Chunk *k : chunks) c += k->size;
            return c;
        }
        Chunk *chunk_of(const void *p) const
        {
            for (Chunk *k : chunks)
                if (k->holds(p)) return k;
            return nullptr;
        }
        void give_back(void *p)
        {
            if (Chunk *k = chunk_of(p))
            {
                auto lv = k->live.find((size_t) (static_cast<char *>(p) - k->base));
                if (lv != k->live.end())
                {
                    k->release(lv->first, lv->second);
                    k->live.erase(lv);
                }
            }
        }
        // blocks whose last users are through go back to the free lists; `wait`: the events still pending are handed to the caller, which
        // waits for them WITHOUT the arena's mutex (every other engine's ctl_alloc / ctl_free stood behind one engine's device work)
        void reap(std::vector<Pending> *still_pending)
        {
            for (size_t k = 0; k < pending.size();)
            {
                if (hipEventQuery(pending[k].ev) == hipErrorNotReady)
                {
                    (void) hipGetLastError();
                    if (still_pending) still_pending->push_back(pending[k]);
                    k++;
                    continue;
                }
                (void) hipGetLastError();
                (void) hipEventDestroy(pending[k].ev);
                give_back(pending[k].p);
                pending[k] = pending.back();
                pending.pop_back();
            }
        }
        void *take(size_t bytes)
        {
            bytes = (bytes + 255) & ~size_t(255);
            for (Chunk *k : chunks)
                if (void *p = k->take(bytes)) return p;
            return nullptr;
        }
        bool grow(size_t bytes)
        {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes > free_b / 2) bytes = free_b / 2;      // (never more than half of what is left)
            bytes &= ~size_t(255);
            void *p = nullptr;
            if (!bytes || hipMalloc(&p, bytes) != hipSuccess || !p)
            {
                (void) hipGetLastError();
                return false;
            }
            Chunk *k = new Chunk();
            k->base = static_cast<char *>(p);
            k->size = bytes;
            k->free.emplace(0, bytes);
            chunks.push_back(k);
            return true;
        }
        // chunks nothing lives in, while the rest still covers `floor`
        void shrink(size_t floor)
        {
            for (size_t k = chunks.size(); k-- > 0;)
            {
                Chunk *c = chunks[k];
                if (!c->live.empty() || capacity() - c->size < floor) continue;
                bool parked = false;
                for (const Pending &pe : pending) parked = parked || c->holds(pe.p);
                if (parked) continue;
                (void) hipFree(c->base);
                delete c;
                chunks.erase(chunks.begin() + (long) k);
            }
        }
    };
    std::mutex gArenaMutex;
    std::map<int, CtlArena *> gArenas;
    std::map<int, size_t> gArenaFloor;                                  // hcv_ctl_reserve: bytes the device holds at least while it has an engine

    size_t arena_floor(int device)
    {
        size_t f = 0;
        if (const char *env = std::getenv("HCV_CTL_RESERVE_MB")) f = (size_t) std::max(0, std::atoi(env)) << 20;
        auto w = gArenaFloor.find(device);
        if (w != gArenaFloor.end()) f = std::max(f, w->second);
        return f;
    }
    CtlArena *arena_of(int device, bool create)
    {
        std::lock_guard<std::mutex> g(gArenaMutex);
        auto it = gArenas.find(device);
        if (it != gArenas.end()) return it->second;
        if (!create) return nullptr;
        CtlArena *a = new CtlArena();
        gArenas[device] = a;
        return a;
    }
    // an engine comes (want > 0) or goes (want < 0) — the calling thread's current device is the arena's
    void arena_engine(int device, long long want)
    {
        CtlArena *a = arena_of(device, true);
        size_t floor;
        {
            std::lock_guard<std::mutex> g(gArenaMutex);
            floor = arena_floor(device);
        }
        std::lock_guard<std::mutex> g(a->mtx);
        if (want >= 0)
        {
            a->engines++;
            a->wanted += (size_t) want;
            const size_t target = std::max(a->wanted, floor);
            if (a->capacity() < target) (void) a->grow(target - a->capacity());
        }
        else
        {
            a->engines--;
            a->wanted -= std::min(a->wanted, (size_t) -want);
            a->reap(nullptr);
            a->shrink(a->engines > 0 ? std::max(a->wanted, floor) : 0);
        }
    }
}

namespace
{
    void arena_reap_all(int device)
    {
        if (CtlArena *a = arena_of(device, false))
        {
            std::vector<CtlArena::Pending> waitfor;
            {
                std::lock_guard<std::mutex> g(a->mtx);
                a->reap(&waitfor);
            }
            for (const CtlArena::Pending &pe : waitfor) (void) hipEventSynchronize(pe.ev);
            std::lock_guard<std::mutex> g(a->mtx);
            a->reap(nullptr);
        }
    }
}

// (C ABI, hcv_api.hip: hcv_ctl_reserve) what the control arena of `device` holds at least while the device has an engine; an arena that
// exists grows to it now (a control call: the driver maps memory under whatever stream is running)
bool ctl_arena_reserve(int device, size_t bytes)
{
    CtlArena *a = nullptr;
    {
        std::lock_guard<std::mutex> g(gArenaMutex);
        gArenaFloor[device] = bytes;
        auto it = gArenas.find(device);
        if (it != gArenas.end()) a = it->second;
    }
    if (a)
    {
        DeviceGuard dg(device);
        std::lock_guard<std::mutex> g(a->mtx);
        if (a->engines > 0 && a->capacity() < bytes) return a->grow(bytes - a->capacity());
    }
    return true;
}

size_t ctl_arena_size(int device)
{
    CtlArena *a = arena_of(device, false);
    if (!a) return 0;
    std::lock_guard<std::mutex> g(a->mtx);
    return a->capacity();
}

hipError_t Engine::ctl_alloc(void **p, size_t bytes)
{
    if (CtlArena *a = arena_of(mDevice, false))
    {
        std::vector<CtlArena::Pending> waitfor;
        void *q = nullptr;
        {
            std::lock_guard<std::mutex> g(a->mtx);
            a->reap(nullptr);
            q = a->take(bytes);
            if (!q) a->reap(&waitfor);
        }
        if (!q && !waitfor.empty())
        {
            // (the control thread may wait for the device — blocks freed a moment ago, not yet through — but not with the arena's mutex)
            for (const CtlArena::Pending &pe : waitfor) (void) hipEventSynchronize(pe.ev);
            (void) hipGetLastError();
            std::lock_guard<std::mutex> g(a->mtx);
            a->reap(nullptr);
            q = a->take(bytes);
        }
        if (q)
        {
            *p = q;
            return hipSuccess;
        }
    }
    static const bool arena_debug = std::getenv("HCV_VERBOSE") != nullptr;
    if (arena_debug) std::fprintf(stderr, "[hcv arena] %zu bytes not served by the arena of device %d: stream-ordered pool\n", bytes, mDevice);
    mArenaMisses.fetch_add(1, std::memory_order_relaxed);
    hipMemPool_t pool = ctl_pool(mDevice);
    const hipError_t e = pool ? hipMallocFromPoolAsync(p, bytes, pool, mCtlStream) : hipMallocAsync(p, bytes, mCtlStream);
    if (e != hipSuccess) (void) hipGetLastError();
    return e;
}

void Engine::ctl_free(void *p)
{
    if (!p) return;
    if (CtlArena *a = arena_of(mDevice, false))
    {
        bool ours;
        {
            std::lock_guard<std::mutex> g(a->mtx);
            ours = a->chunk_of(p) != nullptr;
        }
        if (ours)
        {
            hipEvent_t ev = nullptr;
            const bool have = hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess && hipEventRecord(ev, mCtlStream) == hipSuccess;
            if (!have)
            {
                // (no event to be had: wait the stream out, then the block is free at once)
                (void) hipGetLastError();
                (void) hipStreamSynchronize(mCtlStream);
                if (ev) (void) hipEventDestroy(ev);
                ev = nullptr;
            }
            std::lock_guard<std::mutex> g(a->mtx);
            if (ev)
                a->pending.push_back({ p, ev });
            else
                a->give_back(p);
            return;
        }
    }
    (void) hipFreeAsync(p, mCtlStream);
}

// staging room of a stage for one pair's spectra (grown with the stage's capacity); control thread only
bool Engine::ensure_staging(Stage &st, uint32_t parts)
{
    if (parts <= st.stage_parts) return true;
    HCV_TRY(hipStreamSynchronize(mCtlStream));
    HCV_TRY(hipEventSynchronize(mEvSwapDone));
    if (st.stage_spec) ctl_free(st.stage_spec);
    st.stage_spec = nullptr;
    st.stage_parts = 0;
    const uint32_t want = std::max<uint32_t>(parts, st.Pcap);
    HCV_TRY(ctl_alloc((void **) &st.stage_spec, sizeof(float2) * (size_t) want * st.M));
    st.stage_parts = want;
    return true;
}

// Load / clear one pair's IR (Convolver::set -> ... -> PartitionedConvolve::set, PartitionedConvolve.cpp:173-225).
//
// Phase A, no owner section: the IR is uploaded and transformed into STAGING buffers on the control stream — the pageable host
// copy, the FFTs of every partition and the wait for them happen while the audio thread keeps processing with the pair's
// previous spectra (the reference mutes the pair for those blocks instead, MonoConvolve.cpp:118-140,181-183).
// Phase B, in an owner section: the pair's pending output is retired, the staged spectra are copied into place on the main
// stream (device-to-device, asynchronous) and the bookkeeping is swapped — a short, host-only section with no allocation, no
// host copy and no device wait, which is all the audio thread can ever wait for.
bool Engine::set_ir(uint32_t in, uint32_t out, const float *ir, uint64_t len, bool device_ptr)
{
    if (out >= mCfg.nout || (!mCfg.diag && in >= mCfg.nin)) return false;
    RoctxRange range("hcv:set");
    std::lock_guard<std::mutex> gs(mSetMutex);
    DeviceGuard dg(mDevice);
    if (!ir) len = 0;
    const size_t pair = pair_index(in, out);

    // ---- phase A
    const float *dsrc = nullptr;
    // the previous swap's copies read the staging buffers on the main stream: order this call's writes behind them
    HCV_TRY(hipStreamWaitEvent(mCtlStream, mEvSwapDone, 0));
    if (len)
    {
        if (device_ptr)
            dsrc = ir;
        else
        {
            if (len > mIrCap)
            {
                HCV_TRY(hipStreamSynchronize(mCtlStream));
                if (mIrBuf) ctl_free(mIrBuf);
                mIrBuf = nullptr;
                mIrCap = 0;
                uint64_t want = std::max<uint64_t>(len + len / 2, 65536);       // (headroom: growing IRs do not reallocate every time)
                HCV_TRY(ctl_alloc((void **) &mIrBuf, sizeof(float) * want));
                mIrCap = want;
            }
            HCV_TRY(hipMemcpyAsync(mIrBuf, ir, sizeof(float) * len, hipMemcpyHostToDevice, mCtlStream));
            dsrc = mIrBuf;
        }
    }
    std::vector<uint32_t> newPs(mStages.size(), 0);
    for (size_t si = 0; si < mStages.size(); si++)
    {
        Stage &st = *mStages[si];
        uint64_t seg = len > st.cfg.offset ? len - st.cfg.offset : 0;          // PartitionedConvolve.cpp:192-193
        if (st.cfg.length && st.cfg.length < seg) seg = st.cfg.length;
        const uint64_t cap = (uint64_t) st.Pcap * st.M;
        if (seg > cap) seg = cap;                                               // :195-199 (caller reports the error)
        const uint32_t newP = (uint32_t) ((seg + st.M - 1) / st.M);
        newPs[si] = newP;
        if (!newP) continue;
        if (!ensure_staging(st, newP)) return false;
        HCV_TRY(launch_rfft_ir(st.log2n, dsrc + st.cfg.offset, (long long) seg, (int) newP, st.stage_spec, st.tw, &st.big_ctl, mCtlStream));
    }
    uint64_t taps = 0;
    if (mCfg.has_td)
    {
        const uint64_t lim = mCfg.td_length ? mCfg.td_length : 2044;            // TimeDomainConvolve.cpp:77
        taps = len > mCfg.td_offset ? std::min<uint64_t>(len - mCfg.td_offset, lim) : 0;
        if (taps > 2044) taps = 2044;
        if (taps) HCV_TRY(hipMemcpyAsync(mStageTaps, dsrc + mCfg.td_offset, sizeof(float) * taps, hipMemcpyDeviceToDevice, mCtlStream));
        HCV_TRY(hipMemsetAsync(mStageTaps + taps, 0, sizeof(float) * (2048 - taps), mCtlStream));
        if (mHeadFFT)
        {
            Stage &s0 = *mStages[0];
            HCV_TRY(launch_rfft_ir(s0.log2n, taps ? dsrc + mCfg.td_offset : mHist, (long long) taps, 1, mStageHead, s0.tw, &s0.big_ctl, mCtlStream));
        }
    }
    if (mLeadSlot)
    {
        Stage &tl = *mStages[mPivot];
        const uint64_t first = std::min<uint64_t>(len, tl.M);
        HCV_TRY(launch_rfft_ir(tl.log2n, first ? dsrc : mHist, (long long) first, 1, mStageTailHead, tl.tw, &tl.big_ctl, mCtlStream));
    }
    HCV_TRY(hipStreamSynchronize(mCtlStream));          // the device has consumed `ir`; the staging buffers are complete

    // ---- phase B: between two blocks — run by the audio thread itself while a stream is running (run_exclusive)
    return run_exclusive([&]() -> bool
    {
        if (!fence_background(exact_restart())) return false;
        // what the pair still has to deliver belongs to the spectra about to be replaced: take it out of the timelines now
        if (mLoaded[pair] && !mRetired[pair] && !retire_pair(pair)) return false;
        mRetired[pair] = 1;                                                         // (an empty pair has nothing pending)
        // (the copies of the staged spectra / taps into place and the zeroing of what the old IR had beyond the new one: ONE launch —
        // this section runs on the audio thread while a stream is running; segments the plan cannot take go the long way)
        bool any = false;
        SwapPlan plan;
        auto put = [&](void *dst, const void *src, size_t copy, size_t zero) -> bool
        {
            if (plan.add(dst, src, (long long) copy, (long long) zero)) return true;
            if (copy) HCV_TRY(hipMemcpyAsync(dst, src, copy, hipMemcpyDeviceToDevice, mStream));
            if (zero) HCV_TRY(hipMemsetAsync(static_cast<char *>(dst) + copy, 0, zero, mStream));
            return true;
        };
        for (size_t si = 0; si < mStages.size(); si++)
        {
            Stage &st = *mStages[si];
            const uint32_t newP = std::min(newPs[si], st.Pcap), oldP = st.pact[pair];
            float2 *dst = st.Ht() + pair * st.hstride();
            if (newP || oldP > newP)
                if (!put(dst, st.stage_spec, sizeof(float2) * (size_t) newP * st.M, oldP > newP ? sizeof(float2) * (size_t) (oldP - newP) * st.M : 0)) return false;
            st.live_parts += newP;
            st.live_parts -= oldP;
            st.pact[pair] = newP;
            st.P = *std::max_element(st.pact.begin(), st.pact.end());
            any = any || newP;
        }
        if (mCfg.has_td)
        {
            if (!put(mTaps + pair * 2048, mStageTaps, sizeof(float) * 2048, 0)) return false;
            if (mHeadFFT && !put(mHeadSpec + pair * (size_t) mStages[0]->M, mStageHead, sizeof(float2) * mStages[0]->M, 0)) return false;
            mTdCount[pair] = (uint32_t) taps;
            uint32_t mx = *std::max_element(mTdCount.begin(), mTdCount.end());
            mTdLpad = ((mx + 15) / 16) * 16;
            any = any || taps;
        }
        if (mLeadSlot && !put(mStages[mPivot]->Hs + pair * mStages[mPivot]->hstride(), mStageTailHead, sizeof(float2) * mStages[mPivot]->M, 0)) return false;
        HCV_TRY(launch_swap_in(plan, mStream));
        mLoaded[pair] = any ? 1 : 0;
        __atomic_store_n(&mPending[pair], (uint8_t) 1, __ATOMIC_RELEASE);           // set() always ends in reset()
        mCtlDirty = true;
        HCV_TRY(hipEventRecord(mEvSwapDone, mStream));
        return true;
    });
}

// A section that must run between two blocks, exclusive of the audio thread's enqueue (see CtlJob in hcv_engine.h).
//
// "A stream is running": a process call within the last kStreamingWindowNs.  The window is what keeps control threads away from the
// ownership while an audio thread exists at all — a paced real-time host calls every 0.7 - 2.7 ms, an offline loop back to back; only
// a stream that has been silent for this long (stopped, or not started yet) is taken for stopped.  A control call that arrives within
// the window of a stream that HAS just stopped waits it out (control threads may wait) before it serves itself.
constexpr long long kStreamingWindowNs = 400000000;

static inline long long steady_ns()
{
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Who the audio thread is: the calling thread — or, on a shard's enqueue thread (hcv_shard_pool.h), the user thread that posted the
// block, so that a single-threaded user (process, then set) is recognised on every shard's engine and not only on the first
static thread_local size_t tlsAudioIdentity = 0;
void set_thread_audio_identity(size_t id) { tlsAudioIdentity = id; }
size_t current_thread_identity() { return std::hash<std::thread::id>()(std::this_thread::get_id()) | 1; }
static inline size_t this_thread_hash() { return tlsAudioIdentity ? tlsAudioIdentity : current_thread_identity(); }

// test aid (tests/cpp/audio_contract.cpp, tests/test_audio_thread_contract_gpu.py): a control thread sleeps this long INSIDE its part of
// every posted section's hand-over — after posting, before it looks for the result — and inside every section it runs itself: a
// preempted control thread, on demand.  Nothing of it is on the audio thread's path.
static inline long long test_ctl_stall_us()
{
    static const long long us = std::getenv("HCV_TEST_CTL_STALL_US") ? std::atoll(std::getenv("HCV_TEST_CTL_STALL_US")) : 0;
    return us;
}

bool Engine::run_exclusive(std::function<bool()> fn, int slot)
{
    // a caller that IS the audio thread (one thread making both kinds of call: offline use, most tests) cannot be inside a
    // process call: it takes the ownership directly
    const bool same_thread = mAudioThread.load(std::memory_order_acquire) == this_thread_hash();
    for (;;)
    {
        const long long since = steady_ns() - mLastAudioNs.load(std::memory_order_seq_cst);
        if (!same_thread && since < kStreamingWindowNs)
        {
            // a stream is running: hand the section to the audio thread's next call and wait for it (this is the control thread).
            // (slot 0: the control calls, one at a time under mSetMutex; slot 1: everyone else — synchronize, statistics — one at a time
            // under mQueryMutex, taken HERE only: a caller that is the audio thread never comes this way)
            std::unique_lock<std::mutex> one_at_a_time;
            if (slot == 1) one_at_a_time = std::unique_lock<std::mutex>(mQueryMutex);
            CtlJob job;
            job.fn = fn;
            mMailbox[slot].store(&job, std::memory_order_release);
            if (const long long us = test_ctl_stall_us()) std::this_thread::sleep_for(std::chrono::microseconds(us));
            bool withdrawn = false;
            while (!job.done.load(std::memory_order_acquire))
            {
                if (steady_ns() - mLastAudioNs.load(std::memory_order_acquire) > kStreamingWindowNs)
                {
                    // no call came: the stream has stopped.  Take the section back — unless the audio thread picked it up just now
                    CtlJob *expect = &job;
                    if (mMailbox[slot].compare_exchange_strong(expect, nullptr, std::memory_order_acq_rel))
                    {
                        withdrawn = true;
                        break;
                    }
                }
                std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
            if (!withdrawn) return job.ok;
            continue;                                       // look at the clock again
        }
        // no stream: the control thread owns the engine for the section.  (Dekker with audio_enter: that side stamps mLastAudioNs, then
        // tries the ownership; this side takes the ownership, then reads the stamp — a call that started meanwhile is seen here, and the
        // ownership goes back before anything was touched.  Only a call that starts AFTER this check, inside the section, finds the
        // engine owned: the first call of a stream — start_collisions.)
        uint32_t expect = kOwnerFree;
        if (mOwner.compare_exchange_strong(expect, kOwnerControl, std::memory_order_seq_cst))
        {
            if (!same_thread && steady_ns() - mLastAudioNs.load(std::memory_order_seq_cst) < kStreamingWindowNs)
            {
                mOwner.store(kOwnerFree, std::memory_order_release);
                continue;                                   // a stream started meanwhile: post it
            }
            if (const long long us = test_ctl_stall_us())
                if (!same_thread) std::this_thread::sleep_for(std::chrono::microseconds(us));
            const bool ok = fn();
            mCtlSections.fetch_add(1, std::memory_order_relaxed);
            mOwner.store(kOwnerFree, std::memory_order_release);
            return ok;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(20));    // (another control thread's section, or a process call of the same thread's... never: a control thread waits)
    }
}

// audio thread: sections posted by control threads, between two blocks
void Engine::run_mailbox()
{
    for (int slot = 0; slot < 2; slot++)
    {
        if (!mMailbox[slot].load(std::memory_order_acquire)) continue;
        if (CtlJob *job = mMailbox[slot].exchange(nullptr, std::memory_order_acq_rel))
        {
            const long long t0 = steady_ns();
            job->ok = job->fn();
            const uint64_t took = (uint64_t) (steady_ns() - t0);
            mMailboxRuns.fetch_add(1, std::memory_order_relaxed);
            mMailboxNsTotal.fetch_add(took, std::memory_order_relaxed);
            uint64_t prev = mMailboxNsMax.load(std::memory_order_relaxed);
            while (took > prev && !mMailboxNsMax.compare_exchange_weak(prev, took, std::memory_order_relaxed)) {}
            job->done.store(true, std::memory_order_release);
        }
    }
}

// audio thread, first thing in a process call: ONE attempt at the ownership, no loop, no lock, no device call — for the calls of a
// real-time host.  `samples` = the length of the call: a call of kOfflineCallSamples or more is no audio callback (2048 samples are 21 ms at
// 96 kHz), and for such a call — an offline loop that pauses between its chunks while another thread loads impulse responses — a silent
// block of that size would be the wrong trade: it waits out the control thread's short section instead (sleeping, 100 ms at most; counted in
// start_waits).  Nothing changes for the calls the contract is about.
constexpr uint64_t kOfflineCallSamples = 2048;

bool Engine::audio_enter(uint64_t samples)
{
    const long long now = steady_ns();
    note_device_streaming(mDevice, now);
    mLastAudioNs.store(now, std::memory_order_seq_cst);
    mAudioThread.store(this_thread_hash(), std::memory_order_release);
    uint32_t expect = kOwnerFree;
    if (!mOwner.compare_exchange_strong(expect, kOwnerAudio, std::memory_order_seq_cst))
    {
        bool got = false;
        if (samples >= kOfflineCallSamples)
        {
            for (int k = 0; k < 5000 && !got; k++)
            {
                std::this_thread::sleep_for(std::chrono::microseconds(20));
                expect = kOwnerFree;
                got = mOwner.compare_exchange_strong(expect, kOwnerAudio, std::memory_order_seq_cst);
            }
            if (got) mStartWaits.fetch_add(1, std::memory_order_relaxed);
        }
        if (!got)
        {
            mStartCollisions.fetch_add(1, std::memory_order_relaxed);
            return false;
        }
    }
    run_mailbox();
    return true;
}

// audio thread, end of a call's enqueue: a section posted while the call was being enqueued runs now (between this block and the
// next, as at the start of a call), then the ownership goes back
void Engine::audio_leave()
{
    run_mailbox();
    mLastAudioNs.store(steady_ns(), std::memory_order_seq_cst);
    mOwner.store(kOwnerFree, std::memory_order_release);
}

// (flag writes, no lock: consumed — exchanged — by the next block's apply_pending_resets)
void Engine::reset_pair(uint32_t in, uint32_t out)
{
    if (out >= mCfg.nout || (!mCfg.diag && in >= mCfg.nin)) return;
    __atomic_store_n(&mPending[pair_index(in, out)], (uint8_t) 1, __ATOMIC_RELEASE);
}

void Engine::reset_all()
{
    // (ONE atomic generation count, not a flag per pair raised one after the other: the block that sees it restarts EVERY pair at the
    // same sample — a block racing the loop used to restart some pairs now and the rest one block later)
    mResetAllGen.fetch_add(1, std::memory_order_release);
}

// Every loaded pair restarts: clear the rings and restart the hop clock.  The input-spectrum rings are not
// cleared — like the reference (PartitionedConvolve.cpp:271 clears only the frame/accum buffers) stale slots are
// fenced off by the valid-partition bound instead (mValidPartitions there, h - hv here).
bool Engine::global_reset()
{
    mN = 0;
    mTailHeadPrev = false;
    mCtlDirty = true;
    drop_ghosts();
    HCV_TRY(hipMemsetAsync(mHist, 0, sizeof(float) * mCfg.nin * mHistLen, mStream));
    const size_t pairs = (size_t) mCfg.nout * mNinAlloc;
    for (Stage *st : mStages)
    {
        HCV_TRY(hipMemsetAsync(st->timeline, 0, sizeof(float) * mCfg.nout * st->tl_len, mStream));
        HCV_TRY(hipMemsetAsync(st->hv, 0, sizeof(long long) * pairs, mStream));
        st->max_hv = 0;
        st->hv_zero = true;
    }
    if (mTdValid) HCV_TRY(hipMemsetAsync(mTdValid, 0, sizeof(long long) * pairs, mStream));
    mTdMaxValid = 0;
    return true;
}

bool Engine::synchronize()
{
    DeviceGuard dg(mDevice);
    {
        // (boundary chains the last small block left running past its emit: the main stream goes behind them first — a section, run
        // between two blocks by whoever owns the engine then)
        if (!run_exclusive([this]() -> bool { return fence_chains(); }, 1)) return false;
    }
    HCV_TRY(hipStreamSynchronize(mStream));
    if (mProfiling) collect_events();
    return true;
}

void Engine::set_profiling(bool on)
{
    (void) run_exclusive([this, on]() -> bool { mProfiling = on; return true; }, 1);
}

void Engine::collect_events()
{
    (void) run_exclusive([this]() -> bool { collect_events_owned(); return true; }, 1);
}

// (owner of the engine's state)
void Engine::collect_events_owned()
{
    for (EventPair *ev : mEvents)
    {
        if (!ev->live) continue;
        if (hipEventQuery(ev->b) == hipErrorNotReady)       // background accumulation still running: collect it next time
        {
            (void) hipGetLastError();
            continue;
        }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev->a, ev->b) == hipSuccess && ev->stage < mStages.size()) mStages[ev->stage]->ms += ms;
        else (void) hipGetLastError();
        ev->live = false;
    }
}

bool Engine::stage_stats(size_t s, StageStats *out)
{
    if (!out) return false;
    return run_exclusive([this, s, out]() -> bool
    {
        if (s >= mStages.size()) return false;
        const Stage &st = *mStages[s];
        out->fft_size = st.N;
        out->partitions = st.P;
        out->nin = mCfg.diag ? 1 : mCfg.nin;
        out->nout = mCfg.nout;
        out->mac_launches = st.launches;
        out->mac_hops = st.hops;
        out->mac_ms = st.ms;
        out->ksplit = st.last_ksplit;
        out->out_tile = st.last_ot;
        out->mac_steady_launches = st.steady_launches;
        out->hop_tile = st.last_tt;
        out->launch_partitions = st.last_parts;
        out->fused_launches = st.fused_launches;
        out->fused_stood_down = st.nxm_stood_down;
        out->host_pre_launches = st.host_pre_launches;
        return true;
    }, 1);
}

void Engine::clear_stats()
{
    (void) run_exclusive([this]() -> bool
    {
        for (Stage *st : mStages)
        {
            st->launches = st->hops = st->steady_launches = st->fused_launches = st->host_pre_launches = 0;
            st->ms = 0.0;
        }
        return true;
    }, 1);
}

} // namespace hcv
