// Device-resident N x M multi-stage convolution engine: host-side orchestration of the gfx950 kernels.
// See hcv_engine.h for the HBM layout.

#include "hcv_engine.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>

namespace hcv
{

#define HCV_TRY(expr)                                                                                                  \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) return fail(#expr, e_);                                                                  \
    } while (0)

static long long pow2ceil(long long v)
{
    long long p = 1;
    while (p < v) p <<= 1;
    return p;
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
// 30 % without the overlap, so multi-stage engines keep their streams).  HCV_ONE_STREAM = 0 / 1 forces the choice.
static bool one_stream_mode(const EngineCfg &cfg)
{
    if (const char *env = std::getenv("HCV_ONE_STREAM")) return std::atoi(env) != 0;
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

constexpr int kBgSlices = 16;
constexpr int kTailHeadSplit = 8;

// HCV_EXACT_RESTART=0 falls back to the hop-granular fence alone (the restarted pair may see up to two hops of older input
// per stage and its pending output is not withdrawn) — for A/B comparison only
static bool exact_restart()
{
    static const bool on = !(std::getenv("HCV_EXACT_RESTART") && std::atoi(std::getenv("HCV_EXACT_RESTART")) == 0);
    return on;
}


struct Engine::Stage
{
    StageCfg cfg;
    int log2n = 0;
    uint32_t N = 0, M = 0;
    uint32_t Pcap = 0, P = 0, R = 0, Tmax = 0;
    float2 *Hs = nullptr, *X = nullptr;
    float2 *Y = nullptr;                // scratch of the current block = Yq[block parity]
    float2 *Yq[2] = { nullptr, nullptr };   // split-K partials, double-buffered so MAC(k+1) can run while block k is inverted
    size_t y_elems = 0;
    // Deferred ("time-spread") mode, the GPU form of the reference's partition scheduler (PartitionedConvolve.cpp:321-348):
    // partitions 1..P-1 of hop h+1 only need spectra up to hop h, so they are accumulated BETWEEN the boundaries of hop h
    // and hop h+1, in up to kBgSlices short launches spread over the calls of that hop in step with the samples that
    // have arrived (a single long launch would sit in a hardware queue that other streams share and stall them for
    // milliseconds); the boundary of hop h+1 then only pays partition 0 + the inverse FFT.
    float2 *Ypre = nullptr;             // [kBgSlices][nout][M]: one partial sum per slice; slot 0 receives their total
    long long pre_hop = -1;             // hop index the slices accumulate for (-1 = no plan)
    int bg_parts = 0;                   // partitions 1..bg_parts of that hop are to be accumulated
    int bg_slices = 0, bg_launched = 0; // planned / already launched slices
    hipEvent_t bg_done = nullptr;       // recorded after every background launch
    bool bg_pending = false;
    float *timeline = nullptr;          // [nout][tl_len] this stage's hop results at their emission times
    long long tl_len = 0;
    BigFFTWork big;                     // scratch of the four-step FFT (only for N > 32768)
    hipStream_t stream = nullptr;       // stages are independent until emit(): each runs on its own stream (the MAC stream)
    hipEvent_t mac_done[2] = { nullptr, nullptr };     // the stage's spectral_mac of a block has finished (tail gate)
    hipEvent_t done[2] = { nullptr, nullptr };   // by block parity
    long long *hv = nullptr;
    long long max_hv = 0;
    // exact per-pair restart: device table of the live ghost entries of this stage, grouped by output
    int *gh_start = nullptr;            // [nout + 1]
    GhostEntry *gh_ent = nullptr;       // [pairs]
    std::vector<GhostEntry> gh_host;    // mirror, in table order
    std::vector<size_t> gh_pair;        // pair of each entry
    int gh_count = 0;
    long long gh_min_hr = 0, gh_max_hr = 0;
    std::vector<uint32_t> pact;
    uint64_t live_parts = 0;            // sum of pact
    const float2 *tw = nullptr;
    // stats
    uint64_t launches = 0, hops = 0;
    double ms = 0.0;
    uint32_t last_ksplit = 0, last_ot = 0;
};

struct Engine::GhostEvent
{
    long long t0 = 0;                   // sample the restart took effect at
    int refs = 0;                       // pairs still pointing at this event
    std::vector<int> slot;              // input row -> row of the spectra blocks (-1: not part of the restart)
    std::vector<float2 *> spec;         // per stage: [rows][2][M], frame h at slot h & 1
    std::vector<size_t> bytes;
};

struct Engine::EventPair
{
    hipEvent_t a = nullptr, b = nullptr;
    size_t stage = 0;
    bool live = false;
};

bool Engine::fail(const char *what, hipError_t e)
{
    mErr = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}

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
    HCV_TRY(hipSetDevice(mDevice));

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

    HCV_TRY(hipStreamCreateWithFlags(&mStream, hipStreamNonBlocking));
    mOneStream = one_stream_mode(mCfg);
    if (mOneStream)
        mTdStream = mInStream = mStream;
    else
    {
        HCV_TRY(hipStreamCreateWithFlags(&mTdStream, hipStreamNonBlocking));
        HCV_TRY(hipStreamCreateWithFlags(&mInStream, hipStreamNonBlocking));
    }
    for (int k = 0; k < 2; k++)
    {
        HCV_TRY(hipEventCreateWithFlags(&mEvInput[k], hipEventDisableTiming));
        HCV_TRY(hipEventCreateWithFlags(&mEvTd[k], hipEventDisableTiming));
        HCV_TRY(hipEventCreateWithFlags(&mEvEmit[k], hipEventDisableTiming));
    }
    HCV_TRY(hipEventCreateWithFlags(&mEvCtl, hipEventDisableTiming));
    HCV_TRY(hipEventCreateWithFlags(&mEvSerial, hipEventDisableTiming));

    // three blocks deep: block k+1 is scattered while block k-1's readers may still be running
    mHistLen = pow2ceil(3LL * mMaxBlock + std::max<long long>(nmax, 4096));
    const size_t pairs = (size_t) mCfg.nout * mNinAlloc;

    HCV_TRY(hipMalloc(&mHist, sizeof(float) * mCfg.nin * mHistLen));
    HCV_TRY(hipMemset(mHist, 0, sizeof(float) * mCfg.nin * mHistLen));
    HCV_TRY(hipMalloc(&mDevIn, sizeof(float) * mCfg.nin * mMaxBlock));
    HCV_TRY(hipMalloc(&mDevOut, sizeof(float) * mCfg.nout * mMaxBlock));
    HCV_TRY(hipHostMalloc(&mPinIn, sizeof(float) * mCfg.nin * mMaxBlock, hipHostMallocDefault));
    HCV_TRY(hipHostMalloc(&mPinOut, sizeof(float) * mCfg.nout * mMaxBlock, hipHostMallocDefault));
    if (mCfg.has_td)
    {
        for (int k = 0; k < 2; k++) HCV_TRY(hipMalloc(&mTdOut[k], sizeof(float) * mCfg.nout * mMaxBlock));
        HCV_TRY(hipMalloc(&mTaps, sizeof(float) * pairs * 2048));
        HCV_TRY(hipMemset(mTaps, 0, sizeof(float) * pairs * 2048));
        HCV_TRY(hipMalloc(&mTdValid, sizeof(long long) * pairs));
        HCV_TRY(hipMemset(mTdValid, 0, sizeof(long long) * pairs));
        mTdCount.assign(pairs, 0);
    }
    mPending.assign(pairs, 0);
    mLoaded.assign(pairs, 0);
    mRetired.assign(pairs, 0);
    mGhostOf.assign(pairs, nullptr);

    for (const StageCfg &sc : mCfg.stages)
    {
        Stage *st = new Stage();
        st->cfg = sc;
        st->log2n = ilog2(sc.fft_size);
        st->N = sc.fft_size;
        st->M = sc.fft_size / 2;
        st->pact.assign(pairs, 0);
        mStages.push_back(st);
        st->tw = twiddles(mDevice, st->log2n, &mErr);
        if (!st->tw) return false;
        if (!alloc_stage(*st)) return false;
    }
    // Head through the FFT: the head's taps (<= one hop of the first FFT stage, MonoConvolve.cpp:235-240) form one extra,
    // zero-latency partition of that stage, whose input spectra exist anyway.  Used for hop-aligned blocks of larger
    // matrices, where the direct-form FIR would cost more than all FFT-stage MACs together; ragged blocks and small
    // engines keep the direct form (fir_head_kernel).
    if (mCfg.has_td && !mStages.empty())
    {
        const Stage &s0 = *mStages[0];
        const uint64_t lim = mCfg.td_length ? mCfg.td_length : 2044;
        static const bool allow = !(std::getenv("HCV_HEAD_FFT") && std::atoi(std::getenv("HCV_HEAD_FFT")) == 0);
        if (allow && lim <= s0.M && !is_big_fft(s0.log2n) && (size_t) mCfg.nout * mNinAlloc >= 16)
        {
            mHeadFFT = true;
            HCV_TRY(hipMalloc(&mHeadSpec, sizeof(float2) * pairs * s0.M));
            HCV_TRY(hipMemset(mHeadSpec, 0, sizeof(float2) * pairs * s0.M));
            for (int k = 0; k < 2; k++) HCV_TRY(hipMalloc(&mHeadYq[k], sizeof(float2) * (size_t) s0.Tmax * mCfg.nout * s0.M));
        }
    }
    // Whole-hop mode (see enqueue_chunk): needs the zero-latency ladder — head at [0, a), every shorter stage continuing
    // where the previous coverage ends, the last stage starting exactly one of its hops in
    if (mStages.size() >= 2 || (mCfg.has_td && !mStages.empty()))
    {
        static const bool allow = !(std::getenv("HCV_TAIL_HEAD") && std::atoi(std::getenv("HCV_TAIL_HEAD")) == 0);
        uint64_t end = 0;
        bool ok = allow;
        if (mCfg.has_td)
        {
            ok = ok && mCfg.td_offset == 0 && mCfg.td_length > 0;
            end = mCfg.td_length;
        }
        for (size_t k = 0; ok && k + 1 < mStages.size(); k++)
        {
            const StageCfg &sc = mStages[k]->cfg;
            ok = sc.offset == end && sc.length > 0;
            end = sc.offset + sc.length;
        }
        const Stage &tl = *mStages.back();
        ok = ok && tl.cfg.offset == end && tl.cfg.offset == tl.M && !is_big_fft(tl.log2n);
        if (ok)
        {
            mTailHead = true;
            HCV_TRY(hipMalloc(&mTailHeadSpec, sizeof(float2) * pairs * tl.M));
            HCV_TRY(hipMemset(mTailHeadSpec, 0, sizeof(float2) * pairs * tl.M));
            // (room for kTailHeadSplit k-slices: with one partition the MAC has only the input axis to split)
            for (int k = 0; k < 2; k++) HCV_TRY(hipMalloc(&mTailHeadYq[k], sizeof(float2) * (size_t) tl.Tmax * kTailHeadSplit * mCfg.nout * tl.M));
        }
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
    st.R = st.Pcap + 2 * st.Tmax;      // FFT of block k+1 may write while the MAC of block k still reads
    const size_t hs_elems = pairs * st.Pcap * st.M;
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
    HCV_TRY(hipMalloc(&st.Ypre, sizeof(float2) * (size_t) kBgSlices * mCfg.nout * st.M));
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
    }
    st.tl_len = pow2ceil(2LL * mMaxBlock + st.M);       // two blocks deep (block k+1 adds while block k is emitted)
    HCV_TRY(hipMalloc(&st.timeline, sizeof(float) * mCfg.nout * st.tl_len));
    HCV_TRY(hipMemset(st.timeline, 0, sizeof(float) * mCfg.nout * st.tl_len));
    if (mOneStream)
        st.stream = mStream;
    else
    {
        HCV_TRY(hipStreamCreateWithFlags(&st.stream, hipStreamNonBlocking));
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
    if (st.Hs) (void) hipFree(st.Hs);
    if (st.X) (void) hipFree(st.X);
    for (int k = 0; k < 2; k++)
    {
        if (st.Yq[k]) (void) hipFree(st.Yq[k]);
        if (st.mac_done[k]) (void) hipEventDestroy(st.mac_done[k]);
        st.Yq[k] = nullptr;
        st.mac_done[k] = nullptr;
    }
    if (st.hv) (void) hipFree(st.hv);
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
    st.big.a = st.big.b = nullptr;
    for (int k = 0; k < 2; k++)
        if (st.done[k]) (void) hipEventDestroy(st.done[k]);
    if (st.stream && st.stream != mStream) (void) hipStreamDestroy(st.stream);
    st.Hs = st.X = st.Y = nullptr;
    st.stream = nullptr;
    st.hv = nullptr;
    st.timeline = nullptr;
    st.done[0] = st.done[1] = nullptr;
    st.stream = nullptr;
}

Engine::~Engine()
{
    (void) hipSetDevice(mDevice);
    if (mInStream) (void) hipStreamSynchronize(mInStream);
    if (mTdStream) (void) hipStreamSynchronize(mTdStream);
    for (Stage *st : mStages)
    {
        if (st->stream) (void) hipStreamSynchronize(st->stream);
    }
    if (mStream) (void) hipStreamSynchronize(mStream);
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
    if (mIrBuf) (void) hipFree(mIrBuf);
    if (mTaps) (void) hipFree(mTaps);
    if (mHeadSpec) (void) hipFree(mHeadSpec);
    if (mTailHeadSpec) (void) hipFree(mTailHeadSpec);
    for (int k = 0; k < 2; k++)
        if (mTailHeadYq[k]) (void) hipFree(mTailHeadYq[k]);
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
    if (mGhostHist) (void) hipFree(mGhostHist);
    if (mRetireTmp) (void) hipFree(mRetireTmp);
    if (mGhostPin) (void) hipHostFree(mGhostPin);
    if (mGhostUploaded) (void) hipEventDestroy(mGhostUploaded);
    if (mInStream && mInStream != mStream) (void) hipStreamDestroy(mInStream);
    if (mTdStream && mTdStream != mStream) (void) hipStreamDestroy(mTdStream);
    if (mStream) (void) hipStreamDestroy(mStream);
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
    std::lock_guard<std::mutex> g(mMutex);
    if (s < mStages.size())
    {
        mStages[s]->cfg.offset = offset;
        mStages[s]->cfg.length = length;
    }
}

void Engine::set_td_window(uint64_t offset, uint64_t length)
{
    std::lock_guard<std::mutex> g(mMutex);
    mCfg.td_offset = offset;
    mCfg.td_length = length;
}

// Control work (IR loads, resets, regrow) changes what a background accumulation reads or means: order it after any
// background MAC still in flight and drop the pre-accumulated spectra — unless the caller keeps the plan and corrects them
// itself (a restart of single pairs: retire_pair takes the pair out of the slices already accumulated).  Caller holds mMutex.
bool Engine::fence_background(bool keep_plan)
{
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
bool Engine::ensure_stage_capacity(size_t s, uint64_t capacity)
{
    std::lock_guard<std::mutex> g(mMutex);
    if (s >= mStages.size()) return false;
    (void) hipSetDevice(mDevice);
    Stage &st = *mStages[s];
    const uint32_t newP = (uint32_t) std::max<uint64_t>(1, (capacity + st.M - 1) / st.M);
    if (newP <= st.Pcap) return true;
    if (!fence_background()) return false;

    const size_t pairs = (size_t) mCfg.nout * mNinAlloc;
    const uint32_t newR = newP + 2 * st.Tmax;
    float2 *nHs = nullptr, *nX = nullptr;
    const size_t hs_bytes = sizeof(float2) * pairs * newP * st.M;
    const size_t x_bytes = sizeof(float2) * (size_t) mCfg.nin * newR * st.M;
    if (hipMalloc(&nHs, hs_bytes) != hipSuccess)
    {
        (void) hipGetLastError();
        return false;
    }
    if (hipMalloc(&nX, x_bytes) != hipSuccess)
    {
        (void) hipGetLastError();
        (void) hipFree(nHs);
        return false;
    }
    bool ok = hipMemsetAsync(nHs, 0, hs_bytes, mStream) == hipSuccess && hipMemsetAsync(nX, 0, x_bytes, mStream) == hipSuccess;
    if (ok && st.P > 0)
    {
        ok = launch_regrow_spectra(st.Hs, nHs, (long long) pairs, (int) st.Pcap, (int) newP, (int) st.M, mStream) == hipSuccess;
        const long long h_done = mN / st.M;                // hops completed so far: indices 0 .. h_done-1
        const int live = (int) std::min<long long>(st.P, h_done);
        if (ok && live > 0)
            ok = launch_regrow_ring(st.X, nX, (int) mCfg.nin, (int) st.R, (int) newR, (int) st.M, h_done - 1, live, mStream) == hipSuccess;
    }
    ok = ok && hipStreamSynchronize(mStream) == hipSuccess;
    if (!ok)
    {
        (void) hipFree(nHs);
        (void) hipFree(nX);
        mErr = "stage regrow failed";
        return false;
    }
    mCtlDirty = true;
    (void) hipFree(st.Hs);
    (void) hipFree(st.X);
    st.Hs = nHs;
    st.X = nX;
    st.Pcap = newP;
    st.R = newR;
    return true;
}

bool Engine::set_ir(uint32_t in, uint32_t out, const float *ir, uint64_t len, bool device_ptr)
{
    if (out >= mCfg.nout || (!mCfg.diag && in >= mCfg.nin)) return false;
    std::lock_guard<std::mutex> gs(mSetMutex);
    HCV_TRY(hipSetDevice(mDevice));
    if (!ir) len = 0;

    const float *dsrc = nullptr;
    if (len)
    {
        if (device_ptr)
            dsrc = ir;
        else
        {
            if (len > mIrCap)
            {
                HCV_TRY(hipStreamSynchronize(mStream));
                if (mIrBuf) (void) hipFree(mIrBuf);
                mIrBuf = nullptr;
                mIrCap = 0;
                uint64_t want = std::max<uint64_t>(len, 65536);
                HCV_TRY(hipMalloc(&mIrBuf, sizeof(float) * want));
                mIrCap = want;
            }
            dsrc = mIrBuf;
        }
    }

    {
        std::lock_guard<std::mutex> g(mMutex);
        if (!fence_background(exact_restart())) return false;
        if (len && !device_ptr) HCV_TRY(hipMemcpyAsync(mIrBuf, ir, sizeof(float) * len, hipMemcpyHostToDevice, mStream));
        const size_t pair = pair_index(in, out);
        // what the pair still has to deliver belongs to the spectra about to be replaced: take it out of the timelines now
        if (mLoaded[pair] && !mRetired[pair] && !retire_pair(pair)) return false;
        mRetired[pair] = 1;                                                         // (an empty pair has nothing pending)
        bool any = false;
        for (Stage *sp : mStages)
        {
            Stage &st = *sp;
            uint64_t seg = len > st.cfg.offset ? len - st.cfg.offset : 0;          // PartitionedConvolve.cpp:192-193
            if (st.cfg.length && st.cfg.length < seg) seg = st.cfg.length;
            const uint64_t cap = (uint64_t) st.Pcap * st.M;
            if (seg > cap) seg = cap;                                               // :195-199 (caller reports the error)
            const uint32_t newP = (uint32_t) ((seg + st.M - 1) / st.M);
            const uint32_t oldP = st.pact[pair];
            const uint32_t wr = std::max(newP, oldP);
            if (wr)
            {
                const float *src = seg ? dsrc + st.cfg.offset : mHist;              // never dereferenced when seg == 0
                HCV_TRY(launch_rfft_ir(st.log2n, src, (long long) seg, (int) wr, st.Hs + pair * (size_t) st.Pcap * st.M, st.tw, &st.big, mStream));
            }
            st.live_parts += newP;
            st.live_parts -= st.pact[pair];
            st.pact[pair] = newP;
            st.P = *std::max_element(st.pact.begin(), st.pact.end());
            any = any || newP;
        }
        if (mCfg.has_td)
        {
            const uint64_t lim = mCfg.td_length ? mCfg.td_length : 2044;            // TimeDomainConvolve.cpp:77
            uint64_t taps = len > mCfg.td_offset ? std::min<uint64_t>(len - mCfg.td_offset, lim) : 0;
            if (taps > 2044) taps = 2044;
            float *dst = mTaps + pair * 2048;
            if (taps) HCV_TRY(hipMemcpyAsync(dst, dsrc + mCfg.td_offset, sizeof(float) * taps, hipMemcpyDeviceToDevice, mStream));
            HCV_TRY(hipMemsetAsync(dst + taps, 0, sizeof(float) * (2048 - taps), mStream));
            if (mHeadFFT)
            {
                Stage &s0 = *mStages[0];
                const float *src = taps ? dsrc + mCfg.td_offset : mHist;
                HCV_TRY(launch_rfft_ir(s0.log2n, src, (long long) taps, 1, mHeadSpec + pair * (size_t) s0.M, s0.tw, &s0.big, mStream));
            }
            mTdCount[pair] = (uint32_t) taps;
            uint32_t mx = *std::max_element(mTdCount.begin(), mTdCount.end());
            mTdLpad = ((mx + 15) / 16) * 16;
            any = any || taps;
        }
        if (mTailHead)
        {
            Stage &tl = *mStages.back();
            const uint64_t first = std::min<uint64_t>(len, tl.M);
            HCV_TRY(launch_rfft_ir(tl.log2n, first ? dsrc : mHist, (long long) first, 1, mTailHeadSpec + pair * (size_t) tl.M, tl.tw, &tl.big, mStream));
        }
        mLoaded[pair] = any ? 1 : 0;
        mPending[pair] = 1;                                                         // set() always ends in reset()
        mCtlDirty = true;
    }
    HCV_TRY(hipStreamSynchronize(mStream));
    return true;
}

void Engine::reset_pair(uint32_t in, uint32_t out)
{
    if (out >= mCfg.nout || (!mCfg.diag && in >= mCfg.nin)) return;
    std::lock_guard<std::mutex> g(mMutex);
    mPending[pair_index(in, out)] = 1;
}

void Engine::reset_all()
{
    std::lock_guard<std::mutex> g(mMutex);
    std::fill(mPending.begin(), mPending.end(), 1);
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
    }
    if (mTdValid) HCV_TRY(hipMemsetAsync(mTdValid, 0, sizeof(long long) * pairs, mStream));
    mTdMaxValid = 0;
    return true;
}

// ------------------------------------------------------------------------------------------------ exact per-pair restart
// (see hcv_ghost.hip for the scheme).  All of it is control work on mStream, which every block's emit has ordered after
// the stages' work; callers hold mMutex and have fenced the background accumulation.

void *Engine::ghost_alloc(size_t bytes)
{
    for (size_t k = 0; k < mGhostPool.size(); k++)
        if (mGhostPool[k].first == bytes)
        {
            void *p = mGhostPool[k].second;
            mGhostPool.erase(mGhostPool.begin() + (long) k);
            return p;
        }
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess)
    {
        (void) hipGetLastError();
        return nullptr;
    }
    return p;
}

void Engine::ghost_free(void *p, size_t bytes)
{
    if (p) mGhostPool.emplace_back(bytes, p);
}

void Engine::release_ghost(size_t pair)
{
    GhostEvent *ev = mGhostOf[pair];
    if (!ev) return;
    mGhostOf[pair] = nullptr;
    if (--ev->refs > 0) return;
    for (size_t s = 0; s < ev->spec.size(); s++) ghost_free(ev->spec[s], ev->bytes[s]);
    mGhostEvents.erase(std::find(mGhostEvents.begin(), mGhostEvents.end(), ev));
    delete ev;
}

void Engine::drop_ghosts()
{
    for (size_t p = 0; p < mGhostOf.size(); p++) release_ghost(p);
    for (Stage *st : mStages) st->gh_count = 0;
    mGhostPruneAt = -1;
}

// the restart of `ev` can still reach a launch of stage `st` while the current hop is at most P + 1 past the restart's
static inline bool ghost_live(long long t0, long long now, uint32_t M, uint32_t Pcap)
{
    return now / M - t0 / M <= (long long) Pcap + 2;
}

bool Engine::prune_ghosts()
{
    bool changed = false;
    for (size_t p = 0; p < mGhostOf.size(); p++)
    {
        GhostEvent *ev = mGhostOf[p];
        if (!ev) continue;
        bool live = false;
        for (Stage *st : mStages) live = live || ghost_live(ev->t0, mN, st->M, st->Pcap);
        if (!live)
        {
            release_ghost(p);
            changed = true;
        }
    }
    return changed ? rebuild_ghost_tables() : true;
}

bool Engine::rebuild_ghost_tables()
{
    const size_t pairs = (size_t) mCfg.nout * mNinAlloc;
    const size_t start_bytes = ((sizeof(int) * (mCfg.nout + 1) + 15) / 16) * 16;
    const size_t per_stage = start_bytes + sizeof(GhostEntry) * pairs;
    if (!mGhostPin)
    {
        mGhostPinBytes = per_stage * mStages.size();
        HCV_TRY(hipHostMalloc(&mGhostPin, mGhostPinBytes, hipHostMallocDefault));
        HCV_TRY(hipEventCreateWithFlags(&mGhostUploaded, hipEventDisableTiming));
    }
    else
        HCV_TRY(hipEventSynchronize(mGhostUploaded));       // the previous upload has left the staging buffer
    mGhostPruneAt = -1;
    for (size_t si = 0; si < mStages.size(); si++)
    {
        Stage &st = *mStages[si];
        if (!st.gh_start)
        {
            HCV_TRY(hipMalloc(&st.gh_start, start_bytes));
            HCV_TRY(hipMalloc(&st.gh_ent, sizeof(GhostEntry) * pairs));
        }
        int *start = reinterpret_cast<int *>(mGhostPin + si * per_stage);
        GhostEntry *ent = reinterpret_cast<GhostEntry *>(mGhostPin + si * per_stage + start_bytes);
        st.gh_host.clear();
        st.gh_pair.clear();
        st.gh_min_hr = st.gh_max_hr = 0;
        int n = 0;
        for (uint32_t o = 0; o < mCfg.nout; o++)
        {
            start[o] = n;
            for (uint32_t c = 0; c < mNinAlloc; c++)
            {
                const size_t pair = (size_t) o * mNinAlloc + c;
                const GhostEvent *ev = mGhostOf[pair];
                if (!ev || !ghost_live(ev->t0, mN, st.M, st.Pcap)) continue;
                const int row = ev->slot[mCfg.diag ? o : c];
                if (row < 0) continue;
                const long long h_r = ev->t0 / st.M;
                const float2 *blk = ev->spec[si] + (size_t) row * 2 * st.M;
                GhostEntry e;
                e.h_r = h_r;
                e.g0 = reinterpret_cast<const float4 *>(blk + (size_t) (h_r & 1) * st.M);
                e.g1 = reinterpret_cast<const float4 *>(blk + (size_t) ((h_r + 1) & 1) * st.M);
                e.i = (int) c;
                e.pad = 0;
                ent[n] = e;
                st.gh_host.push_back(e);
                st.gh_pair.push_back(pair);
                st.gh_min_hr = n ? std::min(st.gh_min_hr, h_r) : h_r;
                st.gh_max_hr = n ? std::max(st.gh_max_hr, h_r) : h_r;
                n++;
            }
        }
        start[mCfg.nout] = n;
        st.gh_count = n;
        HCV_TRY(hipMemcpyAsync(st.gh_start, start, start_bytes, hipMemcpyHostToDevice, mStream));
        if (n) HCV_TRY(hipMemcpyAsync(st.gh_ent, ent, sizeof(GhostEntry) * n, hipMemcpyHostToDevice, mStream));
    }
    for (const GhostEvent *ev : mGhostEvents)
    {
        long long until = 0;
        for (Stage *st : mStages) until = std::max(until, (ev->t0 / st->M + (long long) st->Pcap + 3) * (long long) st->M);
        mGhostPruneAt = mGhostPruneAt < 0 ? until : std::min(mGhostPruneAt, until);
    }
    HCV_TRY(hipEventRecord(mGhostUploaded, mStream));
    mCtlDirty = true;
    return true;
}

// Ghost spectra for the pairs restarting at mN: the pre-restart part of the two frames that straddle mN, per stage, for every
// input one of the pairs reads.
bool Engine::make_ghost_event(const std::vector<size_t> &pairs)
{
    for (size_t pair : pairs) release_ghost(pair);
    if (mN <= 0 || pairs.empty() || mStages.empty() || !exact_restart()) return true;
    GhostEvent *ev = new GhostEvent();
    ev->t0 = mN;
    ev->slot.assign(mCfg.nin, -1);
    std::vector<int> rows;
    for (size_t pair : pairs)
    {
        const uint32_t o = (uint32_t) (pair / mNinAlloc), c = (uint32_t) (pair % mNinAlloc);
        const uint32_t row = mCfg.diag ? o : c;
        if (ev->slot[row] < 0)
        {
            ev->slot[row] = (int) rows.size();
            rows.push_back((int) row);
        }
    }
    uint32_t nmax = 0;
    for (Stage *st : mStages) nmax = std::max(nmax, st->N);
    if (!mGhostHist)
    {
        mGhostLen = pow2ceil(2LL * nmax);
        HCV_TRY(hipMalloc(&mGhostHist, sizeof(float) * mCfg.nin * mGhostLen));
    }
    HCV_TRY(launch_ghost_hist(mHist, mHistLen, mHistLen - 1, rows.data(), (int) rows.size(), mGhostHist, mGhostLen, mN, mStream));
    for (Stage *st : mStages)
    {
        const size_t bytes = sizeof(float2) * rows.size() * 2 * st->M;
        float2 *blk = static_cast<float2 *>(ghost_alloc(bytes));
        if (!blk)
        {
            for (size_t s = 0; s < ev->spec.size(); s++) ghost_free(ev->spec[s], ev->bytes[s]);
            delete ev;
            mErr = "out of device memory for the restart spectra";
            return false;
        }
        ev->spec.push_back(blk);
        ev->bytes.push_back(bytes);
        HCV_TRY(launch_rfft_frames(st->log2n, mGhostHist, mGhostLen, mGhostLen - 1, mN / st->M, 2, (int) rows.size(), blk, 2, st->tw, &st->big, mStream));
    }
    for (size_t pair : pairs)
    {
        mGhostOf[pair] = ev;
        ev->refs++;
    }
    mGhostEvents.push_back(ev);
    mCtlDirty = true;
    return true;
}

// spectral_mac + the ghost products of the restarted pairs it reaches (every MAC of a stage goes through here)
bool Engine::mac(Stage &st, const MacShape &s, const MacPlan &pl, const float2 *H, float2 *Y, long long h_first, bool check, hipStream_t stream)
{
    HCV_TRY(launch_spectral_mac(s, pl, st.X, H, Y, st.hv, h_first, check, stream));
    if (st.gh_count && h_first + s.T - 1 >= st.gh_min_hr && h_first - st.gh_max_hr <= (long long) s.P)
        HCV_TRY(launch_ghost_mac(s, H, Y, h_first, st.gh_start, st.gh_ent, nullptr, stream));
    return true;
}

// Take what `pair` still has to deliver after mN out of the timelines: the hop each stage computed last, restricted to the
// pair, with the spectra as they are now (so before a set() replaces them).
bool Engine::retire_pair(size_t pair)
{
    const uint32_t o = (uint32_t) (pair / mNinAlloc), c = (uint32_t) (pair % mNinAlloc);
    const uint32_t row = mCfg.diag ? o : c;
    if (mN <= 0 || o >= mLastNout || row >= mLastNin || !exact_restart()) return true;
    for (size_t si = 0; si < mStages.size(); si++)
    {
        Stage &st = *mStages[si];
        const long long h_r = mN / st.M;
        const long long P = std::min<long long>(st.pact[pair], h_r);
        if (P <= 0) continue;
        if (mTailHeadPrev && si + 1 != mStages.size()) continue;       // whole-hop mode: the shorter stages have nothing pending
        if (!mRetireTmp)
        {
            uint32_t nmax = 0;
            for (Stage *sp : mStages) nmax = std::max(nmax, sp->N);
            HCV_TRY(hipMalloc(&mRetireTmp, sizeof(float) * nmax));
        }
        MacShape sh;
        sh.M = (int) st.M;
        sh.R = (int) st.R;
        sh.P = (int) P;
        sh.Pcap = (int) st.Pcap;
        sh.nin = 1;
        sh.nin_alloc = 1;
        sh.nout = 1;
        sh.diag = 0;
        sh.T = 1;
        sh.max_ksplit = (int) std::max<size_t>(1, st.y_elems / st.M);
        sh.target_blocks = 0;
        MacPlan pl;
        mac_plan(sh, pl);
        float2 *Y = st.Yq[0];
        const float2 *H = st.Hs + pair * (size_t) st.Pcap * st.M;
        HCV_TRY(launch_spectral_mac(sh, pl, st.X + (size_t) row * st.R * st.M, H, Y, st.hv + pair, h_r - 1, true, mStream));
        for (int e = 0; e < st.gh_count; e++)
            if (st.gh_pair[e] == pair)
            {
                GhostEntry one = st.gh_host[e];
                one.i = 0;
                HCV_TRY(launch_ghost_mac(sh, H, Y, h_r - 1, nullptr, nullptr, &one, mStream));
            }
        HCV_TRY(launch_reduce_partials(Y, pl.ksplit, (long long) st.M, (long long) st.M, mStream));
        HCV_TRY(launch_rifft_rows(st.log2n, Y, 1, mRetireTmp, st.tw, &st.big, mStream));
        // the hop's result sits at (h_r - 1 + 1) * M ..; valid half of the frame, scale 1 / (4N) as rifft_overlap_add
        HCV_TRY(launch_timeline_sub(st.timeline + (size_t) o * st.tl_len, st.tl_len - 1, h_r * (long long) st.M, mRetireTmp + st.M, (int) st.M,
                                    1.f / (float) (8 * st.M), mN, mStream));
    }
    // the deferred accumulation for the hop in progress: the slices launched so far hold the pair's products over frames it
    // may no longer see — take them out of slot 0 (the slices still to come are fenced by hv like any other launch)
    for (size_t si = 0; si < mStages.size(); si++)
    {
        Stage &st = *mStages[si];
        if (st.pre_hop < 0 || st.bg_launched <= 0 || st.pact[pair] <= 1) continue;
        const int per = (st.bg_parts + st.bg_slices - 1) / st.bg_slices;
        const long long covered = std::min<long long>(std::min(st.bg_parts, st.bg_launched * per), (long long) st.pact[pair] - 1);
        if (covered <= 0) continue;
        MacShape sh;
        sh.M = (int) st.M;
        sh.R = (int) st.R;
        sh.P = (int) covered;
        sh.Pcap = (int) st.Pcap;
        sh.nin = 1;
        sh.nin_alloc = 1;
        sh.nout = 1;
        sh.diag = 0;
        sh.T = 1;
        sh.max_ksplit = (int) std::max<size_t>(1, st.y_elems / st.M);
        sh.target_blocks = 0;
        MacPlan pl;
        mac_plan(sh, pl);
        float2 *Y = st.Yq[0];
        const float2 *H = st.Hs + pair * (size_t) st.Pcap * st.M + st.M;          // partitions 1 .. covered at hop pre_hop - 1
        HCV_TRY(launch_spectral_mac(sh, pl, st.X + (size_t) row * st.R * st.M, H, Y, st.hv + pair, st.pre_hop - 1, true, mStream));
        for (int e = 0; e < st.gh_count; e++)
            if (st.gh_pair[e] == pair)
            {
                GhostEntry one = st.gh_host[e];
                one.i = 0;
                HCV_TRY(launch_ghost_mac(sh, H, Y, st.pre_hop - 1, nullptr, nullptr, &one, mStream));
            }
        HCV_TRY(launch_reduce_partials(Y, pl.ksplit, (long long) st.M, (long long) st.M, mStream));
        HCV_TRY(launch_timeline_sub(reinterpret_cast<float *>(st.Ypre + (size_t) o * st.M), -1LL, 0, reinterpret_cast<const float *>(Y), 2 * (int) st.M, 1.f,
                                    0, mStream));
    }
    mCtlDirty = true;
    return true;
}


bool Engine::apply_pending_resets()
{
    bool any = false, all = true;
    for (size_t p = 0; p < mPending.size(); p++)
    {
        any = any || mPending[p];
        if (mLoaded[p] && !mPending[p]) all = false;
    }
    if (!any) return true;
    if (!fence_background(!all && exact_restart())) return false;
    if (all)
    {
        if (!global_reset()) return false;
    }
    else
    {
        // Single pairs restart while the others keep running: take their pending output out of the timelines, fence them off
        // the input spectra older than the hop in progress, and prepare the ghost spectra that make the fence exact to the
        // sample (hcv_ghost.hip).  The time-domain head is fenced per sample directly.
        mCtlDirty = true;
        std::vector<size_t> restart;
        for (size_t p = 0; p < mPending.size(); p++)
        {
            if (!mPending[p]) continue;
            if (!mRetired[p] && !retire_pair(p)) return false;
            if (mLoaded[p]) restart.push_back(p);
            else release_ghost(p);
            for (Stage *st : mStages)
            {
                const long long hvv = mN / st->M;
                HCV_TRY(launch_fill_i64(st->hv + p, 1, hvv, mStream));
                st->max_hv = std::max(st->max_hv, hvv);
            }
            if (mTdValid)
            {
                HCV_TRY(launch_fill_i64(mTdValid + p, 1, mN, mStream));
                mTdMaxValid = std::max(mTdMaxValid, mN);
            }
        }
        if (!make_ghost_event(restart)) return false;
        if (!rebuild_ghost_tables()) return false;
    }
    std::fill(mRetired.begin(), mRetired.end(), 0);
    std::fill(mPending.begin(), mPending.end(), 0);
    return true;
}

// One block of at most max_block samples, everything device side.  Caller holds mMutex.
//
// Stream plan for block k (q = k & 1; every event and the FIR output buffer exist twice, indexed by block parity):
//
//   in stream:    wait readers(k-2) ─ scatter_input ─ record in[q]
//   stage s:      wait in[q], emit[q] (= emit of block k-2) ─ rfft_frames → spectral_mac → reduce → rifft_overlap_add ─ record done_s[q]
//   head stream:  wait in[q], emit[q]                       ─ fir_head → tdout[q]                                   ─ record td[q]
//   main stream:  wait done_s[q] for all s, td[q] ─ emit(tdout[q]) ─ record emit[q]
//
// The stages only meet in emit(), so the latency-bound short stages and the FIR head hide under the HBM-bound tail;
// and because block k+1's scatter and FFTs do not wait for block k's emit, consecutive asynchronous calls overlap.
// Ring depths make that safe: the history ring holds three blocks + a frame (a block's readers must be done before
// the block two later is scattered over them), each stage timeline holds two blocks + a hop (emit(k-2) must have
// cleared what block k's hops are added into).
bool Engine::enqueue_chunk(const float *din, int64_t in_stride, float *dout, int64_t out_stride, uint32_t nin_act, uint32_t nout_act, uint32_t B)
{
    const long long n0 = mN;
    const long long hmask = mHistLen - 1;
    const uint32_t rows_in = mCfg.diag ? nout_act : nin_act;
    const int q = (int) (mBlockCount & 1);

    const bool td_any = mCfg.has_td && mTdLpad > 0;
    const bool td_check = mTdMaxValid > 0 && (n0 - (long long) mTdLpad < mTdMaxValid);
    // Whole-hop mode: the block is made of whole, aligned hops of the last stage.  Every output sample of such a block only
    // needs inputs that the last stage's own frames hold, so IR[0 : its hop) is served by ONE extra zero-latency partition
    // of that stage (emitted in the hop's own slot, like the head-through-FFT of the first stage) and the head and the
    // shorter stages — a dozen launches in latency-bound chains — are not run at all.  Entering the mode drops their
    // pending (now duplicate) results; leaving it rebuilds their input spectra from the history ring and catches up on
    // the one hop whose result is due in the new block (see the stage loop).
    const size_t last = mStages.empty() ? 0 : mStages.size() - 1;
    const bool whole_hops = mTailHead && !td_check && (n0 % mStages[last]->M) == 0 && (B % mStages[last]->M) == 0 &&
                            !(mStages[last]->max_hv > n0 / mStages[last]->M);
    const bool entering = whole_hops && !mTailHeadPrev, leaving = !whole_hops && mTailHeadPrev;
    mTailHeadPrev = whole_hops;
    // hop-aligned block of a larger matrix: the head goes through the first stage's FFTs (see init)
    const bool head_fft = !whole_hops && td_any && mHeadFFT && !td_check && (n0 % mStages[0]->M) == 0 && (B % mStages[0]->M) == 0;
    const bool td = td_any && !head_fft && !whole_hops;

    // Serial blocks: everything on the main stream, in program order, with no events at all.  A dependency on a pending event
    // of another stream costs the host ~10 us (a kernel launch 2.6, an event record 1.7 — tools/micro/api_cost.hip), so a
    // small engine running one stage per block (whole-hop mode: 26 calls, five such dependencies, 111 us of host time for
    // 60 us of kernels on the 8x1 workload) is bound by its own enqueue; big engines keep the overlap.  HCV_SERIAL = 0 / 1
    // forces the choice for whole-hop blocks; single-stream engines are always serial.
    static const int serial_env = std::getenv("HCV_SERIAL") ? std::atoi(std::getenv("HCV_SERIAL")) : -1;
    static const double serial_mb = std::getenv("HCV_SERIAL_MB") ? std::atof(std::getenv("HCV_SERIAL_MB")) : 256.0;
    const bool serial = mOneStream || (whole_hops && (serial_env >= 0 ? serial_env != 0
                                                                        : (double) mStages[last]->live_parts * mStages[last]->M * sizeof(float2) < serial_mb * 1048576.0));
    hipStream_t sIn = serial ? mStream : mInStream, sTd = serial ? mStream : mTdStream;
    auto rec = [&](hipEvent_t e, hipStream_t s) -> hipError_t { return serial ? hipSuccess : hipEventRecord(e, s); };
    auto wt = [&](hipStream_t s, hipEvent_t e) -> hipError_t { return serial ? hipSuccess : hipStreamWaitEvent(s, e, 0); };
    if (serial && !mPrevSerial)
    {
        // the previous block's emit waited for all of its work, so the main stream is already behind everything except
        // background slices still in flight on a stage's own stream
        for (Stage *st : mStages)
            if (st->bg_pending) HCV_TRY(hipStreamWaitEvent(mStream, st->bg_done, 0));
    }
    else if (!serial && mPrevSerial)
    {
        // back to the streams: they start behind everything the serial blocks put on the main stream
        HCV_TRY(hipEventRecord(mEvSerial, mStream));
        HCV_TRY(hipStreamWaitEvent(mInStream, mEvSerial, 0));
        HCV_TRY(hipStreamWaitEvent(mTdStream, mEvSerial, 0));
        for (Stage *st : mStages) HCV_TRY(hipStreamWaitEvent(st->stream, mEvSerial, 0));
    }
    mPrevSerial = serial;
    if (mGhostPruneAt >= 0 && n0 >= mGhostPruneAt && !prune_ghosts()) return false;
    // control work queued on the main stream (IR spectra, reset fills, regrown buffers) must land before this block
    if (mCtlDirty)
    {
        HCV_TRY(rec(mEvCtl, mStream));
        HCV_TRY(wt(sIn, mEvCtl));
        // (deferred slices are launched without waiting for the block's input: order them after the control work directly)
        for (Stage *st : mStages) HCV_TRY(wt(st->stream, mEvCtl));
        mCtlDirty = false;
    }
    // HCV_PIPELINE=0 serialises consecutive blocks (block k+1 starts after emit(k)); default lets them overlap
    static const bool pipeline = !(std::getenv("HCV_PIPELINE") && std::atoi(std::getenv("HCV_PIPELINE")) == 0);
    if (!pipeline) HCV_TRY(wt(sIn, mEvEmit[q ^ 1]));
    // the block two back read the history this scatter may overwrite
    for (Stage *st : mStages) HCV_TRY(wt(sIn, st->done[q]));
    HCV_TRY(wt(sIn, mEvTd[q]));
    HCV_TRY(launch_scatter_input(din, in_stride, (int) B, (int) rows_in, mHist, mHistLen, hmask, n0, sIn));
    HCV_TRY(rec(mEvInput[q], sIn));

    if (td)
    {
        HCV_TRY(wt(sTd, mEvInput[q]));
        HCV_TRY(wt(sTd, mEvEmit[q]));      // emit(k-2) has consumed tdout[q]
        const bool check = td_check;
        HCV_TRY(launch_fir_head(mHist, mHistLen, hmask, mTaps, (int) mTdLpad, 2048, (int) nin_act, (int) mNinAlloc, (int) nout_act, mCfg.diag ? 1 : 0,
                                n0, (int) B, mTdValid, check, mTdOut[q], mMaxBlock, sTd));
        HCV_TRY(rec(mEvTd[q], sTd));
    }

    EmitSources src;
    src.count = 0;
    // Tail gate: when this block carries a hop of a long, bandwidth-bound tail stage, the shorter stages' MACs are held
    // until the tail's spectral_mac has finished.  That kernel is one wave of workgroups balanced over every CU and
    // streams through the caches; short-stage MACs whose spectra do not fit those caches, run beside it, are evicted by
    // it, re-read from HBM and take slots from some of its workgroups — together they take longer than one after the
    // other (64x64 with 10 s IRs: tail alone 2.29 ms at 6.9 TB/s + short stages 0.33 ms, against 2.97 ms overlapped).
    // When the short stages fit the caches (16x16) or the tail is short (64x64 with 2 s IRs) the overlap wins and is
    // kept.  HCV_TAIL_GATE = 0 / 1 forces the choice (2: hold the forward FFTs too).
    static const int tail_gate_env = std::getenv("HCV_TAIL_GATE") ? std::atoi(std::getenv("HCV_TAIL_GATE")) : -1;
    int tail_gate = whole_hops ? 0 : tail_gate_env;
    if (tail_gate < 0)
    {
        double small_bytes = 0, small_traffic = 0, tail_bytes = 0;
        for (size_t si = 0; si < mStages.size(); si++)
        {
            const Stage &sg = *mStages[si];
            const double bytes = (double) sg.live_parts * sg.M * sizeof(float2);
            const double hops = (double) ((n0 + B) / sg.M - n0 / sg.M);
            if (si + 1 == mStages.size()) tail_bytes = bytes * std::max(1.0, hops / 8.0);
            else
            {
                small_bytes += bytes;
                small_traffic += bytes * std::max(1.0, hops / 4.0);       // hop tiles of 4 share one read of the spectra
            }
        }
        // measured only to pay when the block carries exactly one tail hop (at two hops per block the hop-tiled tail already
        // shares the chip better: ns64 at 16384-sample blocks 376 ungated vs 357 gated Msamples/s)
        const bool one_tail_hop = !mStages.empty() && (n0 + B) / mStages.back()->M - n0 / mStages.back()->M == 1;
        tail_gate = (one_tail_hop && small_bytes >= 64.0 * 1048576.0 && tail_bytes >= 12.0 * small_traffic) ? 1 : 0;
    }
    hipEvent_t gate = nullptr;

    // Launch the background slices of `st` that are due: all of them at the hop's boundary, otherwise in proportion to the
    // part of the hop's samples that has arrived with this call.  Slice s covers partitions 1 + [a, b) of hop pre_hop:
    // a (b - a)-partition MAC at hop pre_hop - 1 - a over the spectra shifted by 1 + a partitions.
    auto advance_background = [&](Stage &st, bool boundary) -> bool
    {
        const hipStream_t sS = serial ? mStream : st.stream;
        if (st.pre_hop < 0 || st.bg_launched >= st.bg_slices) return true;
        const long long into = (long long) (n0 + B) - st.pre_hop * (long long) st.M;
        int due = boundary ? st.bg_slices : (int) std::min<long long>(st.bg_slices, std::max<long long>(0, into * st.bg_slices / (long long) st.M));
        const int per = (st.bg_parts + st.bg_slices - 1) / st.bg_slices;
        const long long slot_elems = (long long) mCfg.nout * st.M;
        for (; st.bg_launched < due; st.bg_launched++)
        {
            const int a = st.bg_launched * per, b = std::min(st.bg_parts, a + per);
            float2 *slot = st.Ypre + (long long) st.bg_launched * slot_elems;
            if (b <= a)
            {
                HCV_TRY(hipMemsetAsync(slot, 0, sizeof(float2) * slot_elems, sS));
                continue;
            }
            MacShape sb;
            sb.M = (int) st.M;
            sb.R = (int) st.R;
            sb.P = b - a;
            sb.Pcap = (int) st.Pcap;
            sb.nin = (int) nin_act;                                 // slices only run for the full matrix
            sb.nin_alloc = (int) mNinAlloc;
            sb.nout = (int) mCfg.nout;
            sb.diag = mCfg.diag ? 1 : 0;
            sb.T = 1;
            sb.max_ksplit = (int) std::max<size_t>(1, st.y_elems / ((size_t) mCfg.nout * st.M));
            sb.target_blocks = 0;
            MacPlan pb;
            mac_plan(sb, pb);
            const long long hop = st.pre_hop - 1 - a;
            const bool bcheck = (hop - st.max_hv) < (long long) (b - a) - 1;
            float2 *scratch = st.Yq[0];                             // every use of this stage's scratch is ordered on its stream
            if (!mac(st, sb, pb, st.Hs + (size_t) (1 + a) * st.M, scratch, hop, bcheck, sS)) return false;
            HCV_TRY(launch_reduce_partials(scratch, pb.ksplit, slot_elems, slot_elems, sS));
            HCV_TRY(hipMemcpyAsync(slot, scratch, sizeof(float2) * slot_elems, hipMemcpyDeviceToDevice, sS));
            HCV_TRY(hipEventRecord(st.bg_done, sS));
            st.bg_pending = true;
        }
        return true;
    };

    // largest stage first: the tail's spectral_mac is the critical path, the short stages fill in around it
    for (size_t sj = 0; sj < mStages.size(); sj++)
    {
        const size_t si = mStages.size() - 1 - sj;
        Stage &st = *mStages[si];
        const hipStream_t sS = serial ? mStream : st.stream;
        if (whole_hops && si != last)
        {
            if (entering)
            {
                // this stage's pending results duplicate what the last stage now computes: drop them (after the emit that
                // may still be reading them) together with any plan of a deferred accumulation
                HCV_TRY(wt(sS, mEvEmit[q ^ 1]));
                HCV_TRY(hipMemsetAsync(st.timeline, 0, sizeof(float) * mCfg.nout * st.tl_len, sS));
                HCV_TRY(rec(st.done[q], sS));
                HCV_TRY(wt(mStream, st.done[q]));
                st.pre_hop = -1;
            }
            continue;
        }
        src.timeline[src.count] = st.timeline;              // the ring may still hold hops of earlier calls
        src.stride[src.count] = st.tl_len;
        src.mask[src.count] = st.tl_len - 1;
        src.count++;
        const bool tail_head_here = whole_hops && si == last;
        int head_ksplit = 1;
        bool head_on_side = false;
        const bool head_here = (head_fft && si == 0) || tail_head_here;
        const float2 *head_spec = tail_head_here ? mTailHeadSpec : mHeadSpec;
        float2 *head_y = tail_head_here ? mTailHeadYq[q] : mHeadYq[q];
        if (!st.P && !head_here) continue;
        const long long h_first = n0 / st.M;
        const int T = (int) ((n0 + B) / st.M - h_first);
        if (leaving && si != last && st.P && h_first >= 1)
        {
            HCV_TRY(wt(sS, mEvInput[q]));
            st.Y = st.Yq[q];
            // back from whole-hop mode: this stage was not run for a while.  Rebuild the input spectra its partitions reach
            // back to from the history ring, and compute the hop just before this block — its result is emitted during the
            // first hop of the block (every stage has one hop of latency).
            const long long h_lo = std::max<long long>(0, h_first - (long long) st.P);
            HCV_TRY(launch_rfft_frames(st.log2n, mHist, mHistLen, hmask, h_lo, (int) (h_first - h_lo), (int) rows_in, st.X, (int) st.R, st.tw, &st.big, sS));
            MacShape sc;
            sc.M = (int) st.M;
            sc.R = (int) st.R;
            sc.P = (int) std::min<long long>(st.P, h_first);
            sc.Pcap = (int) st.Pcap;
            sc.nin = (int) nin_act;
            sc.nin_alloc = (int) mNinAlloc;
            sc.nout = (int) nout_act;
            sc.diag = mCfg.diag ? 1 : 0;
            sc.T = 1;
            sc.max_ksplit = (int) std::max<size_t>(1, st.y_elems / ((size_t) nout_act * st.M));
            sc.target_blocks = 0;
            MacPlan pc;
            mac_plan(sc, pc);
            const bool ccheck = (h_first - 1 - st.max_hv) < (long long) st.P - 1;
            const long long c_elems = (long long) nout_act * st.M;
            if (!mac(st, sc, pc, st.Hs, st.Y, h_first - 1, ccheck, sS)) return false;
            HCV_TRY(launch_reduce_partials(st.Y, pc.ksplit, c_elems, c_elems, sS));
            HCV_TRY(wt(sS, mEvEmit[q]));        // emit(k-2) has cleared the timeline span reused now
            HCV_TRY(launch_rifft_overlap_add(st.log2n, st.Y, 1, c_elems, h_first - 1, 1, (int) nout_act, st.timeline, st.tl_len, st.tl_len - 1, st.tw,
                                             &st.big, sS));
            HCV_TRY(rec(st.done[q], sS));          // (recorded again below when this block has hops of its own)
            HCV_TRY(wt(mStream, st.done[q]));
        }
        const bool full_matrix = nout_act == mCfg.nout && nin_act == (mCfg.diag ? mCfg.nout : mCfg.nin);
        // Deferred mode: calls shorter than the hop (real-time block sizes) with more than one partition.
        static const bool allow_defer = !(std::getenv("HCV_DEFER") && std::atoi(std::getenv("HCV_DEFER")) == 0);
        static const int slices_env = std::getenv("HCV_BG_SLICES") ? std::atoi(std::getenv("HCV_BG_SLICES")) : kBgSlices;
        if (T <= 0)
        {
            if (st.pre_hop >= 0 && (!full_matrix || st.pre_hop != h_first)) st.pre_hop = -1;     // the plan no longer fits what is being processed
            if (st.pre_hop < 0 && allow_defer && full_matrix && st.P > 1 && B < st.M && h_first >= 1)
            {
                // no plan for the hop in progress (the first small call after large ones, or control work dropped it): make it
                // now — the frames it needs are complete — so that the boundary does not pay the whole accumulation inline
                st.bg_parts = (int) std::min<long long>(st.P - 1, h_first);
                st.bg_slices = std::max(1, std::min(std::min(kBgSlices, slices_env), st.bg_parts));
                st.bg_launched = 0;
                st.pre_hop = h_first;
            }
            if (st.pre_hop >= 0 && !advance_background(st, false)) return false;
            continue;
        }

        // one stream per stage: forward FFT, MAC, inverse.  (Side streams for a large stage's FFTs were built and measured
        // twice — dedicated ones and the input / main streams — and were slower each time: c5 2.12 / 2.33 vs 1.98 ms per step.)
        hipStream_t sM = sS, sF = sS, sI = sS;
        st.Y = st.Yq[q];

        HCV_TRY(wt(sF, mEvInput[q]));
        if (gate && tail_gate >= 2) HCV_TRY(wt(sF, gate));
        HCV_TRY(launch_rfft_frames(st.log2n, mHist, mHistLen, hmask, h_first, T, (int) rows_in, st.X, (int) st.R, st.tw, &st.big, sF));
        if (gate && tail_gate == 1) HCV_TRY(wt(sM, gate));
        if (head_here)
        {
            // head = partition "-1": Yh[t][o] = sum_i X[i][h_t] * Hhead[o][i]
            MacShape hs;
            hs.M = (int) st.M;
            hs.R = (int) st.R;
            hs.P = 1;
            hs.Pcap = 1;
            hs.nin = (int) nin_act;
            hs.nin_alloc = (int) mNinAlloc;
            hs.nout = (int) nout_act;
            hs.diag = mCfg.diag ? 1 : 0;
            hs.T = T;
            hs.max_ksplit = tail_head_here ? kTailHeadSplit : 1;
            hs.target_blocks = 0;
            MacPlan hp;
            mac_plan(hs, hp);
            static const bool head_side = !(std::getenv("HCV_HEAD_STREAM") && std::atoi(std::getenv("HCV_HEAD_STREAM")) == 0);
            if (tail_head_here && head_side && !serial)
            {
                // whole-hop mode: the head partition's MAC, reduction and inverse run on the otherwise idle head stream, beside
                // the tail MAC, so the last stage's own stream carries only FFT -> MAC -> reduce -> inverse (c4 853 -> 917,
                // c5 66.4 -> 68.2, c3 72 -> 84 Msamples/s).  Both inverses add into the same timeline; the adds are atomic.
                HCV_TRY(rec(st.mac_done[q], sM));                       // = "forward FFTs of this block are done"
                HCV_TRY(wt(sTd, st.mac_done[q]));
                HCV_TRY(wt(sTd, mEvEmit[q]));
                const long long he = (long long) T * nout_act * st.M;
                if (!mac(st, hs, hp, head_spec, head_y, h_first, false, sTd)) return false;
                HCV_TRY(launch_reduce_partials(head_y, hp.ksplit, he, he, sTd));
                HCV_TRY(launch_rifft_overlap_add(st.log2n, head_y, 1, 0, h_first - 1, T, (int) nout_act, st.timeline, st.tl_len, st.tl_len - 1, st.tw,
                                                 &st.big, sTd));
                HCV_TRY(rec(mEvTd[q], sTd));
                head_on_side = true;
            }
            else
            {
                if (!mac(st, hs, hp, head_spec, head_y, h_first, false, sM)) return false;
                head_ksplit = hp.ksplit;
            }
        }

        // hops since the last global reset bound how many partitions can have input yet (mValidPartitions in the
        // reference, PartitionedConvolve.cpp:285,322,373): right after a reset the reduction is short
        const long long p_live = std::min<long long>(st.P, h_first + T);
        MacShape sh;
        sh.M = (int) st.M;
        sh.R = (int) st.R;
        sh.P = (int) p_live;
        sh.Pcap = (int) st.Pcap;
        sh.nin = (int) nin_act;
        sh.nin_alloc = (int) mNinAlloc;
        sh.nout = (int) nout_act;
        sh.diag = mCfg.diag ? 1 : 0;
        sh.T = T;
        sh.max_ksplit = (int) std::max<size_t>(1, st.y_elems / ((size_t) T * nout_act * st.M));
        sh.target_blocks = 0;
        const bool check = (h_first - st.max_hv) < (long long) st.P - 1;
        const long long y_elems = (long long) T * nout_act * st.M;

        const bool defer = allow_defer && st.P > 0 && T == 1 && B < st.M && st.P > 1 && p_live >= 1 && nout_act == mCfg.nout &&
                           nin_act == (mCfg.diag ? mCfg.nout : mCfg.nin);
        const bool have_pre = defer && st.pre_hop == h_first;

        EventPair *ev = nullptr;
        auto begin_event = [&]() -> bool
        {
            if (!mProfiling) return true;
            for (EventPair *c : mEvents)
                if (!c->live) { ev = c; break; }
            if (!ev)
            {
                ev = new EventPair();
                HCV_TRY(hipEventCreate(&ev->a));
                HCV_TRY(hipEventCreate(&ev->b));
                mEvents.push_back(ev);
            }
            ev->stage = si;
            ev->live = true;
            HCV_TRY(hipEventRecord(ev->a, sM));
            return true;
        };

        // ---- MAC phase (stream sM)
        MacPlan pl;
        pl.ksplit = 1;
        if (st.P)
        {
            if (have_pre)
            {
                // boundary of a hop whose partitions 1..P-1 were accumulated in the background: whatever slices are still
                // due, their total into slot 0, then partition 0 only
                if (!advance_background(st, true)) return false;
                HCV_TRY(launch_reduce_partials(st.Ypre, st.bg_slices, (long long) mCfg.nout * st.M, (long long) mCfg.nout * st.M, sM));
                MacShape s0 = sh;
                s0.P = 1;
                s0.max_ksplit = 1;
                mac_plan(s0, pl);
                if (!mac(st, s0, pl, st.Hs, st.Y, h_first, check, sM)) return false;
            }
            else
            {
                mac_plan(sh, pl);
                if (!begin_event()) return false;
                if (!mac(st, sh, pl, st.Hs, st.Y, h_first, check, sM)) return false;
                if (ev) HCV_TRY(hipEventRecord(ev->b, sM));
                st.launches++;
                st.hops += (uint64_t) T;
                st.last_ksplit = (uint32_t) pl.ksplit;
                st.last_ot = (uint32_t) pl.ot;
            }
        }
        if (tail_gate && sj == 0 && mStages.size() > 1 && !mOneStream && st.P && !defer && !have_pre)
        {
            HCV_TRY(rec(st.mac_done[q], sM));
            gate = st.mac_done[q];
        }

        // ---- inverse phase (stream sI): every read-modify-write of this stage's timeline happens on this stream
        HCV_TRY(wt(sI, mEvEmit[q]));             // emit(k-2) has cleared the timeline span reused now
        if (head_here && !head_on_side)
        {
            HCV_TRY(launch_reduce_partials(head_y, head_ksplit, (long long) T * nout_act * st.M, (long long) T * nout_act * st.M, sI));
            HCV_TRY(launch_rifft_overlap_add(st.log2n, head_y, 1, 0, h_first - 1, T, (int) nout_act, st.timeline, st.tl_len, st.tl_len - 1, st.tw,
                                             &st.big, sI));           // h_first - 1: emitted with NO latency (hop h at h*M)
        }
        if (st.P)
        {
            if (have_pre)
            {
                // inverse FFT of Y (partition 0) + Ypre (partitions >= 1): the two buffers are read as two "partials"
                if (is_big_fft(st.log2n))
                {
                    HCV_TRY(launch_reduce_partials(st.Y, 2, (long long) (st.Ypre - st.Y), y_elems, sI));
                    HCV_TRY(launch_rifft_overlap_add(st.log2n, st.Y, 1, y_elems, h_first, 1, (int) nout_act, st.timeline, st.tl_len, st.tl_len - 1,
                                                     st.tw, &st.big, sI));
                }
                else
                    HCV_TRY(launch_rifft_overlap_add(st.log2n, st.Y, 2, (long long) (st.Ypre - st.Y), h_first, 1, (int) nout_act, st.timeline, st.tl_len,
                                                     st.tl_len - 1, st.tw, &st.big, sI));
            }
            else
            {
                HCV_TRY(launch_reduce_partials(st.Y, pl.ksplit, y_elems, y_elems, sI));
                HCV_TRY(launch_rifft_overlap_add(st.log2n, st.Y, 1, y_elems, h_first, T, (int) nout_act, st.timeline, st.tl_len, st.tl_len - 1, st.tw,
                                                 &st.big, sI));
            }
        }
        HCV_TRY(rec(st.done[q], sI));
        HCV_TRY(wt(mStream, st.done[q]));

        st.pre_hop = -1;
        if (defer)
        {
            // plan the background accumulation for hop h+1: partitions 1 .. min(P-1, h+1) (those that have input), in up to
            // kBgSlices launches that the following calls issue as the hop's samples arrive (advance_background)
            st.bg_parts = (int) std::min<long long>(st.P - 1, h_first + 1);
            st.bg_slices = std::max(1, std::min(std::min(kBgSlices, slices_env), st.bg_parts));
            st.bg_launched = 0;
            st.pre_hop = st.bg_parts > 0 ? h_first + 1 : -1;
        }
    }

    if (td || whole_hops) HCV_TRY(wt(mStream, mEvTd[q]));
    HCV_TRY(wt(mStream, mEvInput[q]));           // a block with no live stage still orders after its scatter
    HCV_TRY(launch_emit(src, n0, (int) B, (int) nout_act, td ? mTdOut[q] : nullptr, mMaxBlock, dout, out_stride, mStream));
    HCV_TRY(rec(mEvEmit[q], mStream));
    mN += B;
    mBlockCount++;
    mLastNin = rows_in;
    mLastNout = nout_act;
    return true;
}

bool Engine::process(const float *const *ins, float *const *outs, uint32_t nin_act, uint32_t nout_act, uint64_t n, bool accumulate)
{
    HCV_TRY(hipSetDevice(mDevice));
    nout_act = std::min(nout_act, mCfg.nout);
    nin_act = std::min(nin_act, mCfg.nin);
    if (!nout_act || !n) return true;
    const uint32_t rows_in = mCfg.diag ? nout_act : nin_act;

    for (uint64_t pos = 0; pos < n; pos += mMaxBlock)
    {
        const uint32_t B = (uint32_t) std::min<uint64_t>(mMaxBlock, n - pos);
        for (uint32_t i = 0; i < rows_in; i++) std::memcpy(mPinIn + (size_t) i * B, ins[i] + pos, sizeof(float) * B);
        {
            std::lock_guard<std::mutex> g(mMutex);
            if (!apply_pending_resets()) return false;
            // the upload goes on the main stream and is handed to the block like control work: a serial block scatters on the
            // main stream, a streamed one makes its input stream wait for it (enqueue_chunk, mCtlDirty)
            if (rows_in)
            {
                HCV_TRY(hipMemcpyAsync(mDevIn, mPinIn, sizeof(float) * rows_in * B, hipMemcpyHostToDevice, mStream));
                mCtlDirty = true;
            }
            if (!enqueue_chunk(mDevIn, B, mDevOut, B, nin_act, nout_act, B)) return false;
            HCV_TRY(hipMemcpyAsync(mPinOut, mDevOut, sizeof(float) * nout_act * B, hipMemcpyDeviceToHost, mStream));
        }
        HCV_TRY(hipStreamSynchronize(mStream));
        for (uint32_t o = 0; o < nout_act; o++)
        {
            float *dst = outs[o] + pos;
            const float *src = mPinOut + (size_t) o * B;
            if (accumulate)
                for (uint32_t j = 0; j < B; j++) dst[j] += src[j];
            else
                std::memcpy(dst, src, sizeof(float) * B);
        }
    }
    if (mProfiling) collect_events();
    return true;
}

bool Engine::process_dev(const float *ins, int64_t in_stride, float *outs, int64_t out_stride, uint32_t nin_act, uint32_t nout_act, uint64_t n,
                         bool sync)
{
    HCV_TRY(hipSetDevice(mDevice));
    nout_act = std::min(nout_act, mCfg.nout);
    nin_act = std::min(nin_act, mCfg.nin);
    if (!nout_act || !n) return true;
    {
        std::lock_guard<std::mutex> g(mMutex);
        if (!apply_pending_resets()) return false;
        for (uint64_t pos = 0; pos < n; pos += mMaxBlock)
        {
            const uint32_t B = (uint32_t) std::min<uint64_t>(mMaxBlock, n - pos);
            if (!enqueue_chunk(ins + pos, in_stride, outs + pos, out_stride, nin_act, nout_act, B)) return false;
        }
    }
    if (sync) return synchronize();
    return true;
}

bool Engine::synchronize()
{
    HCV_TRY(hipSetDevice(mDevice));
    HCV_TRY(hipStreamSynchronize(mStream));
    if (mProfiling) collect_events();
    return true;
}

void Engine::set_profiling(bool on)
{
    std::lock_guard<std::mutex> g(mMutex);
    mProfiling = on;
}

void Engine::collect_events()
{
    std::lock_guard<std::mutex> g(mMutex);
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
    std::lock_guard<std::mutex> g(mMutex);
    if (s >= mStages.size() || !out) return false;
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
    return true;
}

void Engine::clear_stats()
{
    std::lock_guard<std::mutex> g(mMutex);
    for (Stage *st : mStages)
    {
        st->launches = st->hops = 0;
        st->ms = 0.0;
    }
}

} // namespace hcv
