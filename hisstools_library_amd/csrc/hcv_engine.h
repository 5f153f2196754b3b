// Device-resident N x M multi-stage convolution engine (internal).
//
// One Engine replaces a whole tree of reference objects:
//     Convolver -> NToMonoConvolve[out] -> MonoConvolve[in] -> {TimeDomainConvolve, PartitionedConvolve x <=4}
// (Convolver.cpp:17-18, NToMonoConvolve.cpp:7-8, MonoConvolve.cpp:235-252).  Where the reference keeps one
// input-spectrum ring, one forward FFT and one inverse FFT PER (in,out) PAIR PER STAGE, the engine keeps one
// input ring and one forward FFT per INPUT and one inverse FFT per OUTPUT per stage, and accumulates over
// inputs and partitions in the frequency domain.
//
// HBM layout per FFT stage s (N = fft size, M = N/2 bins, float2 = interleaved complex bin):
//     Hs [nout][nin_alloc][lead + Pcap][M] float2   IR partition spectra, bin-contiguous (streamed by spectral_mac); lead = 1 on
//                                              the last stage of a zero-latency ladder: slot 0 = spectrum of IR[0 : M), whole-hop mode
//     X  [nin][R][M]               float2      ring of the last R input spectra per input (R >= Pcap + Tmax)
//     Y  [ksplit][T][nout][M]      float2      split-K partial sums of one process call
//     hv [nout][nin_alloc]         int64       first hop each pair may see (per-pair reset)
// plus per engine:
//     hist     [nin][RH]  float                input history ring (overlap-save frames + FIR history)
//     timeline [nout][OR] float (per stage)    output timeline ring; the stage ADDS its hop results at their
//                                              emission time, emit() sums the stages' rings and clears the block
//     taps     [nout][nin_alloc][2048] float   time-domain head taps, zero padded
//     ghost spectra, pooled                    per restart of single pairs and stage: [inputs restarted][2][M] float2 (hcv_ghost.hip)
//
// Control calls stage their work beside the audio thread (set_ir: upload + FFTs into staging buffers on a control stream, then
// a swap section; ensure_stage_capacity: new buffers filled beside the running audio thread); process() never waits for them:
// the engine's host state has an owner, not a lock (Engine::mOwner).
//
// Source files: hcv_engine.hip (set-up, IR loading, capacity growth), hcv_engine_block.hip (the per-block scheduler:
// enqueue_chunk -> enqueue_stage, streams, serial blocks, deferred slices), hcv_engine_restart.hip (exact per-pair restart:
// ghost spectra, retiring, resets, active-matrix changes); private structures in hcv_engine_impl.h.
#pragma once

#include "hcv_kernels.h"

#include <atomic>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace hcv
{
    long long order_violations();   // hcv_engine.hip: violations counted by HCV_ORDER_CHECK so far, -1 = the check is off
    class OrderCheck;               // hcv_order_check.h: HCV_ORDER_CHECK=1, the stream / event order asserted at enqueue time
    // one step of a bounded spin-wait on the host (the audio thread's lock poll, the shard pool's hand-offs)
    inline void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#elif defined(__aarch64__)
        __asm__ __volatile__("yield" ::: "memory");
#else
        std::this_thread::yield();
#endif
    }

    struct StageCfg
    {
        uint32_t fft_size = 0;      // power of two, 2^5 .. 2^20
        uint64_t offset = 0;        // first IR sample of this stage's segment
        uint64_t length = 0;        // segment length limit, 0 = "to the end of the IR"
        uint64_t capacity = 0;      // IR samples this stage can hold (rounded up to a multiple of fft_size/2)
    };

    struct EngineCfg
    {
        uint32_t nin = 1, nout = 1;
        bool diag = false;          // parallel mode: output o is fed by input o only (Convolver.cpp:24-41)
        bool has_td = false;
        uint64_t td_offset = 0, td_length = 0;   // td_length 0 = up to 2044 taps (TimeDomainConvolve.cpp:62-87)
        std::vector<StageCfg> stages;
        int device = -1;            // -1 = current default (HCV_DEVICE env or 0)
        uint32_t max_block = 0;     // 0 = default (HCV_MAX_BLOCK env or 32768)
        int pivot = -1;             // >= 0: the stages behind this index are far-tail rungs of the extended ladder (hcv_api.hip: Layout::extend_tail):
                                    // whole-hop mode runs on stage `pivot` and the rungs keep their own (deferred) schedule beside it.  -1 = the last stage
    };

    struct StageStats
    {
        uint32_t fft_size, partitions, nin, nout;
        uint64_t mac_launches, mac_hops;
        double mac_ms;              // summed HIP-event time of this stage's spectral_mac launches (profiling on)
        uint32_t ksplit, out_tile;
        uint64_t mac_steady_launches;   // of mac_launches: the unchecked instantiation with nontemporal IR loads
        uint32_t hop_tile, launch_partitions;
        uint64_t fused_launches;        // of mac_launches: whole blocks run as ONE launch (hcv_fft_split.hip: fused_block_*_kernel)
        uint64_t fused_stood_down;      // times the n x m block was stood down (64 .. 4096 blocks: its forward launches kept arriving late)
        uint64_t host_pre_launches;     // hop-sized host-pointer blocks whose partitions >= 1 went out ahead of the upload (host_pre_mac)
    };

    class Engine
    {
    public:
        static Engine *create(const EngineCfg &cfg, std::string *err);
        ~Engine();

        uint32_t nin() const { return mCfg.nin; }
        uint32_t nout() const { return mCfg.nout; }
        bool diag() const { return mCfg.diag; }
        size_t num_stages() const { return mStages.size(); }
        uint64_t stage_capacity(size_t s) const;
        uint32_t stage_partitions(size_t s, uint32_t in, uint32_t out) const;
        uint32_t td_taps(uint32_t in, uint32_t out) const;

        // change the IR window a stage / the head takes at the NEXT set_ir (PartitionedConvolve::setOffset/setLength)
        void set_stage_window(size_t s, uint64_t offset, uint64_t length);
        void set_td_window(uint64_t offset, uint64_t length);

        // grow the (last) stage so it can hold `capacity` IR samples; false on allocation failure
        bool ensure_stage_capacity(size_t s, uint64_t capacity);

        // load / clear one pair's IR in every stage (ir == nullptr or len == 0 clears).  Blocks until the device has
        // consumed `ir`.  device_ptr: ir is device memory on this engine's GPU.
        bool set_ir(uint32_t in, uint32_t out, const float *ir, uint64_t len, bool device_ptr);

        void reset_pair(uint32_t in, uint32_t out);     // consumed at the next process (MonoConvolve.cpp:148-152,185-193)
        void reset_all();

        // host-pointer streaming call.  outs[o] is overwritten (accumulate=false) or added to.
        bool process(const float *const *ins, float *const *outs, uint32_t nin_act, uint32_t nout_act, uint64_t n, bool accumulate);
        // the same in two halves for one block of at most max_block() samples, so that several engines (shards) overlap:
        // begin = stage the inputs, enqueue the block and its download; end = wait for the download, deliver the samples
        bool process_begin(const float *const *ins, uint32_t nin_act, uint32_t nout_act, uint32_t B);
        bool process_end(float *const *outs, uint32_t nout_act, uint32_t B, bool accumulate);
        // device-resident call: ins/outs are [rows][stride] float on this GPU.  Asynchronous unless sync=true.
        // `after` (optional): an event of ANOTHER engine or device that every kernel of this call which writes `outs` must follow
        // (the sharded object's row root has then read the previous contents of a partial block, hcv_api.hip)
        bool process_dev(const float *ins, int64_t in_stride, float *outs, int64_t out_stride, uint32_t nin_act, uint32_t nout_act, uint64_t n,
                         bool sync, hipEvent_t after = nullptr);
        // synchronous call on PINNED host memory the caller registered (hcv_host_register): [rows][stride] blocks given by their
        // host address and their device mapping.  Small blocks run on the mapping in place; larger ones are moved by the copy
        // engines straight from / to the caller's memory (no staging memcpy on the host).
        bool process_pinned(const float *ins_host, const float *ins_map, int64_t in_stride, float *outs_host, float *outs_map, int64_t out_stride,
                            uint32_t nin_act, uint32_t nout_act, uint64_t n);
        bool synchronize();
        // for the layers that combine several engines (shards, collectives): the stream that ENDS every block — the emit launch, or
        // a wait for the stage stream whose inverse wrote the caller's block (direct output); work enqueued on it after a call
        // follows the call's output.  It is NOT the only writer of the output buffer: a streamed whole-hop block writes it from the
        // last stage's stream, which is why a dependency of the NEXT call's writes on foreign work goes through process_dev's `after`.
        hipStream_t main_stream() const { return mStream; }

        // The audio-thread contract (MemorySwap::attempt, MemorySwap.h:182-185; MonoConvolve.cpp:181-183; ThreadLocks.hpp:51-87): process
        // never waits for a control call.  There is no lock on the path: the engine's host state is OWNED — by the thread inside a process
        // call's enqueue, or, only while no stream is running, by a control thread inside a swap section (mOwner, one compare-exchange,
        // no loop).  While a stream is running control threads never take the ownership: they post their section and the audio thread
        // runs it between two of its blocks (mailbox_runs; what that cost it: mailbox_ns_max / mailbox_ns_total).  start_collisions counts
        // the one case left in which a process call cannot have the state: the FIRST call of a stream (no call for kStreamingWindowNs
        // before it) arriving while a control thread is inside a section it began during the pause — that call's block is silent, as every
        // pair under a set() is in the reference, and the call after it proceeds.  (A call of 2048 samples or more — an offline loop, not an
        // audio callback — waits the section out instead: start_waits.)
        struct RtStats { uint64_t start_collisions, mailbox_runs, mailbox_ns_max, mailbox_ns_total, ctl_sections, arena_misses, start_waits; };
        RtStats rt_stats() const { return { mStartCollisions.load(), mMailboxRuns.load(), mMailboxNsMax.load(), mMailboxNsTotal.load(), mCtlSections.load(), mArenaMisses.load(), mStartWaits.load() }; }
        void clear_rt_stats() { mStartCollisions = 0; mMailboxRuns = 0; mMailboxNsMax = 0; mMailboxNsTotal = 0; mCtlSections = 0; mArenaMisses = 0; mStartWaits = 0; }

        void set_profiling(bool on);
        // HCV_REFERENCE_QUIRKS (hcv_api.hip): the next blocks leave the time-domain head out — what MonoConvolve::process does to it in a
        // zero-latency layout of fewer than four FFT sizes (MonoConvolve.cpp:195-197: the stage behind the missing one overwrites it)
        void set_drop_head(bool on) { mDropHead.store(on, std::memory_order_relaxed); }
        bool profiling() const { return mProfiling; }
        bool stage_stats(size_t s, StageStats *out);
        void clear_stats();

        const std::string &last_error() const { return mErr; }
        int device() const { return mDevice; }
        uint32_t max_block() const { return mMaxBlock; }

    private:
        struct Stage;
        struct EventPair;
        Engine() = default;
        bool init(const EngineCfg &cfg);
        bool fail(const char *what, hipError_t e);
        bool alloc_stage(Stage &st);
        void free_stage(Stage &st);
        bool global_reset();
        bool fence_background(bool keep_plan = false);
        bool fence_chains(bool keep_forward = false);
        bool join_forward_stream();
        bool input_behind_forward();
        // Host-pointer calls of one whole hop on a streamed engine (c4, ns64, c5): the part of the hop's multiply-accumulate that needs
        // nothing of the new block — partitions >= 1, over spectra the ring holds already: all but 1 / (P + 1) of the launch — is enqueued
        // BEFORE the caller's samples are staged and uploaded, so that the staging copy and the PCIe transfer run beside it instead of in
        // front of it (hcv_engine_block.hip: host_pre_mac).  What it left for the block's own enqueue:
        struct PreMac { bool valid = false; uint64_t block = 0; int ksplit = 0; long long h_mac = 0; uint32_t nin = 0, nout = 0; };
        PreMac mPre;
        bool host_pre_mac(uint32_t nin_act, uint32_t nout_act, uint32_t B);
        bool ensure_staging(Stage &st, uint32_t parts);
        // control-path device memory: stream-ordered allocation on the control stream (hipMallocAsync / hipFreeAsync).  The
        // synchronous calls take runtime-wide locks and, for hipFree, wait for the whole device: an audio thread's launches
        // stood behind them for up to 19 ms while a control thread regrew a stage.  (ctl_alloc'ed memory: ctl_free only.)
        hipError_t ctl_alloc(void **p, size_t bytes);
        void ctl_free(void *p);
        // Ownership of the engine's host state and enqueue order (the streams' program order, the rings' bookkeeping).  kOwnerAudio: a
        // process call between audio_enter and audio_leave; kOwnerControl: a control thread inside a section, taken only when no process
        // call was made for kStreamingWindowNs (and given up again at once if one turns out to have started).  The audio thread tries ONCE.
        enum : uint32_t { kOwnerFree = 0, kOwnerAudio = 1, kOwnerControl = 2 };
        std::atomic<uint32_t> mOwner { kOwnerFree };
        // The control mailbox.  A control call's swap section (set_ir phase B, the pointer swap of a regrow, a fence for synchronize) must
        // run between two blocks.  While a stream is running (a process call within the last kStreamingWindowNs) the control thread
        // POSTS the section and waits for it (control threads may wait); the audio thread runs it at the start — or the end — of its
        // next call, inside the ownership it holds anyway (the reference mutes the pair meanwhile, here the pair plays its previous
        // IR until the swap).  With no stream running the control thread takes the ownership itself.  Control calls are serialised by
        // mSetMutex and readers of the statistics by mQueryMutex, each posting one section at a time: two slots.
        struct CtlJob
        {
            std::function<bool()> fn;
            std::atomic<bool> done { false };
            bool ok = false;
        };
        bool run_exclusive(std::function<bool()> fn, int slot = 0);        // control threads
        bool audio_enter(uint64_t samples);                 // audio thread: stamp, take the ownership (false: a stream-start collision), run posted sections
        void audio_leave();                                 // audio thread: posted sections once more, stamp, ownership back
        void run_mailbox();
        // the ownership of a process call, given back on every way out of it (an error return included)
        struct OwnerGuard
        {
            Engine *e;
            explicit OwnerGuard(Engine *eng) : e(eng) {}
            void leave() { if (e) { e->audio_leave(); e = nullptr; } }
            ~OwnerGuard() { leave(); }
            OwnerGuard(const OwnerGuard &) = delete;
            OwnerGuard &operator=(const OwnerGuard &) = delete;
        };
        std::atomic<CtlJob *> mMailbox[2] = { { nullptr }, { nullptr } };
        std::atomic<long long> mLastAudioNs { 0 };
        std::atomic<size_t> mAudioThread { 0 };             // (hash of) the thread that made the last process call
        std::atomic<uint64_t> mMailboxRuns { 0 };           // sections the audio thread ran for control threads
        std::atomic<uint64_t> mMailboxNsMax { 0 }, mMailboxNsTotal { 0 };   // ... and what they took of its calls
        std::atomic<uint64_t> mCtlSections { 0 };           // sections control threads ran themselves (no stream running)
        std::atomic<uint64_t> mStartCollisions { 0 };
        std::atomic<uint64_t> mStartWaits { 0 };            // calls of kOfflineCallSamples or more that waited a control section out instead (audio_enter)
        bool apply_pending_resets();
        std::atomic<uint32_t> mResetAllGen { 0 };           // reset_all() calls so far
        uint32_t mResetAllSeen = 0;                         // ... applied so far (audio side)
        std::vector<size_t> mRestartScratch;                // ... and the pairs that restart (likewise)
        std::vector<uint8_t> mTaken;                        // apply_pending_resets: the flags taken this block (no allocation on the audio thread)
        bool update_active_matrix(uint32_t rows_in, uint32_t nout_act);
        struct Block;
        bool enqueue_chunk(const float *din, int64_t in_stride, float *dout, int64_t out_stride, uint32_t nin_act, uint32_t nout_act, uint32_t B);
        bool enqueue_stage(Block &blk, size_t si, size_t sj);
        static MacShape mac_shape(const Stage &st, int P, int Pcap, int nin, int nin_alloc, int nout, int diag, int T, int max_ksplit);
        bool advance_background(const Block &blk, Stage &st, bool boundary);
        bool catch_up_stage(const Block &blk, Stage &st, long long h_first, bool rebuild_spectra);
        size_t pair_index(uint32_t in, uint32_t out) const { return (size_t) out * mNinAlloc + (mCfg.diag ? 0 : in); }
        void collect_events();
        void collect_events_owned();
        // exact per-pair restart (hcv_ghost.hip): ghost spectra of the input before the restart, the pair's pending output retired
        struct GhostEvent;
        bool mac(Stage &st, const MacShape &s, const MacPlan &pl, const float2 *H, float2 *Y, long long h_first, bool check, hipStream_t stream);
        bool retire_pair(size_t pair);
        bool make_ghost_event(const std::vector<size_t> &pairs);
        bool rebuild_ghost_tables();
        void release_ghost(size_t pair);
        void drop_ghosts();
        bool prune_ghosts();
        void *ghost_alloc(size_t bytes);
        void ghost_free(void *p, size_t bytes);

        EngineCfg mCfg;
        int mDevice = 0;
        uint32_t mMaxBlock = 0, mNinAlloc = 1;
        std::vector<Stage *> mStages;
        std::mutex mSetMutex;       // serialises the control calls (shared IR staging buffer, one mailbox slot)
        std::mutex mQueryMutex;     // serialises the readers of statistics / synchronize (the other mailbox slot)
        std::string mErr;

        // mStream: control + emit (and the D2H copy of the host path); mInStream: input scatter (and the H2D copy);
        // one stream per FFT stage; mTdStream: FIR head.  Events are double-buffered by block parity so block k+1 can
        // start while block k is still draining (see enqueue_chunk).
        hipStream_t mStream = nullptr, mInStream = nullptr, mTdStream = nullptr;
        hipEvent_t mEvInput[2] = { nullptr, nullptr }, mEvTd[2] = { nullptr, nullptr }, mEvEmit[2] = { nullptr, nullptr };
        hipEvent_t mEvCtl = nullptr;
        hipStream_t mCtlStream = nullptr;   // control work beside the audio streams: IR upload + FFTs into staging, capacity growth
        hipEvent_t mEvSwapDone = nullptr;   // the last swap's device-to-device copies (main stream) have read the staging buffers
        hipEvent_t mEvSnap = nullptr;       // capacity growth: "everything enqueued so far"
        hipEvent_t mEvHostDone = nullptr;   // host path: the block's download has landed in the pinned buffer
        bool mHostMuted = false;            // host path: the block between process_begin and process_end was given up
        float *mStageTaps = nullptr;        // staging of set_ir: head taps, head spectrum, tail-head spectrum
        float2 *mStageHead = nullptr, *mStageTailHead = nullptr;
        hipEvent_t mEvSerial = nullptr;     // end of a run of serial blocks (see enqueue_chunk)
        bool mPrevSerial = false;           // the previous block ran serially on the main stream
        // small engines, whole-hop blocks: the forward transforms of block k+1 run on a second stream beside block k's
        // multiply-accumulate and inverse (the block's time is a chain of latency-bound launches; asynchronous callers overlap them)
        hipStream_t mPipeStream = nullptr;
        hipEvent_t mEvPipe[2] = { nullptr, nullptr };
        hipEvent_t mEvPipeEnd[4] = { nullptr, nullptr, nullptr, nullptr };   // ends of the last four pipelined blocks
        uint64_t mPipeSeq = 0;              // pipelined blocks enqueued so far
        uint32_t mPipeRun = 0;              // consecutive pipelined single-hop blocks before the one being enqueued
        int64_t mPipeRecA = -1, mPipeRecB = -1;   // the last two pipelined blocks that recorded an end event (sequence numbers)
        uint32_t mPipeSince = 0;            // pipelined blocks since the pipe stream was last lined up behind the main stream
        bool mPrevPipe2 = false;
        // n x m fused blocks (hcv_fused_nxm.hip): the forward transforms of such a block run on the pipe stream with NO event between the two
        // streams — they meet inside the multiply-accumulate launch.  Whatever else touches the rings (a block of another kind, control
        // work, synchronize) first puts the main stream behind them by an event (join_forward_stream, from fence_chains)
        bool mFwdPending = false;           // forward launches on the pipe stream the main stream has not been put behind yet
        bool mPrevNxm = false;              // the previous block was such a block
        uint64_t mNxmRun = 0;               // such blocks since the pipe stream was last lined up behind the main stream
        uint32_t mNxmEvery = 3;             // ... every how many of them record their end (from the rings' depth, enqueue_chunk)
        hipEvent_t mEvNxmEnd[4] = { nullptr, nullptr, nullptr, nullptr };    // ends of every mNxmEvery-th such block (back-pressure on the pipe stream, enqueue_chunk)
        hipEvent_t mEvFwd = nullptr;
        hipEvent_t mEmitFirstEv = nullptr;                  // the last block's emit event when it was enqueued in front of late chains
        hipEvent_t mHostWait = nullptr;                     // what process_end waits for
        bool mCallWaits = false;            // the call being enqueued waits for its result (host pointers, sync = true): nothing to pipeline
        long long mArenaWant = 0;           // bytes this engine asked the device's control arena to hold (init; taken back by the destructor)
        std::atomic<uint64_t> mArenaMisses { 0 };   // control-path allocations the arena could not serve (stream-ordered pool: may stall the streams)
        std::vector<void *> mParked;        // buffers replaced by a regrow whose hipFree would stall the device: freed with the engine
        bool mPrevDirect = false;           // the previous block's history was written by its last stage's forward FFTs (direct input)
        bool mPrevPlain = false;            // the previous block was a plain small call: its samples were filed in the ring from the MAIN stream
        bool mCtlDirty = false;             // control work (IR loads, resets, regrow) was queued on mStream since the last block
        bool mExtDirty = false;             // the main stream was made to wait for a foreign event (process_dev `after`): a streamed block fans it out
        uint64_t mBlockCount = 0;
        OrderCheck *mOrd = nullptr;
        int mStreamsSpread = -2;        // hcv_queue_probe.hip's verdict at creation: streams replaced, -1 = not run, -2 = not needed
        int mPinXcd = 0;                    // the XCD this engine's pinned tiny launches go to (hcv_kernels.h: xcd_pin_for); engines are spread over the eight

        // rings and staging
        float *mHist = nullptr;     long long mHistLen = 0;
        float *mTdOut[2] = { nullptr, nullptr };   // FIR output, double-buffered by block parity
        float *mDevIn = nullptr, *mDevOut = nullptr;
        float *mPinIn = nullptr, *mPinOut = nullptr;
        float *mPinInDev = nullptr, *mPinOutDev = nullptr;      // device mappings of the pinned buffers (zero-copy small blocks)
        float *mIrBuf = nullptr;    uint64_t mIrCap = 0;

        // time-domain head
        float2 *mHeadSpec = nullptr;        // [nout][nin_alloc][M0] head taps as ONE zero-latency partition of the first FFT stage
        float2 *mHeadYq[2] = { nullptr, nullptr };   // [Tmax0][nout][M0], by block parity
        bool mHeadFFT = false;              // the head may take the FFT path (taps fit one hop of the first stage)
        // Whole-hop mode: for calls made of whole, aligned hops of the LAST stage everything in front of that stage's segment
        // (head + shorter stages) is one extra zero-latency partition of it
        // (the spectrum of IR[0 : Mlast) lives in the lead slot of the last stage's spectra, Stage::lead)
        bool mTailHead = false;             // the layout allows whole-hop blocks (contiguous zero-latency ladder, or a lone FFT stage)
        bool mLeadSlot = false;             // ... of the first kind: the pivot stage carries the lead slot (Stage::lead)
        size_t mPivot = 0;                  // the stage whole-hop blocks run on: the last one, or the last in front of the extended ladder's rungs
        bool mTailHeadPrev = false;         // the previous block ran in whole-hop mode
        float *mTaps = nullptr;
        long long *mTdValid = nullptr;
        std::vector<uint32_t> mTdCount;     // taps per pair
        uint32_t mTdLpad = 0;
        long long mTdMaxValid = 0;

        // per pair
        std::vector<uint8_t> mPending, mLoaded;
        std::vector<uint8_t> mRetired;      // the pair's pending output has already been taken out of the timelines (by set_ir)
        std::vector<GhostEvent *> mGhostOf; // the restart whose ghost spectra the pair still needs (nullptr: none)
        std::vector<GhostEvent *> mGhostEvents;
        std::vector<std::pair<size_t, void *>> mGhostPool;      // released ghost-spectrum blocks (bytes, pointer), reused by size
        float *mGhostHist = nullptr;        // [nin][mGhostLen] masked copy of the input history around a restart
        long long mGhostLen = 0;
        float *mRetireTmp = nullptr;        // one inverse-transformed frame of the largest stage
        char *mGhostPin = nullptr;          // pinned staging of the device tables below
        size_t mGhostPinBytes = 0;
        hipEvent_t mGhostUploaded = nullptr;
        long long mGhostPruneAt = -1;       // sample count at which the oldest restart stops mattering (-1: none)
        uint32_t mLastNin = 0, mLastNout = 0;   // active matrix of the previous block
        long long mN = 0;                   // samples since the last global reset
        std::atomic<bool> mProfiling { false };
        std::atomic<bool> mDropHead { false };
        bool mOneStream = false;            // every kernel on mStream (small engines: dependency hops cost more than overlap gains)
        std::vector<EventPair *> mEvents;
    };

    const float2 *twiddles(int device, int log2n, std::string *err);   // cached per (device, size)
    // the control arena of a device (hcv_engine.hip): what it is to hold — before the device's first engine — and what it holds
    bool ctl_arena_reserve(int device, size_t bytes);
    size_t ctl_arena_size(int device);
    // who counts as "the audio thread" for the calling thread's process calls (hcv_engine.hip): a shard's enqueue thread takes the identity
    // of the user thread that posted the block
    void set_thread_audio_identity(size_t id);
    size_t current_thread_identity();
}
