// Private to the engine's translation units (hcv_engine.hip: set-up and control; hcv_engine_restart.hip: exact per-pair
// restart; hcv_engine_block.hip: the per-block scheduler): the structures behind Engine's opaque members.
#pragma once

#include "hcv_engine.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "hcv_order_check.h"         // (last: it routes the event / wait / synchronize calls of the engine's translation units through its hooks)

namespace hcv
{

// hcv_queue_probe.hip: fresh streams swapped until they feed different hardware queues (roles[share] the anchor's own)
int spread_streams(hipStream_t anchor, hipStream_t **roles, int n, int share);
void note_device_streaming(int device, long long now_ns);      // (Engine::audio_enter: no experiment beside a running stream)

// hcv_engine.hip: the engines' streams come from a per-device pool and go back to it, idle, when their engine goes — never destroyed.  A
// stream is a queue handle: an idle one carries nothing over.  Creating and destroying a dozen streams per engine was where two full-suite
// runs and two knob-matrix runs died (glibc abort on a free inside hipStreamDestroy, no heap error in the instrumented host code under
// AddressSanitizer: profiles/r05_stream_pool.txt), and a pooled stream keeps the hardware queue it was given.
hipError_t stream_take(int device, hipStream_t *s);
void stream_give(int device, hipStream_t s);

#define HCV_TRY(expr)                                                                                                  \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) return fail(#expr, e_);                                                                  \
    } while (0)

// HCV_ROCTX=1 (the reference has no tracing; rocprofv3 --marker-trace shows these): ranges `hcv:block` around every block's
// enqueue and `hcv:set` / `hcv:resize` around the control calls.  The marker library (rocprofiler-sdk's, else roctracer's) is opened at run time
// on first use — the product has no link dependency on it — and with the variable unset a range costs one load of a static flag.
struct RoctxApi
{
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
};
const RoctxApi *roctx_api();                 // hcv_engine.hip; nullptr: off, or no marker library to be had
struct RoctxRange
{
    const RoctxApi *api;
    explicit RoctxRange(const char *name) : api(roctx_api()) { if (api) (void) api->push(name); }
    ~RoctxRange() { if (api) (void) api->pop(); }
    RoctxRange(const RoctxRange &) = delete;
    RoctxRange &operator=(const RoctxRange &) = delete;
};

inline long long pow2ceil(long long v)
{
    long long p = 1;
    while (p < v) p <<= 1;
    return p;
}

// the calling thread's current HIP device is put back when an entry point returns (the engine may live on another GPU than
// the one the caller — PyTorch, say — is working on)
struct DeviceGuard
{
    int prev = -1;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) (void) hipSetDevice(device);
        else prev = -1;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void) hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

constexpr int kBgSlices = 16;
constexpr int kFoldMax = 8;              // the inverse adds up to this many split-K slices itself instead of a reduce_partials launch
constexpr int kBoundarySlices = 7;       // k-slices of the partition-0 MAC at a deferred hop's boundary, in the slots behind the background slices'

// HCV_EXACT_RESTART=0 falls back to the hop-granular fence alone (the restarted pair may see up to two hops of older input
// per stage and its pending output is not withdrawn) — for A/B comparison only
inline bool exact_restart()
{
    static const bool on = !(std::getenv("HCV_EXACT_RESTART") && std::atoi(std::getenv("HCV_EXACT_RESTART")) == 0);
    return on;
}


struct Engine::Stage
{
    StageCfg cfg;
    int log2n = 0;
    uint32_t N = 0, M = 0;
    uint32_t Pcap = 0, P = 0, R = 0, Tmax = 0, ring_extra = 0;
    // Hs = [pairs][lead + Pcap][M].  lead = 1 on the last stage of a whole-hop capable layout: slot 0 of every pair then holds
    // the spectrum of IR[0 : M) — everything in front of the stage's own segment — so that in whole-hop mode the stage runs ONE
    // zero-latency uniform convolution over lead + P partitions (Y(h) = sum_p' X[h - p'] H'[p']); the stage's own partitions
    // follow at Ht()
    uint32_t lead = 0;
    float2 *Hs = nullptr, *X = nullptr;
    size_t hstride() const { return (size_t) (Pcap + lead) * M; }       // float2 between two pairs' spectra
    int hparts() const { return (int) (Pcap + lead); }                  // the same in partitions (MacShape::Pcap is this stride)
    float2 *Ht() const { return Hs + (size_t) lead * M; }               // the stage's own partitions of pair 0
    float2 *Y = nullptr;                // scratch of the current block = Yq[block parity]
    float2 *Yq[2] = { nullptr, nullptr };   // split-K partials, double-buffered so MAC(k+1) can run while block k is inverted
    size_t y_elems = 0;
    // Deferred ("time-spread") mode, the GPU form of the reference's partition scheduler (PartitionedConvolve.cpp:321-348):
    // partitions 1..P-1 of hop h+1 only need spectra up to hop h, so they are accumulated BETWEEN the boundaries of hop h
    // and hop h+1, in up to kBgSlices short launches spread over the calls of that hop in step with the samples that
    // have arrived (a single long launch would sit in a hardware queue that other streams share and stall them for
    // milliseconds); the boundary of hop h+1 then only pays partition 0 + the inverse FFT.
    float2 *Ypre = nullptr;             // [kBgSlices + kBoundarySlices][nout][M]: one partial sum per background slice, then the boundary's
                                        // partition-0 slices; the inverse adds them up (HCV_BOUNDARY_KSPLIT = 1: slot 0 receives the total)
    long long pre_hop = -1;             // hop index the slices accumulate for (-1 = no plan)
    int bg_parts = 0;                   // partitions 1..bg_parts of that hop are to be accumulated
    int bg_slices = 0, bg_launched = 0; // planned / already launched slices
    hipEvent_t bg_done = nullptr;       // recorded after every background launch
    bool bg_pending = false;
    int chain_pending = -1;             // parity of the `done` event of a boundary chain the main stream has not waited for yet (fence_chains)
    float *timeline = nullptr;          // [nout][tl_len] this stage's hop results at their emission times
    long long tl_len = 0;
    BigFFTWork big;                     // scratch of the four-step FFT (only for N > 32768)
    BigFFTWork big_ctl;                 // the same for IR transforms on the control stream
    bool hs_ctl = false;                // Hs / X came from ctl_alloc (a regrow) rather than from init's hipMalloc
    float2 *stage_spec = nullptr;       // staging for one pair's spectra (set_ir phase A), stage_parts partitions
    uint32_t stage_parts = 0;
    hipStream_t stream = nullptr;       // stages are independent until emit(): each runs on its own stream (the MAC stream)
    hipStream_t stream2 = nullptr;      // the pivot stage of an extended ladder: whole-hop blocks of odd parity run their chain here (two lanes,
                                        // enqueue_stage), so that consecutive blocks' latency-bound chains overlap
    hipEvent_t mac_done[2] = { nullptr, nullptr };     // the stage's spectral_mac of a block has finished (tail gate)
    hipEvent_t done[2] = { nullptr, nullptr };   // by block parity
    long long *hv = nullptr;
    long long max_hv = 0;
    bool hv_zero = true;                // every pair's first hop is 0 (nothing but global resets since the stage was made): the ramp-up's bound is uniform
    unsigned *coop_bar = nullptr;       // fused blocks: the two monotonic hand-over counters (one-output engines only)
    unsigned long long *coop_flags = nullptr;   // fused blocks: per-task completion marks (hcv_kernels.h: kFusedMacTasks + kFusedFwdTasks), one-output engines only
    unsigned coop_arrived[2] = { 0, 0 };    // fused blocks: what the two hand-over counters read once everything launched so far has arrived
    unsigned *nxm_helped = nullptr, *nxm_helped_dev = nullptr;     // host-mapped: n x m launches that had to do their forward transforms themselves
    unsigned nxm_helped_seen = 0, nxm_strikes = 0;
    uint64_t nxm_backoff = 64;          // blocks the next stand-down lasts (enqueue_chunk)
    uint64_t nxm_strike_block = 0, nxm_off_until = 0, nxm_stood_down = 0;               // (engine block counts) the last strike; separate kernels until this block
    unsigned coop_arrived_nxm[kFusedShards] = {};       // ... of the n x m block's sharded counter (hcv_fused_sync.h), per shard
    unsigned long long coop_seq = 0;        // fused blocks launched so far
    bool coop_off = false;
             // a fused launch was refused by the runtime: this stage takes the separate kernels from then on
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
    uint32_t last_ksplit = 0, last_ot = 0, last_tt = 0, last_parts = 0;
    uint64_t steady_launches = 0, fused_launches = 0, host_pre_launches = 0;
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

// Everything one block's enqueue needs to know, decided once on the host (enqueue_chunk) and shared by the helpers below.
struct Engine::Block
{
    const float *din = nullptr;
    float *dout = nullptr;
    int64_t in_stride = 0, out_stride = 0;
    uint32_t nin_act = 0, nout_act = 0, rows_in = 0, B = 0;
    long long n0 = 0, hmask = 0;
    int q = 0;                          // block parity: every event and double buffer is indexed by it
    size_t last = 0;                    // index of the stage whole-hop blocks run on (Engine::mPivot): the last one, or the one in front of
                                        // the extended ladder's rungs
    bool td_any = false, td_check = false, whole_hops = false, entering = false, leaving = false, head_fft = false, td = false;
    bool serial = false, full_matrix = false;
    bool pipe_far = false;              // ... in a run of single-hop blocks (its transforms wait for the block three or four back)
    bool pipe2 = false;                 // serial whole-hop block with its forward transforms on the pipe stream
    bool nxm = false;                   // serial whole-hop block of a matrix with several outputs as the two meeting launches of hcv_fused_nxm.hip
    FusedNxmPlan nxm_plan = {};
    bool direct_out = false;            // whole-hop block: the inverse writes the caller's block itself, no timeline, no emit launch
    bool direct_in = false;             // the (only) running stage's forward FFTs read the caller's block themselves: no scatter launch
    bool emitted = false;               // a plain small call: the head kernel has delivered the block itself (no emit launch)
    uint32_t late_mask = 0;             // bit 2 * stage + parity: boundary chains of this block that run on past its emit
    bool emit_first = false;            // a small streamed block: its emit is enqueued IN FRONT of the late chains (enqueue_chunk); the timelines are
                                        // registered and the stages' streams are behind emit(k-2) already
    int tail_gate = 0;
    hipEvent_t gate = nullptr;          // the tail's spectral_mac of this block has finished (tail gate)
    hipStream_t main = nullptr, sIn = nullptr, sTd = nullptr;
    EmitSources src;

    // serial blocks run on the main stream alone, in program order: no event is recorded or waited for
    hipError_t rec(hipEvent_t e, hipStream_t s) const { return serial ? hipSuccess : hipEventRecord(e, s); }
    hipError_t wt(hipStream_t s, hipEvent_t e) const { return serial ? hipSuccess : hipStreamWaitEvent(s, e, 0); }
    hipStream_t stage_stream(hipStream_t own) const { return serial ? main : own; }
};

// the block's constants under the names the code below uses
#define HCV_BLOCK_LOCALS(b)                                                                                                                        \
    const long long n0 = (b).n0, hmask = (b).hmask;                                                                                                \
    const uint32_t nin_act = (b).nin_act, nout_act = (b).nout_act, rows_in = (b).rows_in, B = (b).B;                                                \
    const int q = (b).q;                                                                                                                           \
    const size_t last = (b).last;                                                                                                                  \
    const bool whole_hops = (b).whole_hops, entering = (b).entering, leaving = (b).leaving, head_fft = (b).head_fft, serial = (b).serial,          \
               full_matrix = (b).full_matrix;                                                                                                      \
    const int tail_gate = (b).tail_gate;                                                                                                           \
    const hipStream_t sTd = (b).sTd;                                                                                                               \
    const bool direct_in = (b).direct_in;                                                                                                          \
    auto rec = [&](hipEvent_t e_, hipStream_t s_) { return (b).rec(e_, s_); };                                                                     \
    auto wt = [&](hipStream_t s_, hipEvent_t e_) { return (b).wt(s_, e_); };                                                                       \
    (void) n0; (void) hmask; (void) nin_act; (void) nout_act; (void) rows_in; (void) B; (void) q; (void) last; (void) whole_hops; (void) entering; \
    (void) leaving; (void) head_fft; (void) serial; (void) full_matrix; (void) tail_gate; (void) sTd; (void) rec; (void) wt; (void) direct_in

} // namespace hcv
