// The block of an n x m matrix in three launches on two streams, two of which meet INSIDE a launch: one whole-hop block of an engine
// with SEVERAL outputs — Convolver::process for one hop of its last stage (Convolver.cpp:138-154 -> NToMonoConvolve.cpp:35-43 ->
// PartitionedConvolve.cpp:243-385) —
//
//   forward stream:  fwd_publish_kernel     input i's new frame -> X[i][h], the hop filed in the history ring      (nin workgroups of 512)
//   main stream:     mac_meet_kernel        Y = sum_{i, p >= 1} X[i][h - p] H[o][i][p]           (every workgroup: its eight k-slices)
//                                           ... wait for the forward transforms' arrival counters ...
//                                           Y += sum_i X[i][h] H[o][i][0]                        (the lead slot: the nin new terms)
//                                           -> the k-slices' sums meet in LDS -> `ms` partial spectra per output
//                    rifft_split_emit       the inverse transforms, summing the `ms` partials as they stage them (hcv_fft_split.hip)
//
// in place of four launches that meet through events.  What it is for: one rank's share of config 4 strong-scaled over 8 GPUs — 64 inputs
// x 8 output rows, 2 s IRs, the shape row (e) of SURVEY section 8 turns on.  There the separate launches were: transforms on the pipe
// stream with an event each way (~10 us of the main stream's time) -> spectral_mac with 48 k-slices through memory (25 MB written, 25 MB
// read back) -> reduce_partials -> inverse: 0.106 ms per 8192-sample block around a 71 us multiply-accumulate.  Here
//   * the forward transforms of block n run on their own stream with NO event in either direction: they are launched first, need
//     nothing of the main stream (only their own predecessor, by stream order), and the multiply-accumulate needs them for its last
//     twelfth only (the lead slot) — by then their arrival counters, agent-scope and monotonic over the launches, are long reached;
//   * the k-slices are the 8 waves of a 512-thread workgroup (64 lanes = 128 bins each) whose sums meet in LDS: 4 partial spectra per
//     output go through memory instead of 48, and the inverse adds them up as it stages them — no reduce launch.
// The inverse stays a launch of its own: as a third phase of the multiply-accumulate launch (built, measured) its workgroups waited 12 us
// for the last arrival to become visible — 64 workgroups polling the counters the 256 arrivals land on, all at the memory side — and then
// staged the partial spectra through agent-scope loads (another 7 us): 96 us per launch against 70 + a kernel boundary + 9.
//
// Footprints on purpose.  The multiply-accumulate kernel takes ONE workgroup per CU whose eight waves hold at most 2 x 168 of a SIMD's
// 512 registers; the forward kernel's workgroup (512 threads, 80 registers, 70 KiB of LDS beside this kernel's 70) fits beside it on
// every CU.  A first version filled the register file (sixteen waves of 128): the forward launch of the SAME block, started a few
// microseconds behind it by a stream wait, found no CU to run on until the multiply-accumulate's waits had run out — 1.5 ms per block.
//
// Forward progress is by construction, as in the one-output blocks (hcv_fused_sync.h): the wait is bounded and a workgroup whose wait runs
// out does the missing forward transforms itself — the SAME body, bit for bit, so whoever runs a task writes the same values.  A forward
// launch the device schedules late (another engine's work in front of it on the hardware queue) costs time, never correctness: the
// main stream's workgroups compute the spectra themselves, the late launch writes them once more.  The engine joins the forward stream
// into the main stream by an event before anything else touches the rings (a block of another kind, control work: Engine::fence_chains).
//
// Everything the forward kernel writes is WRITTEN THROUGH at agent scope (spectra and the ring's new hop alike): its consumers run in
// another launch that started before it ended, and read the new spectra with agent-scope loads.  It is the engine's whole-frame
// transform (one workgroup per input: a wave's stores are 512 contiguous bytes, whole lines).  The residue-split transform of the
// one-output blocks was tried first — 9 workgroups per input, but a residue class's bins are 128 bytes apart: its write-through stores
// were half a million partial-line writes per block, each a read-modify-write at the memory side (50 - 90 us per launch, and the
// multiply-accumulate beside it starved); with plain stores and an agent-scope release per workgroup instead, 80 - 150 us.

#include "hcv_engine.h"
#include "hcv_fft_frames_device.h"
#include "hcv_order_check.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>

namespace hcv
{

struct FusedNxmParams
{
    float *hist;
    const float *in;
    float2 *X;                  // [nin][Rring][M]
    const float2 *H;            // [nout][nin_alloc][hparts][M], lead slot first
    float2 *Y;                  // [ms][nout][M]
    const float2 *tw;
    FusedSyncSharded sy;
    long long hist_stride, in_stride, hist_mask, n0, h;
    long long pair_stride4, out_stride4;        // float4 between two inputs' / two outputs' spectra
    int Rring, P, slot, nin, nout;              // P = live partitions, lead slot included; slot = h mod Rring
    int ms, tiles, kper_old;                    // k-slice groups (partial spectra per output), output tiles of 8, (i, p >= 1) terms per wave
    unsigned long long *hint;                   // [kNxmHints] marks 128 bytes apart: "inverse workgroup m of launch `seq` has started"
    int hint_wait;                              // > 0: the forward launch holds itself back until the PREVIOUS block's inverse has started (marks to look at)
    unsigned long long *progress;               // the sequence number of the newest multiply-accumulate launch whose first workgroup is through (see
                                                // fwd_publish_kernel: a forward launch that arrives four blocks late keeps its hands off the rings)
    unsigned *helped;                           // host memory (mapped): launches whose wait for the forward transforms ran out — the engine's cue to
                                                // take the separate kernels for a while (enqueue_stage)
};

static_assert(kShards == kFusedShards && kShardStride == kFusedShardStride, "hcv_kernels.h sizes the engine's counters");

namespace
{
    constexpr int kNxmLog2N = 14, kNxmOT = 8, kNxmWaves = 8;
    constexpr int kNxmHints = 64, kNxmHintStride = 16, kNxmHintBase = 8192;     // (inside the stage's flag array, behind the forward tasks' marks)
    constexpr int kNxmProgress = kNxmHintBase + (kNxmHints + 1) * kNxmHintStride;      // (a line of its own behind the hint marks)
    static_assert(kNxmProgress + kNxmHintStride <= kFusedFwdTasks, "");

    // The forward transform of input i's new frame, whole, by one thread group of 512: the engine's own whole-frame transform
    // (hcv_kernels.hip: rfft_frames_direct_kernel — LDS Stockham, the new hop read from the caller's block and filed in the history ring
    // by the first pass, the real post-pass writing the packed spectrum), with every store written through
    template <int LOG2N> __device__ __forceinline__ void forward_frame(const FusedNxmParams &a, float2 *lds, int tid, int i)
    {
        constexpr int LOG2M = LOG2N - 1, M = 1 << LOG2M, TG = FFTGeom<LOG2M>::TG;
        static_assert(TG == 64 * kNxmWaves, "the helping path runs this body on the multiply-accumulate kernel's threads");
        const LdsBuf<float2> s = { lds };
        const DirectFrameLoadT<true> ld = { a.hist + (long long) i * a.hist_stride, a.in + (long long) i * a.in_stride, (a.h - 1) * (long long) M, a.hist_mask, a.n0, M / 2, true };
        LdsFFT<LOG2M, TG>::run(ld, LdsIO<float2>{ s }, s, tid, a.tw);
        real_post_store<LOG2M, TG, true>(s, tid, a.tw, a.X + ((long long) i * a.Rring + a.slot) * M);
    }

    // (test aid, HCV_NXM_TEST_DELAY_US: holds the forward stream back in front of every forward launch — the stream stuck behind other
    // engines' packets, made to order)
    __global__ void nxm_delay_kernel(unsigned long long ticks)
    {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    }

    // One workgroup of the forward kernel = input i (task i, the helping path's numbering)
    template <int LOG2N>
    __global__ __launch_bounds__(64 * kNxmWaves) __attribute__((amdgpu_waves_per_eu(6, 6))) void fwd_publish_kernel(FusedNxmParams a)
    {
        extern __shared__ __attribute__((aligned(16))) float2 dynf[];
        const int task = (int) blockIdx.x, tid = (int) threadIdx.x;
        if (a.hint_wait)
        {
            // A scheduling hint, nothing more.  Nothing orders this launch against the main stream, so in a GPU-bound stream it ran as soon
            // as its own predecessor was through — BESIDE the multiply-accumulate of an earlier block, for 45 - 60 us instead of 16, and that
            // launch 8 - 10 us longer for it.  The place for it is beside the previous block's INVERSE, when three CUs in four are idle: each
            // workgroup sleeps until a workgroup of the previous block's inverse launch says it has started (one mark per workgroup, its
            // own line; ~0.6 us between looks), for 0.25 ms at most.  A stream with gaps finds the mark set.  (Marks set by the
            // multiply-accumulate's workgroups as they END put this launch beside that launch's last 15 us: no gain.)
            if (tid == 0)
            {
                const unsigned long long *mark = a.hint + (task % a.hint_wait) * kNxmHintStride;
                for (int k = 0; k < 400; k++)
                {
                    if ((long long) (__hip_atomic_load(mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (a.sy.seq - 1)) >= 0) break;
                    __builtin_amdgcn_s_sleep(20);
                }
            }
            __syncthreads();
        }
        // Nothing makes the MAIN stream wait for this one: a multiply-accumulate launch whose wait runs out does the transforms itself and
        // goes on, so with this stream held up — behind other engines' packets in a shared hardware queue — the main stream can be blocks
        // ahead when this launch finally runs, and its transform would then file an OLD hop over a newer one in the history ring (eight
        // hops deep) and an old spectrum into a ring slot that has come round.  The main stream says how far it is (mac_meet_kernel's first
        // workgroup, as it ends); a workgroup that finds it four blocks past its own block — everything that could read this block's
        // spectrum as NEW is long through, the helpers have written every value this launch would write — only counts itself in.
        // (tests/test_fused_nxm_gpu.py::test_four_engines_at_once with every wait forced out: one run in some dozens gave garbage once the
        // forward stream no longer shared the main stream's queue.)
        // ... and a workgroup that arrives LESS late than that must not redo a task a helper has done either: the helper's
        // values stand, the multiply-accumulate launch may be through and the main stream past it — and with it the caller's block,
        // which the next call's upload (or the caller) is then free to overwrite: a transform of THAT would file the next block's
        // samples as this hop.  So: the task's own mark carries this launch's number (a helper did it), or the multiply-accumulate
        // launch of this block has started to end (`progress`, every task then done) — count in, write nothing.
        __shared__ int stale_b;
        if (tid == 0)
            stale_b = (long long) (__hip_atomic_load(a.progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.sy.seq) >= 0 ||
                      (long long) (__hip_atomic_load(a.sy.flagF + task, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.sy.seq) >= 0;
        __syncthreads();
        if (!stale_b) forward_frame<LOG2N>(a, dynf, tid, task);
        grid_publish_sharded(tid, a.sy.flagF + task, a.sy.seq, a.sy.bar, task);
    }

    template <int LOG2N> struct NxmMac
    {
        static constexpr int N = 1 << LOG2N, M = N / 2, M2 = M / 2, OT = kNxmOT, BINROWS = M2 / 64;
        const FusedNxmParams &a;
        float2 *dyn;
        int tid;
        float *nyq;                 // [8 waves][8 outputs] in LDS: bin 0's Nyquist products (see mac_old)
        float4 acc[OT];

        // task m = (k-slice group s, output tile, bin row): tiles of one (s, bin row) are 64 apart in m, so they land on ONE XCD and
        // meet in its L2 when they read the same input spectra
        __device__ __forceinline__ void decode(int m, int &bb, int &tile, int &s) const
        {
            bb = m % BINROWS;
            const int rest = m / BINROWS;
            tile = rest % a.tiles;
            s = rest / a.tiles;
        }
        static __device__ __forceinline__ void cmac(float4 &c, const float4 &x, const float4 &hv)
        {
            c.x += x.x * hv.x - x.y * hv.y;
            c.y += x.x * hv.y + x.y * hv.x;
            c.z += x.z * hv.z - x.w * hv.w;
            c.w += x.z * hv.w + x.w * hv.z;
        }
        // ---- the terms (i, p >= 1): everything they read is older than this block.  Wave kw of group s takes terms
        //      [g kper_old, (g + 1) kper_old) of the flattened (i, p - 1) axis, g = 8 s + kw; H streamed with nontemporal 16-byte loads
        __device__ __forceinline__ void mac_old(int m)
        {
            int bb, tile, s;
            decode(m, bb, tile, s);
            const int lane = tid & 63, kw = __builtin_amdgcn_readfirstlane(tid >> 6);
            const unsigned b4 = (unsigned) (bb * 64 + lane);
            const bool owns_bin0 = b4 == 0;
            const int o0 = tile * OT;
            // bin 0 carries (DC, Nyquist) and needs two real products instead of a complex one (PartitionedConvolve.cpp:398-406, 424-425):
            // the ONE lane of a bin row's waves that owns it keeps the sum of the Nyquist products x.y h.y beside the complex sums and
            // repairs its bin at the end.  Those eight sums live in LDS, added to by that lane alone (its own program order: no barrier,
            // nothing atomic about it) — as registers they were eight of every lane's budget and put the loop's operands into scratch.
            float *nq = nyq + kw * OT;
#pragma unroll
            for (int j = 0; j < OT; j++)
            {
                acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (owns_bin0) nq[j] = 0.f;
            }
            const int P1 = a.P - 1;
            const int K = a.nin * P1;
            const int g = s * kNxmWaves + kw;
            const int k0 = g * a.kper_old, k1 = min(K, k0 + a.kper_old);
            if (k0 >= k1) return;
            const float4 *X4 = reinterpret_cast<const float4 *>(a.X), *H4 = reinterpret_cast<const float4 *>(a.H);
            const int i_first = k0 / P1, i_last = (k1 - 1) / P1;
            for (int i = i_first; i <= i_last; i++)
            {
                const int pa = 1 + ((i == i_first) ? k0 - i_first * P1 : 0);
                const int pb = 1 + ((i == i_last) ? k1 - i_last * P1 : P1);
                const float4 *xrow = X4 + (long long) i * a.Rring * M2;
                // (dead outputs of a ragged last tile re-read the last live one; their sums are never stored)
                const float4 *hrow[OT];
#pragma unroll
                for (int j = 0; j < OT; j++) hrow[j] = H4 + (long long) min(o0 + j, a.nout - 1) * a.out_stride4 + (long long) i * a.pair_stride4;
                // (two partitions' eighteen loads in flight per wave: eight waves per CU must cover the memory latency alone)
#pragma unroll 2
                for (int p = pa; p < pb; p++)
                {
                    float4 hval[OT];
                    const unsigned hoff = (unsigned) p * (unsigned) M2 + b4;
#pragma unroll
                    for (int j = 0; j < OT; j++) hval[j] = load_nt(hrow[j] + hoff);
                    int sl = a.slot - p;
                    if (sl < 0) sl += a.Rring;
                    const float4 x = xrow[(unsigned) sl * (unsigned) M2 + b4];
#pragma unroll
                    for (int j = 0; j < OT; j++)
                    {
                        cmac(acc[j], x, hval[j]);
                        if (owns_bin0) nq[j] += x.y * hval[j].y;
                    }
                }
            }
        }
        // ---- the nin lead terms X[i][h] H[o][i][0] (input i by wave g, g + 8 ms, ...: the spectra were written a moment ago by
        //      another launch), the waves' sums added up in LDS in wave order, the group's partial spectrum stored
        __device__ __forceinline__ void mac_new(int m)
        {
            int bb, tile, s;
            decode(m, bb, tile, s);
            const int lane = tid & 63, kw = __builtin_amdgcn_readfirstlane(tid >> 6);
            const unsigned b4 = (unsigned) (bb * 64 + lane);
            const bool owns_bin0 = b4 == 0;
            const int o0 = tile * OT;
            const float4 *X4 = reinterpret_cast<const float4 *>(a.X), *H4 = reinterpret_cast<const float4 *>(a.H);
            const int g = s * kNxmWaves + kw;
            float *nq = nyq + kw * OT;
            for (int i = g; i < a.nin; i += kNxmWaves * a.ms)
            {
                const float2 *xp = reinterpret_cast<const float2 *>(X4 + ((long long) i * a.Rring + a.slot) * M2 + b4);
                const float2 lo = get2<true>(xp), hi = get2<true>(xp + 1);
                const float4 x = make_float4(lo.x, lo.y, hi.x, hi.y);
                float4 hval[OT];
#pragma unroll
                for (int j = 0; j < OT; j++) hval[j] = load_nt(H4 + (long long) min(o0 + j, a.nout - 1) * a.out_stride4 + (long long) i * a.pair_stride4 + b4);
#pragma unroll
                for (int j = 0; j < OT; j++)
                {
                    cmac(acc[j], x, hval[j]);
                    if (owns_bin0) nq[j] += x.y * hval[j].y;
                }
            }
            if (owns_bin0)
            {
#pragma unroll
                for (int j = 0; j < OT; j++)
                {
                    const float ny = nq[j];
                    acc[j].x += ny;                         // sum(x.x h.x - x.y h.y) + sum(x.y h.y) = the DC products; repaired per slice, the slices add up
                    acc[j].y = ny;                          // the Nyquist products
                }
            }
            float4 *red = reinterpret_cast<float4 *>(dyn);  // [8 waves][8 outputs][64 lanes] (free: whatever used the LDS before ended in a barrier)
#pragma unroll
            for (int j = 0; j < OT; j++) red[(kw * OT + j) * 64 + lane] = acc[j];
            __syncthreads();
            if (kw < OT && o0 + kw < a.nout)
            {
                // output o0 + kw by wave kw, the slices in wave order
                float4 sum = red[kw * 64 + lane];
#pragma unroll
                for (int k = 1; k < kNxmWaves; k++)
                {
                    const float4 q = red[(k * OT + kw) * 64 + lane];
                    sum.x += q.x; sum.y += q.y; sum.z += q.z; sum.w += q.w;
                }
                reinterpret_cast<float4 *>(a.Y + ((long long) s * a.nout + (o0 + kw)) * M)[b4] = sum;       // (read by the NEXT launch: a plain store)
            }
        }
    };

    // What a workgroup does once its wait for the forward transforms has run out: the forward tasks nobody has completed (helpers start
    // at different ones and skip what got done meanwhile), then its own multiply-accumulate from the start (nothing was kept).  Out of line,
    // entered as the last thing the kernel does (hcv_fused_sync.h: fused_slow_path has the reasons).
    template <int LOG2N> __device__ __noinline__ void nxm_slow(const FusedNxmParams *ka, float2 *dyn, int tid, int m, int nmac, int *slot_b, float *nyq)
    {
        const FusedSyncSharded &sy = ka->sy;
        const int nfwd = ka->nin;
        // (told to the host, once per launch: a forward launch that does not come — stuck behind another stream's packets in a hardware
        // queue the two share, there are four queues for a dozen streams — is a condition that lasts, and every block would pay the wait)
        if (m == 0 && tid == 0 && ka->helped) __hip_atomic_fetch_add(ka->helped, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const int first = (int) ((long long) m * nfwd / nmac);
        for (int k = next_undone(tid, sy.flagF, sy.seq, nfwd, first, 0, slot_b); k < nfwd; k = next_undone(tid, sy.flagF, sy.seq, nfwd, first, k + 1, slot_b))
        {
            int task = first + k;
            if (task >= nfwd) task -= nfwd;
            forward_frame<LOG2N>(*ka, dyn, tid, task);
            grid_publish_sharded(tid, sy.flagF + task, sy.seq, nullptr, task);
        }
        NxmMac<LOG2N> b = { *ka, dyn, tid, nyq };
        b.mac_old(m);
        b.mac_new(m);
        if (m == 0 && tid == 0) __hip_atomic_store(ka->progress, sy.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    template <int LOG2N>
    __global__ __launch_bounds__(64 * kNxmWaves) __attribute__((amdgpu_waves_per_eu(3, 3))) void mac_meet_kernel(FusedNxmParams a)
    {
        extern __shared__ __attribute__((aligned(16))) float2 dyn[];
        __shared__ int slot_b[2];
        __shared__ float nyq[kNxmWaves * kNxmOT];
        const int m = (int) blockIdx.x, tid = (int) threadIdx.x;
        NxmMac<LOG2N> b = { a, dyn, tid, nyq };
        b.mac_old(m);
        if (!grid_wait_bounded_sharded(tid, a.sy.bar, a.sy.targetA, a.sy.spin, slot_b))
        {
            nxm_slow<LOG2N>((const FusedNxmParams *) __builtin_amdgcn_kernarg_segment_ptr(), dyn, tid, m, (int) gridDim.x, slot_b, nyq);
            return;
        }
        b.mac_new(m);
        if (m == 0 && tid == 0) __hip_atomic_store(a.progress, a.sy.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (how far the main stream is: fwd_publish_kernel)
    }

    bool nxm_enabled()
    {
        static const bool on = !(std::getenv("HCV_COOP") && std::atoi(std::getenv("HCV_COOP")) == 0);
        return on;
    }
}

// The launch plan: `ms` groups of 8 k-slices so that about one workgroup per CU streams the spectra, every wave keeping
// at least six terms, the partial spectra within the room Y has and few enough for the inverse to add up itself
bool fused_block_nxm_plan(int log2n, int nin, int nout, int P, size_t y_elems, FusedNxmPlan *pl)
{
    if (!nxm_enabled() || log2n != kNxmLog2N || nin < 1 || nout < 2 || P < 2) return false;
    constexpr int M = 1 << (kNxmLog2N - 1), BINROWS = M / 2 / 64;
    const int tiles = (nout + kNxmOT - 1) / kNxmOT;
    const long long base = (long long) BINROWS * tiles;
    const long long K = (long long) nin * (P - 1);
    int ms = 1;
    // (one workgroup per CU, all of them resident at once: with 512 of them in two rounds — and twice the partial spectra for the inverse to add
    // up — the 64 x 8 block took 0.085 ms against 0.083; with 128, half the CUs idle, 0.097)
    while (base * ms * 2 <= 256 && K / (kNxmWaves * ms * 2) >= 6 && (size_t) (ms * 2) * nout * M <= y_elems && ms * 2 <= 8) ms *= 2;
    // (the forward tasks' marks are flagF[0 .. nin): the hint and progress marks live in the same array from kNxmHintBase on — ADVICE r5)
    if ((size_t) ms * nout * M > y_elems || nin > kNxmHintBase) return false;
    pl->ms = ms;
    pl->tiles = tiles;
    pl->kper_old = (int) std::max<long long>(1, (K + kNxmWaves * ms - 1) / (kNxmWaves * ms));
    pl->nmac = (int) (base * ms);
    pl->nfwd = nin;                                     // one whole-frame transform per input
    return true;
}

hipError_t launch_fused_block_nxm(const FusedNxmPlan &pl, float *hist, long long hist_stride, long long hist_mask, const float *in, long long in_stride, long long n0,
                                  long long h, int nin, int nin_alloc, int nout, float2 *X, int Rring, const float2 *H, int hparts, int P, float2 *Y, float *out,
                                  long long out_stride, const float2 *tw, unsigned *bar, unsigned long long *flags, unsigned *arrived, unsigned long long *seq,
                                  hipStream_t fwd_stream, hipStream_t st, bool chained, hipEvent_t ev_begin, hipEvent_t ev_end, unsigned *helped)
{
    constexpr int LOG2N = kNxmLog2N, M = 1 << (LOG2N - 1);
    constexpr size_t lds_fwd = sizeof(float2) * (size_t) lds_padded(M);                  // the whole-frame transform (the helping path runs it too)
    constexpr size_t lds_red = sizeof(float4) * (size_t) kNxmWaves * kNxmOT * 64;
    constexpr size_t lds = lds_fwd > lds_red ? lds_fwd : lds_red;
    static std::atomic<bool> allowed[64];                 // (several engines' host threads come through here: found by ThreadSanitizer)
    int dev = 0;
    (void) hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !allowed[dev].load(std::memory_order_acquire))
    {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(mac_meet_kernel<LOG2N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(fwd_publish_kernel<LOG2N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_fwd);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) allowed[dev].store(true, std::memory_order_release);
    }
    // (HCV_ORDER_CHECK: the forward launch's accesses on its stream, then the hand-over the counters make, then the meeting launch's)
    ORD_ACCESS(fwd_stream, hist, (h - 1) * (long long) M, h * (long long) M, hist_mask + 1, false, "history ring (the hop before the caller's block)");
    ORD_ACCESS(fwd_stream, hist, n0, n0 + (long long) M, hist_mask + 1, true, "history ring (the caller's block filed by the forward launch)");
    ORD_ACCESS(fwd_stream, X, h, h + 1, (long long) Rring, true, "input-spectrum ring slot (n x m forward launch)");
    ORD_MEET(fwd_stream, st);
    ORD_ACCESS(st, X, std::max<long long>(0, h - (P - 1)), h + 1, (long long) Rring, false, "input-spectrum ring slots (n x m multiply-accumulate)");
    ORD_ACCESS(st, Y, 0, 1, 0, true, "partial spectra (n x m multiply-accumulate)");
    FusedNxmParams a;
    a.hist = hist; a.in = in; a.X = X; a.H = H; a.Y = Y; a.tw = tw;
    a.hist_stride = hist_stride; a.in_stride = in_stride; a.hist_mask = hist_mask; a.n0 = n0; a.h = h;
    a.pair_stride4 = (long long) hparts * (M / 2);
    a.out_stride4 = (long long) nin_alloc * a.pair_stride4;
    a.Rring = Rring; a.P = P; a.slot = (int) (h % Rring); a.nin = nin; a.nout = nout;
    a.ms = pl.ms; a.tiles = pl.tiles; a.kper_old = pl.kper_old;
    a.sy = fused_sync_sharded(bar, flags, arrived, *seq, (unsigned) pl.nfwd);
    a.hint = flags + kNxmHintBase;
    a.progress = flags + kNxmProgress;
    a.helped = fused_spin() > 0 ? helped : nullptr;      // (HCV_COOP_SPIN=0 is the test suite's way to run the helping path on purpose)
    // (a forward launch is normally through tens of microseconds before it is needed: a shorter wait than the one-output blocks' — ~0.1 ms —
    // before the multiply-accumulate's workgroups do the transforms themselves)
    a.sy.spin = std::min(a.sy.spin, 256);
    a.hint_wait = (chained && out) ? std::min(kNxmHints, nout * 8) : 0;     // (marks to look at; `chained`: the previous launch on these counters was the block before this one)
    static const int test_delay_us = std::getenv("HCV_NXM_TEST_DELAY_US") ? std::atoi(std::getenv("HCV_NXM_TEST_DELAY_US")) : 0;
    if (test_delay_us > 0) hipLaunchKernelGGL(nxm_delay_kernel, dim3(1), dim3(64), 0, fwd_stream, (unsigned long long) test_delay_us * 100ull);
    hipLaunchKernelGGL((fwd_publish_kernel<LOG2N>), dim3(pl.nfwd), dim3(64 * kNxmWaves), lds_fwd, fwd_stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;                          // (nothing ran: the counters stand where they stood)
    // (the forward launch is in and WILL count its workgroups in, whatever becomes of the launches below: the host's totals follow it.
    // Should one of them be refused, the caller joins the forward stream and runs the separate kernels for this block)
    fused_arrivals_sharded(arrived, (unsigned) pl.nfwd);
    *seq += 1;
    if (ev_begin && (e = hipEventRecord(ev_begin, st)) != hipSuccess) return e;      // (profiling: the multiply-accumulate launch's own time)
    hipLaunchKernelGGL((mac_meet_kernel<LOG2N>), dim3(pl.nmac), dim3(64 * kNxmWaves), lds, st, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (ev_end && (e = hipEventRecord(ev_end, st)) != hipSuccess) return e;
    if (!out) return hipSuccess;                             // (the caller adds the hop to the stage's timeline itself: the extended ladder's pivot stage)
    return launch_rifft_emit_split(LOG2N, Y, pl.ms, (long long) nout * M, 1, nout, out, out_stride, tw, st, a.hint, std::min(kNxmHints, nout * 8), a.sy.seq);
}

// (Engine::init, once per device: HIP loads a translation unit's code object at the first launch of one of its kernels — 0.3 - 0.8 ms on the
// calling thread, which for the kernels of a control section or a restart is the audio thread in mid-stream; asking for a kernel's attributes loads it now)
void preload_fused_nxm()
{
    hipFuncAttributes fa;
    (void) hipFuncGetAttributes(&fa, reinterpret_cast<const void *>((mac_meet_kernel<kNxmLog2N>)));
    (void) hipGetLastError();
}

} // namespace hcv
