// The fused block of an n x m matrix: one whole-hop block of an engine with SEVERAL outputs — Convolver::process for one hop of its
// last stage (Convolver.cpp:138-154 -> NToMonoConvolve.cpp:35-43 -> PartitionedConvolve.cpp:243-385) — as TWO launches on two
// streams that meet inside the second one, in place of four launches on one or two streams that meet through events:
//
//   forward stream:  fwd_publish_kernel     input i's new frame -> X[i][h], the hop filed in the ring  (nin workgroups of 512)
//   main stream:     fused_mac_inv_kernel   Y = sum_{i, p >= 1} X[i][h - p] H[o][i][p]           (every workgroup: its eight k-slices)
//                                           ... wait for the forward transforms' arrival counter ...
//                                           Y += sum_i X[i][h] H[o][i][0]                        (the lead slot: the nin new terms)
//                                           -> the k-slices' sums meet in LDS -> `ms` partial spectra per output in memory
//                                           ... wait for the multiply-accumulate arrival counter ...
//                                           inverse transforms (the first nout R/2 workgroups), summing the `ms` partials as they
//                                           stage them, the hop's samples straight into the caller's block
//
// What this buys (one rank's share of config 4 strong-scaled over 8 GPUs, 64 inputs x 8 output rows, 2 s IRs — the shape row (e) of
// SURVEY section 8 turns on): the separate launches were  transforms (pipe stream, an event each way: ~10 us of the main stream's
// time) -> spectral_mac (48 k-slices through memory: 25 MB written, 25 MB read back by reduce_partials) -> reduce_partials ->
// inverse, 0.111 - 0.115 ms per 8192-sample block around a 71 us multiply-accumulate.  Here the forward transforms of block n run
// on their own stream with NO event in either direction: they are launched first, need nothing of the main stream (only their own
// predecessor, by stream order) and in the steady state run in the shadow of block n - 1's inverse transforms, when most CUs are idle;
// block n's multiply-accumulate needs them only for its last twelfth (the lead slot) and finds their arrival counter — the
// monotonic agent-scope counters of the one-output fused blocks (hcv_fused_sync.h) — long reached.  The k-slices are the 8 waves of a
// 512-thread workgroup (64 lanes = 128 bins each) whose sums meet in LDS, so 4 partial spectra per output go through memory
// instead of 48, and the inverse adds them up while it stages them: no reduce launch, no kernel boundary between any two phases.
//
// Forward progress is by construction, as in the one-output blocks: every wait is bounded and a workgroup whose wait runs out does
// the missing tasks itself (fused_slow_path; the forward transforms included — the forward kernel is the SAME body, bit for bit,
// so whoever runs a task writes the same values).  A forward launch that the device schedules late (another engine's work in
// front of it on the hardware queue) therefore costs time, never correctness: the main stream's workgroups compute the spectra
// themselves, the late launch writes them once more.  The engine joins the forward stream into the main stream by an event
// before anything else touches the rings (a block of another kind, control work: Engine::fence_chains).
//
// Everything the forward kernel writes is RELEASED at agent scope before its workgroup counts itself in (spectra and the ring's new hop
// alike: plain stores, then the L2's dirty lines written back): its consumers run in another launch that started before it ended, and
// read the new spectra with agent-scope loads.  The multiply-accumulate's partial spectra go out as 16-byte write-through stores.

#include "hcv_engine.h"
#include "hcv_fft_split_device.h"
#include "hcv_fft_frames_device.h"
#include "hcv_mac_params.h"

#include <algorithm>
#include <cstdlib>

namespace hcv
{

struct FusedNxmParams
{
    float *hist;
    const float *in;
    float *out;
    float2 *X;                  // [nin][Rring][M]
    const float2 *H;            // [nout][nin_alloc][hparts][M], lead slot first
    float2 *Y;                  // [ms][nout][M]
    const float2 *tw, *tws;
    FusedSyncSharded sy;
    long long hist_stride, in_stride, out_stride, hist_mask, n0, h;
    long long pair_stride4, out_stride4;        // float4 between two inputs' / two outputs' spectra
    int Rring, P, slot, nin, nout;              // P = live partitions, lead slot included; slot = h mod Rring
    int ms, tiles, kper_old, nfwd;              // k-slices through memory, output tiles of 8, (i, p >= 1) terms per wave
    int dbg;                                    // DIAGNOSTIC (HCV_NXM_DBG): 1 = no inverse, 2 = no forward launch / no wait for it, 4 = no lead terms
};

namespace
{
    constexpr int kNxmLog2N = 14, kNxmLog2R = 4, kNxmOT = 8, kNxmWaves = 8;

    // The forward transform of input i's new frame, whole, by one thread group of 512: the engine's own whole-frame transform
    // (hcv_kernels.hip: rfft_frames_direct_kernel — LDS Stockham, the new hop read from the caller's block and filed in the history ring
    // by the first pass, the real post-pass writing the packed spectrum), everything it writes WRITTEN THROUGH at agent scope: its
    // consumers run in another launch that started before this one ends.  A wave's stores are 512 contiguous bytes each — whole lines.
    // (A first version used the residue-split transform of the one-output blocks, 9 workgroups per input: a residue class's bins are 128
    // bytes apart, so its write-through stores were half a million partial-line writes per block, each a read-modify-write at the memory
    // side — the launch took 50 - 90 us and held the multiply-accumulate up; with plain stores and an agent-scope release per workgroup
    // instead, 80 - 150 us.)
    template <int LOG2N> __device__ __forceinline__ void forward_frame(const FusedNxmParams &a, float2 *lds, int tid, int i)
    {
        constexpr int LOG2M = LOG2N - 1, M = 1 << LOG2M, TG = FFTGeom<LOG2M>::TG;
        static_assert(TG == 64 * 8, "the helping path runs this body on the multiply-accumulate kernel's 512 threads");
        const LdsBuf<float2> s = { lds };
        const DirectFrameLoadT<true> ld = { a.hist + (long long) i * a.hist_stride, a.in + (long long) i * a.in_stride, (a.h - 1) * (long long) M, a.hist_mask, a.n0, M / 2, true };
        LdsFFT<LOG2M, TG>::run(ld, LdsIO<float2>{ s }, s, tid, a.tw);
        real_post_store<LOG2M, TG, true>(s, tid, a.tw, a.X + ((long long) i * a.Rring + a.slot) * M);
    }

    // One workgroup of the forward kernel = input i (task i, the helping path's numbering)
    template <int LOG2N>
    __global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6, 6))) void fwd_publish_kernel(FusedNxmParams a)
    {
        extern __shared__ __attribute__((aligned(16))) float2 dynf[];
        const int task = (int) blockIdx.x, tid = (int) threadIdx.x;
        forward_frame<LOG2N>(a, dynf, tid, task);
        grid_publish_sharded(tid, a.sy.flagF + task, a.sy.seq, a.sy.bar, task);
    }

    template <int LOG2N, int LOG2R> struct FusedNxmBodies
    {
        static constexpr int N = 1 << LOG2N, M = N / 2, M2 = M / 2, R = 1 << LOG2R, S = N >> LOG2R, TG = 64 * kNxmWaves, OT = kNxmOT;
        static constexpr int BINROWS = M2 / 64;
        const FusedNxmParams &a;
        float2 *dyn;
        int tid;
        float *nyq;                 // [8 waves][8 outputs] in LDS: bin 0's Nyquist products (see mac_old)
        float4 acc[OT];

        __device__ __forceinline__ void forward(int task) const { forward_frame<LOG2N>(a, dyn, tid, task); }
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
            // nothing atomic about it) — as registers they were eight of every lane's 128 and put the loop's operands into scratch.
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
                put4_agent(reinterpret_cast<float4 *>(a.Y + ((long long) s * a.nout + (o0 + kw)) * M) + b4, sum);
            }
        }
        // inverse task j of output o = m / (R/2): sample classes 2 j, 2 j + 1 of the hop, the `ms` partial spectra added up as they are staged
        __device__ __forceinline__ void inverse(int m) const
        {
            const int o = m / (R / 2), j = m % (R / 2);
            rifft_split_body<LOG2N, LOG2R, true, TG>(dyn, tid, j, a.Y + (long long) o * M, a.ms, (long long) a.nout * M, a.out + (long long) o * a.out_stride - M, a.tw,
                                                     a.tws);
        }
    };

    template <int LOG2N, int LOG2R>
    __device__ __noinline__ void fused_nxm_slow(const FusedNxmParams *ka, float2 *dyn, int tid, int m, bool from_mac_wait, int *slot_b, float *nyq)
    {
        constexpr int R = 1 << LOG2R;
        FusedNxmBodies<LOG2N, LOG2R> b = { *ka, dyn, tid, nyq };
        const int nmac = FusedNxmBodies<LOG2N, LOG2R>::BINROWS * ka->tiles * ka->ms;
        fused_slow_path_sharded(b, ka->sy, m, ka->nfwd, nmac, ka->nout * (R / 2), from_mac_wait, slot_b);
    }

    // Footprint on purpose: the LDS (82 KiB: the inverse's staged spectrum) admits ONE such workgroup per CU, its eight waves take at most
    // 2 x 168 of a SIMD's 512 registers — so beside it every CU keeps room for workgroups of the forward kernel (256 threads, 80
    // registers, 17 KiB).  A first version filled the register file (sixteen waves of 128): the forward launch of the SAME block, started
    // a few microseconds behind it by a stream wait, found no CU to run on until this kernel's waits had run out — 1.5 ms per block.
    template <int LOG2N, int LOG2R>
    __global__ __launch_bounds__(64 * kNxmWaves) __attribute__((amdgpu_waves_per_eu(3, 3))) void fused_mac_inv_kernel(FusedNxmParams a)
    {
        constexpr int R = 1 << LOG2R;
        extern __shared__ __attribute__((aligned(16))) float2 dyn[];
        __shared__ int slot_b[2];
        __shared__ float nyq[kNxmWaves * kNxmOT];
        FusedNxmBodies<LOG2N, LOG2R> b = { a, dyn, (int) threadIdx.x, nyq };
        // (the forward transforms are another launch's workgroups: this launch has none of its own, its first workgroup is task 0)
        if (a.dbg & 16) return;
        if (a.dbg)
        {
            const int m = (int) blockIdx.x;
            const unsigned long long t0 = wall_clock64();
            b.mac_old(m);
            const unsigned long long t1 = wall_clock64();
            if (!(a.dbg & 2)) (void) grid_wait_bounded_sharded<true>(b.tid, a.sy.bar, a.sy.targetA, 1 << 20, slot_b);
            const unsigned long long t2 = wall_clock64();
            if (!(a.dbg & 4)) b.mac_new(m);
            const unsigned long long t3 = wall_clock64();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t3a = wall_clock64();
            __syncthreads();
            const unsigned long long t3b = wall_clock64();
            if (!(a.dbg & 64)) grid_publish_sharded(b.tid, a.sy.flagM + m, a.sy.seq, a.sy.bar + kShards * kShardStride, m);
            const unsigned long long t4 = wall_clock64();
            if ((a.dbg & 32) && b.tid == 0 && (m == 0 || m == 255))
            {
                // DIAGNOSTIC: start / end of this workgroup kept in the unused tail of Y, printed by a later launch
                unsigned long long *log = reinterpret_cast<unsigned long long *>(a.Y + (long long) 40 * a.nout * 8192) + (m ? 512 : 0);
                if (a.sy.seq >= 100 && a.sy.seq < 140)
                {
                    log[2 * (a.sy.seq - 100)] = t0;
                    log[2 * (a.sy.seq - 100) + 1] = t4;
                }
                if (a.sy.seq == 150 || a.sy.seq == 151)
                    for (int k = 0; k < 40; k++)
                        if (log[2 * k]) printf("nxm log wg %d seq %d: start %.2f end %.2f\n", m, 100 + k, (double) (log[2 * k] % 100000000ull) * 0.01, (double) (log[2 * k + 1] % 100000000ull) * 0.01);
            }
            if ((a.dbg & 8) && a.sy.seq == 60 && b.tid == 0 && (t4 - t3) > 800)
                printf("nxm slow publish wg %d: waitcnt %.2f barrier %.2f publish %.2f\n", m, (t3a - t3) * 0.01, (t3b - t3a) * 0.01, (t4 - t3b) * 0.01);
            if ((a.dbg & 1) || m >= a.nout * (R / 2))
            {
                if ((a.dbg & 8) && a.sy.seq == 60 && b.tid == 0)
                {
                    unsigned xcc = 0, hwid = 0;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
                    printf("nxm seq %llu wg %3d xcc %u cu %2u se %u: start %9.2f us, mac_old %.2f, mac_new %.2f, publish %.2f, end %9.2f\n", a.sy.seq, m, xcc & 15,
                           (hwid >> 8) & 15, (hwid >> 13) & 7, (double) (t0 % 100000000ull) * 0.01, (t1 - t0) * 0.01, (t3 - t2) * 0.01, (t4 - t3) * 0.01,
                           (double) (t4 % 100000000ull) * 0.01);
                }
                return;
            }
            (void) grid_wait_bounded_sharded<true>(b.tid, a.sy.bar + kShards * kShardStride, a.sy.targetB, 1 << 20, slot_b + 1);
            const unsigned long long t5 = wall_clock64();
            b.inverse(m);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t6 = wall_clock64();
            if ((a.dbg & 8) && a.sy.seq == 60 && (m == 0 || m == 63) && b.tid == 0)
                printf("nxm wg %d: mac_old %.2f us, wait A %.2f, mac_new %.2f, publish %.2f, wait B %.2f, inverse %.2f\n", m, (t1 - t0) * 0.01, (t2 - t1) * 0.01,
                       (t3 - t2) * 0.01, (t4 - t3) * 0.01, (t5 - t4) * 0.01, (t6 - t5) * 0.01);
            return;
        }
        fused_roles_consumer(b, a.sy, (int) blockIdx.x, a.nout * (R / 2), slot_b, [&](int m, bool from_mac_wait)
                             { fused_nxm_slow<LOG2N, LOG2R>((const FusedNxmParams *) __builtin_amdgcn_kernarg_segment_ptr(), dyn, b.tid, m, from_mac_wait, slot_b, nyq); });
    }

    bool nxm_enabled()
    {
        static const bool on = !(std::getenv("HCV_COOP") && std::atoi(std::getenv("HCV_COOP")) == 0) &&
                               !(std::getenv("HCV_COOP_NXM") && std::atoi(std::getenv("HCV_COOP_NXM")) == 0);
        return on;
    }
}

// The launch plan: `ms` groups of 8 k-slices so that about one workgroup per CU (two at most) streams the spectra, every wave keeping
// at least six terms, the partial spectra within the room Y has
bool fused_block_nxm_plan(int log2n, int nin, int nout, int P, size_t y_elems, FusedNxmPlan *pl)
{
    if (!nxm_enabled() || log2n != kNxmLog2N || nin < 1 || nout < 2 || P < 2) return false;
    constexpr int M = 1 << (kNxmLog2N - 1), BINROWS = M / 2 / 64, R = 1 << kNxmLog2R;
    const int tiles = (nout + kNxmOT - 1) / kNxmOT;
    const long long base = (long long) BINROWS * tiles;
    const long long K = (long long) nin * (P - 1);
    int ms = 1;
    while (base * ms * 2 <= 512 && K / (kNxmWaves * ms * 2) >= 6 && (size_t) (ms * 2) * nout * M <= y_elems && ms * 2 <= 8) ms *= 2;
    const long long nmac = base * ms;
    if ((size_t) ms * nout * M > y_elems) return false;
    if ((long long) nout * (R / 2) > nmac) return false;                       // the inverse's workgroups are the first of the multiply-accumulate's
    if (nmac > kFusedNxmMacTasks || nin > kFusedFwdTasks) return false;
    pl->ms = ms;
    pl->tiles = tiles;
    pl->kper_old = (int) std::max<long long>(1, (K + kNxmWaves * ms - 1) / (kNxmWaves * ms));
    pl->nmac = (int) nmac;
    pl->nfwd = nin;                                     // one whole-frame transform per input
    return true;
}

hipError_t launch_fused_block_nxm(const FusedNxmPlan &pl, float *hist, long long hist_stride, long long hist_mask, const float *in, long long in_stride, long long n0,
                                  long long h, int nin, int nin_alloc, int nout, float2 *X, int Rring, const float2 *H, int hparts, int P, float2 *Y, float *out,
                                  long long out_stride, const float2 *tw, unsigned *bar, unsigned long long *flags, unsigned *arrived, unsigned long long *seq,
                                  hipStream_t fwd_stream, hipStream_t st)
{
    constexpr int LOG2N = kNxmLog2N, LOG2R = kNxmLog2R, M = 1 << (LOG2N - 1), S = 1 << (LOG2N - LOG2R), R = 1 << LOG2R;
    const float2 *tws = fft_split_sub_table(LOG2N - LOG2R);
    if (!tws) return hipErrorInvalidValue;
    constexpr size_t lds_inv = sizeof(float2) * (size_t) (M + lds_padded(S) + S + R), lds_red = sizeof(float4) * (size_t) kNxmWaves * kNxmOT * 64;
    constexpr size_t lds_fwd = sizeof(float2) * (size_t) lds_padded(M);                 // (the helping path's forward transform: within lds_inv)
    constexpr size_t lds = lds_inv > lds_red ? lds_inv : lds_red;
    static_assert(lds_fwd <= lds_inv, "");
    static bool allowed[64] = {};
    int dev = 0;
    (void) hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !allowed[dev])
    {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fused_mac_inv_kernel<LOG2N, LOG2R>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int) lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(fwd_publish_kernel<LOG2N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_fwd);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) allowed[dev] = true;
    }
    FusedNxmParams a;
    a.hist = hist; a.in = in; a.out = out; a.X = X; a.H = H; a.Y = Y; a.tw = tw; a.tws = tws;
    a.hist_stride = hist_stride; a.in_stride = in_stride; a.out_stride = out_stride; a.hist_mask = hist_mask; a.n0 = n0; a.h = h;
    a.pair_stride4 = (long long) hparts * (M / 2);
    a.out_stride4 = (long long) nin_alloc * a.pair_stride4;
    a.Rring = Rring; a.P = P; a.slot = (int) (h % Rring); a.nin = nin; a.nout = nout;
    a.ms = pl.ms; a.tiles = pl.tiles; a.kper_old = pl.kper_old; a.nfwd = pl.nfwd;
    a.sy = fused_sync_sharded(bar, flags, arrived, *seq, (unsigned) pl.nfwd, (unsigned) pl.nmac, kFusedNxmMacTasks);
    static const int dbg = std::getenv("HCV_NXM_DBG") ? std::atoi(std::getenv("HCV_NXM_DBG")) : 0;
    a.dbg = dbg;
    static const int dbg_lds = std::getenv("HCV_NXM_LDS") ? std::atoi(std::getenv("HCV_NXM_LDS")) : 0;
    const size_t lds_use = dbg_lds > 0 ? (size_t) dbg_lds : lds;
    if (dbg & 2)
    {
        hipLaunchKernelGGL((fused_mac_inv_kernel<LOG2N, LOG2R>), dim3(pl.nmac), dim3(64 * kNxmWaves), lds_use, st, a);
        fused_arrivals_sharded(arrived, 1, (unsigned) pl.nmac);
        *seq += 1;
        return hipGetLastError();
    }
    hipLaunchKernelGGL((fwd_publish_kernel<LOG2N>), dim3(pl.nfwd), dim3(512), lds_fwd, fwd_stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;                          // (nothing ran: the counters stand where they stood)
    hipLaunchKernelGGL((fused_mac_inv_kernel<LOG2N, LOG2R>), dim3(pl.nmac), dim3(64 * kNxmWaves), lds, st, a);
    e = hipGetLastError();
    if (e != hipSuccess)
    {
        // The forward launch is in and WILL count its workgroups in: the host's total of that counter follows it, the launch sequence
        // number too (its flags carry it); the multiply-accumulate counter stands.  The caller runs the separate kernels for this block.
        fused_arrivals_sharded(arrived, 0, (unsigned) pl.nfwd);
        *seq += 1;
        return e;
    }
    fused_arrivals_sharded(arrived, 0, (unsigned) pl.nfwd);
    fused_arrivals_sharded(arrived, 1, (unsigned) pl.nmac);
    *seq += 1;
    return hipSuccess;
}

} // namespace hcv
