// The per-block scheduler of the engine: which kernels one process() block launches, on which streams and in which order
// (stream plan in front of enqueue_chunk), and the host- and device-pointer entry points that feed it.

#include "hcv_engine_impl.h"

#include <chrono>

namespace hcv
{

// (the lead slot's nin terms are a short reduction per output: up to six k-slices fill the chip — measured on 64 x 64, 2 s IRs, staged /
// registered host calls in ms: 2 slices 0.697 / 0.649, 3: 0.704 / 0.658, 6: 0.668 / 0.625, 8: 0.667 / 0.622, the split off: 0.736 / 0.662; the
// old partitions keep the k-slices of the whole launch, and the reduction in front of the inverse takes all of them)
constexpr int kPreNewSlices = 6;


// How many launches the deferred accumulation of a hop is spread over: one per partition up to kBgSlices for real-time calls (short
// launches, evenly through the hop: no call waits long behind one); for hop-sized calls of a long stage — throughput callers, the
// extended ladder's rungs under 8192-sample blocks — no more than half as many as the hop has calls, so that every slice carries
// two or three partitions (a 134 MB launch runs at 2 TB/s between its ramps, a 400 MB one at 5).
static int bg_slice_count(int bg_parts, uint64_t M, uint32_t B)
{
    int slices = std::max(1, std::min(kBgSlices, bg_parts));
    if (B >= 4096 && M >= 8 * (uint64_t) B) slices = std::max(1, std::min<int>(slices, (int) (M / B) / 2));
    return slices;
}

// Launch the background slices of `st` that are due: all of them at the hop's boundary, otherwise in proportion to the
// part of the hop's samples that has arrived with this call.  Slice s covers partitions 1 + [a, b) of hop pre_hop:
// a (b - a)-partition MAC at hop pre_hop - 1 - a over the spectra shifted by 1 + a partitions.
bool Engine::advance_background(const Block &blk, Stage &st, bool boundary)
{
    HCV_BLOCK_LOCALS(blk);
    const hipStream_t sS = serial ? mStream : st.stream;
    if (st.pre_hop < 0 || st.bg_launched >= st.bg_slices) return true;
    // Slices come due a little AHEAD of the even grid (kBgLead = 192 samples; stages of at least 4096 points): on the grid the
    // tail's slice k falls due at sample k M / 16 = a multiple of 512, i.e. in the very call that also carries the hop boundaries of the
    // 256- and 1024-point stages — with 32-sample calls of the 64 x 64 / 10 s engine every sixteenth call took 0.48 ms (a 0.2 ms tail
    // slice on top of two boundaries) against 0.075 ms for its neighbours.  192 samples early the slice lands in a plain call at 32
    // and 64 samples per call (the 10th of 16, the 5th of 8) and, at 128 samples per call, in the call BEFORE the one with the
    // 1024-point stage's boundary (p99 0.45 -> 0.28 ms there; 64 samples of lead, enough for the smaller calls, is not).
    constexpr long long kBgLead = 192;
    const long long lead = st.M >= 2048 ? std::min<long long>(kBgLead, (long long) st.M / (2 * std::max(1, st.bg_slices))) : 0;
    const long long into = (long long) (n0 + B) - st.pre_hop * (long long) st.M + lead;
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
        MacShape sb = mac_shape(st, /* P */ b - a, /* Pcap */ st.hparts(),
                                /* nin */ (int) nin_act, /* nin_alloc */ (int) mNinAlloc, /* nout */ (int) mCfg.nout, /* diag */ mCfg.diag ? 1 : 0,
                                /* T */ 1, /* max_ksplit */ (int) std::max<size_t>(1, st.y_elems / ((size_t) mCfg.nout * st.M)));       // (full matrix only)
        MacPlan pb;
        mac_plan(sb, pb);
        // A slice of a long stage fills the chip with its bin blocks alone (the extended ladder's rungs: 128 and 1024 bin blocks per
        // output tile): no k-slices then — partial sums through memory and their reduction launch moved as many bytes again as the
        // slice's spectra (c5 on the ladder: 34 + 16 us of reduction per 8192-sample block) — and the sums go straight into the slot.
        const bool direct = pb.ksplit > 1 && (long long) pb.binblocks * pb.outtiles * pb.tz >= 256;
        if (direct)
        {
            sb.max_ksplit = 1;
            mac_plan(sb, pb);
            if ((long long) pb.binblocks * pb.outtiles * pb.tz < 512 && pb.ot == 8)
            {
                // (one workgroup per CU streams at a third of what two do: four outputs per thread, twice the workgroups; the input
                // spectra are read twice, a sixteenth of the slice's bytes)
                sb.ot_cap = 4;
                mac_plan(sb, pb);
            }
        }
        const long long hop = st.pre_hop - 1 - a;
        const bool bcheck = (hop - st.max_hv) < (long long) (b - a) - 1;
        float2 *scratch = st.Yq[0];                             // every use of this stage's scratch is ordered on its stream
        if (!mac(st, sb, pb, st.Ht() + (size_t) (1 + a) * st.M, pb.ksplit == 1 ? slot : scratch, hop, bcheck, sS)) return false;
        if (pb.ksplit > 1) HCV_TRY(launch_reduce_partials(scratch, pb.ksplit, slot_elems, slot_elems, sS, slot));       // (the sum goes straight into the slot)
        HCV_TRY(hipEventRecord(st.bg_done, sS));
        st.bg_pending = true;
    }
    return true;
}

// Back from whole-hop mode: this stage's own machinery was not run for a while.  Rebuild the input spectra its partitions reach
// back to from the history ring (not for the last stage: it kept its ring up to date), and compute the hop just before this
// block — its result is emitted during the first hop of the block (every stage has one hop of latency).
bool Engine::catch_up_stage(const Block &blk, Stage &st, long long h_first, bool rebuild_spectra)
{
    HCV_BLOCK_LOCALS(blk);
    const hipStream_t sS = blk.stage_stream(st.stream);
    HCV_TRY(wt(sS, mEvInput[q]));
    st.Y = st.Yq[q];
    if (rebuild_spectra)
    {
        const long long h_lo = std::max<long long>(0, h_first - (long long) st.P);
        HCV_TRY(launch_rfft_frames(st.log2n, mHist, mHistLen, hmask, h_lo, (int) (h_first - h_lo), (int) rows_in, st.X, (int) st.R, st.tw, &st.big, sS));
    }
    const MacShape sc = mac_shape(st, /* P */ (int) std::min<long long>(st.P, h_first), /* Pcap */ st.hparts(),
                                  /* nin */ (int) nin_act, /* nin_alloc */ (int) mNinAlloc, /* nout */ (int) nout_act, /* diag */ mCfg.diag ? 1 : 0,
                                  /* T */ 1, /* max_ksplit */ (int) std::max<size_t>(1, st.y_elems / ((size_t) nout_act * st.M)));
    MacPlan pc;
    mac_plan(sc, pc);
    const bool ccheck = (h_first - 1 - st.max_hv) < (long long) st.P - 1;
    const long long c_elems = (long long) nout_act * st.M;
    if (!mac(st, sc, pc, st.Ht(), st.Y, h_first - 1, ccheck, sS)) return false;
    HCV_TRY(launch_reduce_partials(st.Y, pc.ksplit, c_elems, c_elems, sS));
    HCV_TRY(wt(sS, mEvEmit[q]));        // emit(k-2) has cleared the timeline span reused now
    HCV_TRY(launch_rifft_overlap_add(st.log2n, st.Y, 1, c_elems, h_first - 1, 1, (int) nout_act, st.timeline, st.tl_len, st.tl_len - 1, st.tw,
                                     &st.big, sS));
    HCV_TRY(rec(st.done[q], sS));          // (recorded again below when this block has hops of its own)
    HCV_TRY(wt(mStream, st.done[q]));
    return true;
}

// One stage's share of a block: forward FFTs of the hops the block completes, the spectral MAC (with the head partition where
// the mode has one), the inverse into the stage's timeline — or, when the block completes no hop of this stage, the deferred
// slices that are due.  sj counts from the largest stage (0 = the tail, launched first: its MAC is the critical path).
bool Engine::enqueue_stage(Block &blk, size_t si, size_t sj)
{
    HCV_BLOCK_LOCALS(blk);
    Stage &st = *mStages[si];
    // Two lanes for the pivot stage of an extended ladder: a whole-hop block's chain on that stage — transforms, multiply-accumulate,
    // reduction, inverse: four latency-bound launches, ~95 us for c5's 16 x 16 x 8 partitions — ran on ONE stream, block after block, and
    // WAS the ladder's step (0.137 ms) while the rungs' slices beside it left the memory system half idle.  Blocks of odd parity take the
    // stage's second stream: the only thing block k's chain needs of block k - 1's is its forward transforms (the history ring's hop, and
    // the ring of spectra in stream order behind them) — one event, the block's input event; the inverses add into disjoint spans of the
    // timeline, the partial sums are double-buffered by parity as ever.  (HCV_PIVOT_LANES = 1: one lane.)  c5 on the ladder 0.1355 -> 0.130 ms per
    // step, 64 x 64 / 10 s 0.722 -> 0.707.  (The block's emit on the lane as well, instead of the main stream — which shares a hardware queue
    // with one of the lanes —: built, no gain, 0.128 - 0.131 against 0.131; not kept.)
    const bool lane2 = !serial && whole_hops && si == last && st.stream2 != nullptr && (q & 1);
    const hipStream_t sS = serial ? mStream : (lane2 ? st.stream2 : st.stream);
    if (whole_hops && entering && si <= last)
    {
        // what this stage has pending duplicates what the pivot stage's whole-hop convolution now computes (for the pivot stage
        // itself: its own partitions' result for the block's first hop): drop it — after the emit that may still be reading
        // it — together with any plan of a deferred accumulation.  (The rungs of an extended ladder, si > last, go on as they were.)
        HCV_TRY(wt(sS, mEvEmit[q ^ 1]));
        HCV_TRY(hipMemsetAsync(st.timeline, 0, sizeof(float) * mCfg.nout * st.tl_len, sS));
        HCV_TRY(rec(st.done[q], sS));
        HCV_TRY(wt(mStream, st.done[q]));
        st.pre_hop = -1;
    }
    if (whole_hops && si < last) return true;
    if (!blk.emit_first)
    {
        blk.src.timeline[blk.src.count] = st.timeline;          // the ring may still hold hops of earlier calls
        blk.src.stride[blk.src.count] = st.tl_len;
        blk.src.mask[blk.src.count] = st.tl_len - 1;
        blk.src.count++;
    }
    const bool tail_head_here = whole_hops && si == last;
    int head_ksplit = 1;
    const bool head_here = head_fft && si == 0;
    const float2 *head_spec = mHeadSpec;
    float2 *head_y = mHeadYq[q];
    if (!st.P && !head_here && !tail_head_here) return true;
    const long long h_first = n0 / st.M;
    const int T = (int) ((n0 + B) / st.M - h_first);
    if (leaving && si <= last && st.P && h_first >= 1 && !catch_up_stage(blk, st, h_first, /* rebuild_spectra */ si != last)) return false;
    // Deferred mode: calls shorter than the hop (real-time block sizes) with more than one partition.
    static const bool allow_defer = !(std::getenv("HCV_DEFER") && std::atoi(std::getenv("HCV_DEFER")) == 0);
    if (T <= 0)
    {
        if (st.pre_hop >= 0 && (!full_matrix || st.pre_hop != h_first)) st.pre_hop = -1;     // the plan no longer fits what is being processed
        if (st.pre_hop < 0 && allow_defer && full_matrix && st.P > 1 && B < st.M && h_first >= 1)
        {
            // no plan for the hop in progress (the first small call after large ones, or control work dropped it): make it
            // now — the frames it needs are complete — so that the boundary does not pay the whole accumulation inline
            st.bg_parts = (int) std::min<long long>(st.P - 1, h_first);
            st.bg_slices = bg_slice_count(st.bg_parts, st.M, B);
            st.bg_launched = 0;
            st.pre_hop = h_first;
        }
        if (st.pre_hop >= 0 && !advance_background(blk, st, false)) return false;
        return true;
    }

    // one stream per stage: forward FFT, MAC, inverse.  (Side streams for a large stage's FFTs were built and measured
    // twice — dedicated ones and the input / main streams — and were slower each time: c5 2.12 / 2.33 vs 1.98 ms per step.)
    hipStream_t sM = sS, sF = sS, sI = sS;
    st.Y = st.Yq[q];

    // The fused block (hcv_fft_split.hip: fused_block_1x1_kernel / fused_block_nx1_kernel; HCV_COOP = 0 switches it off): forward
    // transforms, multiply-accumulate and inverse of a steady-state one-hop block of an engine with ONE output as ONE launch with
    // in-launch hand-overs instead of two or three kernel boundaries (c1: 0.0166 -> 0.0101 ms per block).  With the MAC's HIP events
    // on (profiling) the block takes the separate launches, so that the statistics keep their meaning.
    if (blk.nxm && tail_head_here && T == 1)
    {
        const int Pw = (int) (st.P + st.lead);
        EventPair *pe = nullptr;
        if (mProfiling)
        {
            // (the multiply-accumulate launch's own time, as the separate kernels' statistics have it)
            for (EventPair *c : mEvents)
                if (!c->live) { pe = c; break; }
            if (!pe)
            {
                pe = new EventPair();
                HCV_TRY(hipEventCreate(&pe->a));
                HCV_TRY(hipEventCreate(&pe->b));
                mEvents.push_back(pe);
            }
            pe->stage = si;
        }
        const hipError_t fe = launch_fused_block_nxm(blk.nxm_plan, mHist, mHistLen, hmask, blk.din, blk.in_stride, n0, h_first, (int) rows_in, (int) mNinAlloc,
                                                     (int) nout_act, st.X, (int) st.R, st.Hs, st.hparts(), Pw, st.Y, blk.direct_out ? blk.dout : nullptr, blk.out_stride,
                                                     st.tw, st.coop_bar, st.coop_flags, st.coop_arrived_nxm, &st.coop_seq, mPipeStream, sS, /* chained */ mNxmRun > 0,
                                                     pe ? pe->a : nullptr, pe ? pe->b : nullptr, st.nxm_helped_dev);
        if (pe && fe == hipSuccess) pe->live = true;
        if (fe == hipSuccess && !blk.direct_out)
        {
            // (an extended ladder's pivot stage: the hop goes into the stage's timeline, emitted with NO latency — hop h at h M — beside the
            // rungs' timelines; the inverse adds the partial spectra up as it loads them)
            HCV_TRY(wt(sS, mEvEmit[q]));        // emit(k-2) has cleared the timeline span reused now
            HCV_TRY(launch_rifft_overlap_add(st.log2n, st.Y, blk.nxm_plan.ms, (long long) nout_act * st.M, h_first - 1, 1, (int) nout_act, st.timeline, st.tl_len,
                                             st.tl_len - 1, st.tw, &st.big, sS));
        }
        mFwdPending = true;                     // (a refused second launch leaves the first on the pipe stream all the same)
        if (fe == hipSuccess)
        {
            if (mNxmRun % mNxmEvery == 0) HCV_TRY(hipEventRecord(mEvNxmEnd[(mNxmRun / mNxmEvery) & 3], sS));        // (see the back-pressure note in enqueue_chunk)
            mNxmRun++;
            mPrevNxm = true;
            st.launches++;
            st.hops += 1;
            st.last_ksplit = (uint32_t) blk.nxm_plan.ms;
            st.last_ot = 8;
            st.last_tt = 1;
            st.last_parts = (uint32_t) Pw;
            st.fused_launches++;
            st.steady_launches++;                   // (the unchecked, nontemporal instantiation of this path: it takes no other)
            HCV_TRY(rec(mEvInput[q], sS));
            if (blk.direct_out) HCV_TRY(rec(mEvEmit[q], sS));      // (the inverse delivered the block itself; otherwise the emit launch records it)
            HCV_TRY(rec(st.done[q], sS));
            HCV_TRY(wt(mStream, st.done[q]));
            st.pre_hop = -1;
            return true;
        }
        // refused: the separate kernels, from now on — behind whatever the forward launch may still write
        (void) hipGetLastError();
        st.coop_off = true;
        blk.nxm = false;
        if (!join_forward_stream()) return false;
    }
    const bool fuse_one = T == 1 && fused_block_1x1_applies(st.log2n);
    const bool fuse_hops = T > 1 && rows_in == 1 && fused_block_hops_applies(st.log2n, T);       // (several hops of a short stage: config 2)
    if (tail_head_here && direct_in && blk.direct_out && !blk.pipe2 && serial && rows_in >= 1 && nout_act == 1 && mCfg.nout == 1 && !mCfg.diag && !mProfiling &&
        !st.gh_count && !st.coop_off && st.coop_flags && (fuse_one || fuse_hops))
    {
        const int Pw = (int) (st.P + st.lead);
        const long long h_mac = h_first - (long long) (1 - st.lead);
        const long long p_live = std::max<long long>(0, std::min<long long>(Pw, h_mac + T));
        const bool wcheck = (h_mac - st.max_hv) < (long long) Pw - 1;
        if (!wcheck && p_live >= 1 && (long long) rows_in * p_live <= 2048 && (!fuse_hops || (p_live == Pw && st.y_elems >= (size_t) T * st.M)))
        {
            hipError_t fe;
            if (fuse_hops)
                fe = launch_fused_block_hops(st.log2n, mHist, hmask, blk.din, n0, h_first, T, st.X, (int) st.R, st.Hs, (int) p_live, h_mac, st.Y, blk.dout, st.tw,
                                             st.coop_bar, st.coop_flags, st.coop_arrived, &st.coop_seq, sS);
            else if (rows_in == 1 && p_live <= 16)
                fe = launch_fused_block_1x1(st.log2n, mHist, hmask, blk.din, n0, h_first, st.X, (int) st.R, st.Hs, (int) p_live, h_mac, st.Y, blk.dout, st.tw,
                                            st.coop_bar, st.coop_flags, st.coop_arrived, &st.coop_seq, sS);
            else
                fe = launch_fused_block_nx1(st.log2n, mHist, mHistLen, hmask, blk.din, blk.in_stride, n0, h_first, (int) rows_in, st.X, (int) st.R, st.Hs,
                                            (long long) st.hstride(), (int) p_live, h_mac, st.Y, blk.dout, st.tw, st.coop_bar, st.coop_flags, st.coop_arrived, &st.coop_seq, sS);
            if (fe == hipSuccess)
            {
                st.launches++;
                st.hops += (uint64_t) T;
                st.last_ksplit = 1;
                st.last_ot = 1;
                st.last_tt = (uint32_t) T;
                st.last_parts = (uint32_t) p_live;
                st.fused_launches++;
                HCV_TRY(rec(mEvInput[q], sS));
                HCV_TRY(rec(mEvEmit[q], sS));
                HCV_TRY(rec(st.done[q], sS));
                HCV_TRY(wt(mStream, st.done[q]));
                st.pre_hop = -1;
                return true;
            }
            // refused (nothing ran, the counters stand where they stood): the separate kernels, from now on
            st.coop_off = true;
        }
    }

    const bool direct_here = direct_in && si == last;       // (a rung of the extended ladder reads its frames from the history ring)
    if (direct_here && blk.pipe2)
    {
        // two-stream pipeline of a small engine: this block's transforms run on the pipe stream, beside the previous block's
        // multiply-accumulate and inverse on the main stream.  The ring slots they write are read by the MAC of the block two
        // back at the latest (R = Pcap + 2 Tmax), hence the wait for that block's end; the MAC of this block waits for them.
        // Single-hop blocks need less: the ring holds Pcap + 2 Tmax >= Pcap + 4 spectra and a whole-hop MAC reads Pcap + 1 of them, so
        // the slot of hop n was last read by the MAC of block n - 4.  Waiting for the end of block n - 3 instead of block n - 2
        // takes the wait off the cycle (transforms of n | MAC .. inverse of n - 1): with the closer event the transforms can only
        // start when block n - 2 has ended and, hand-over included, end after block n - 1 does.  (HCV_PIPE_DEPTH = 1: the closer event.)
        // An event record costs the recording stream about 4 us here, so a pipelined block records ONE end event (mEvPipeEnd, a ring of
        // four) in place of st.done[q]; the first block after the pipe stream was lined up behind the main stream (mEvSerial) needs
        // no wait at all.
        // In a run of single-hop blocks only every second block records its end: the transforms of
        // block n need the MAC of block n - 4 done, and the newer of the last two recorded ends that is not younger than n - 3 — or,
        // failing that, the latest one — covers it.
        const long long cur = (long long) mPipeSeq;
        const bool far = T == 1 && mPipeRun >= 3 && mPipeSince >= 3 && st.Tmax >= 2;
        blk.pipe_far = far;
        if (mPipeSince >= 2 && mPipeRecA >= 0)      // (blocks from before the line-up are behind mEvSerial, which this stream has waited for)
        {
            const long long need = cur - (far ? 4 : 2);            // the block whose MAC must have ended
            const long long pick = (mPipeRecB >= need && mPipeRecB >= 0) ? mPipeRecB : mPipeRecA;
            HCV_TRY(hipStreamWaitEvent(mPipeStream, mEvPipeEnd[pick & 3], 0));
        }
        HCV_TRY(launch_rfft_frames_direct(st.log2n, mHist, mHistLen, hmask, blk.din, blk.in_stride, n0, h_first, T, (int) rows_in, st.X, (int) st.R, st.tw,
                                          mPipeStream));
        HCV_TRY(hipEventRecord(mEvPipe[q], mPipeStream));
        HCV_TRY(hipStreamWaitEvent(sM, mEvPipe[q], 0));
    }
    else if (direct_here)
    {
        // the transforms read the caller's block themselves and file it in the history ring: no scatter, no wait for one
        // (two lanes: behind the previous block's transforms, which ran on the other one)
        if (st.stream2 && !serial) HCV_TRY(wt(sF, mEvInput[q ^ 1]));
        HCV_TRY(launch_rfft_frames_direct(st.log2n, mHist, mHistLen, hmask, blk.din, blk.in_stride, n0, h_first, T, (int) rows_in, st.X, (int) st.R, st.tw, sF));
        HCV_TRY(rec(mEvInput[q], sF));          // "the block's input is in the ring"
    }
    else
    {
        HCV_TRY(wt(sF, mEvInput[q]));
        if (blk.gate && tail_gate >= 2) HCV_TRY(wt(sF, blk.gate));
        HCV_TRY(launch_rfft_frames(st.log2n, mHist, mHistLen, hmask, h_first, T, (int) rows_in, st.X, (int) st.R, st.tw, &st.big, sF));
    }
    if (blk.gate && tail_gate == 1) HCV_TRY(wt(sM, blk.gate));

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

    if (tail_head_here)
    {
        // Whole-hop mode: ONE zero-latency uniform convolution over the lead slot (IR[0 : M)) and the stage's own partitions,
        //     Y(h) = sum_{p' = 0 .. P} X[h - p'] H'[p'],
        // one spectral_mac launch, one inverse, emitted in the hop's own slot.  Nothing is parked for the next block (the
        // stage's own partitions, which normally run one hop ahead of their emission, are simply computed in the block that
        // emits them), so the head chain, its stream hand-overs and the second inverse of the earlier scheme are gone.
        // (a lone stage has no lead slot: its partitions' one hop of latency is taken by evaluating hops h - 1 .. in block h)
        const int Pw = (int) (st.P + st.lead);
        const long long h_mac = h_first - (long long) (1 - st.lead);
        const long long p_live = std::max<long long>(0, std::min<long long>(Pw, h_mac + T));
        const size_t ks_cap = std::max<size_t>(1, st.y_elems / ((size_t) T * nout_act * st.M));
        MacShape sw = mac_shape(st, /* P */ (int) p_live, /* Pcap */ st.hparts(),
                                /* nin */ (int) nin_act, /* nin_alloc */ (int) mNinAlloc, /* nout */ (int) nout_act, /* diag */ mCfg.diag ? 1 : 0,
                                /* T */ T, /* max_ksplit */ (int) ks_cap);
        const bool wcheck = (h_mac - st.max_hv) < (long long) Pw - 1;
        sw.steady = (!wcheck && !st.gh_count) ? 1 : 0;      // (offline calls of 32 hops or more: the matrix cores, hcv_mac_mfma.hip)
        // (... and their ramp-up after a global reset: one bound for every pair, hop 0 — the kernel stages earlier hops as silence)
        bool uniform = false;
        if (wcheck && !st.gh_count && st.hv_zero && T >= 32)
        {
            sw.steady = 1;
            sw.hop_min = 0;
            uniform = true;
        }
        MacPlan pw;
        // A host-pointer call whose partitions >= 1 went out ahead of the upload (host_pre_mac): slices [0, ks) of Y hold them; what is
        // left is the lead slot's nin terms over the NEW spectra, into the slices behind.
        const bool pre = mPre.valid && mPre.block == mBlockCount && !serial && !wcheck && T == 1 && mPre.h_mac == h_mac && mPre.nin == nin_act &&
                         mPre.nout == nout_act && sM == st.stream && !mProfiling;
        mPre.valid = false;
        if (pre)
        {
            MacShape sn = mac_shape(st, /* P */ 1, /* Pcap */ st.hparts(), /* nin */ (int) nin_act, /* nin_alloc */ (int) mNinAlloc, /* nout */ (int) nout_act, /* diag */ 0,
                                    /* T */ 1, /* max_ksplit */ kPreNewSlices);
            MacPlan pn;
            mac_plan(sn, pn);
            if (!mac(st, sn, pn, st.Hs, st.Y + (size_t) mPre.ksplit * nout_act * st.M, h_mac, false, sM)) return false;
            pw = pn;
            pw.ksplit = mPre.ksplit + pn.ksplit;            // (the inverse — or the reduction in front of it — takes all of them)
            pw.nt = 1;
            st.host_pre_launches++;
        }
        else
        {
            mac_plan(sw, pw);
            if (uniform && !pw.mfma)
            {
                // (not a shape the matrix-core kernel takes: the checked register tiles, as ever)
                uniform = false;
                sw.steady = 0;
                mac_plan(sw, pw);
            }
            if (!begin_event()) return false;
            if (!mac(st, sw, pw, st.Hs, st.Y, h_mac, wcheck && !uniform, sM)) return false;
            if (ev) HCV_TRY(hipEventRecord(ev->b, sM));
        }
        st.launches++;
        st.hops += (uint64_t) T;
        st.last_ksplit = (uint32_t) pw.ksplit;
        st.last_ot = (uint32_t) pw.ot;
        st.last_tt = (uint32_t) pw.tt;
        st.last_parts = (uint32_t) p_live;
        if (!wcheck && pw.nt) st.steady_launches++;
        const long long w_elems = (long long) T * nout_act * st.M;
        // (every workgroup of a residue-split inverse stages the WHOLE spectrum: it takes the one summed slice)
        const bool fold_w = pw.ksplit > 1 && pw.ksplit <= kFoldMax && ((long long) T * nout_act >= 16 || serial) &&
                            !(blk.direct_out && fft_split_applies(st.log2n, T * (int) nout_act));
        if (!fold_w) HCV_TRY(launch_reduce_partials(st.Y, pw.ksplit, w_elems, w_elems, sI));
        if (blk.direct_out)
        {
            // the inverse delivers the block itself; "emit" of this block = the end of this launch
            HCV_TRY(launch_rifft_emit(st.log2n, st.Y, fold_w ? pw.ksplit : 1, w_elems, T, (int) nout_act, blk.dout, blk.out_stride, st.tw, sI));
            HCV_TRY(rec(mEvEmit[q], sI));
        }
        else
        {
            HCV_TRY(wt(sI, mEvEmit[q]));         // emit(k-2) has cleared the timeline span reused now
            HCV_TRY(launch_rifft_overlap_add(st.log2n, st.Y, fold_w ? pw.ksplit : 1, w_elems, h_first - 1, T, (int) nout_act, st.timeline, st.tl_len,
                                             st.tl_len - 1, st.tw, &st.big, sI));   // h_first - 1: emitted with NO latency (hop h at h*M)
        }
        if (blk.pipe2)
        {
            // (serial blocks record no events otherwise; in a run of single-hop blocks every second one does)
            if (!blk.pipe_far || (long long) mPipeSeq - mPipeRecA >= 2)
            {
                HCV_TRY(hipEventRecord(mEvPipeEnd[mPipeSeq & 3], sI));
                mPipeRecB = mPipeRecA;
                mPipeRecA = (int64_t) mPipeSeq;
            }
            mPipeSeq++;
            mPipeSince++;
            mPipeRun = T == 1 ? mPipeRun + 1 : 0;
        }
        HCV_TRY(rec(st.done[q], sI));
        HCV_TRY(wt(mStream, st.done[q]));
        st.pre_hop = -1;
        return true;
    }

    if (head_here)
    {
        // head = partition "-1": Yh[t][o] = sum_i X[i][h_t] * Hhead[o][i]
        // (the buffer holds Tmax hops of one slice: a call of fewer hops — the real-time sizes — splits the reduction over the inputs
        //  into the room that is left, up to eight ways; one slice meant every wave walking all the inputs, 85 us on 64 x 64)
        const MacShape hs = mac_shape(st, /* P */ 1, /* Pcap */ 1,
                                      /* nin */ (int) nin_act, /* nin_alloc */ (int) mNinAlloc, /* nout */ (int) nout_act, /* diag */ mCfg.diag ? 1 : 0,
                                      /* T */ T, /* max_ksplit */ (int) std::max<long long>(1, std::min<long long>(8, (long long) st.Tmax / std::max(1, T))));
        MacPlan hp;
        mac_plan(hs, hp);
        if (!mac(st, hs, hp, head_spec, head_y, h_first, false, sM)) return false;
        head_ksplit = hp.ksplit;
    }

    // hops since the last global reset bound how many partitions can have input yet (mValidPartitions in the
    // reference, PartitionedConvolve.cpp:285,322,373): right after a reset the reduction is short
    const long long p_live = std::min<long long>(st.P, h_first + T);
    MacShape sh = mac_shape(st, /* P */ (int) p_live, /* Pcap */ st.hparts(),
                            /* nin */ (int) nin_act, /* nin_alloc */ (int) mNinAlloc, /* nout */ (int) nout_act, /* diag */ mCfg.diag ? 1 : 0,
                            /* T */ T, /* max_ksplit */ (int) std::max<size_t>(1, st.y_elems / ((size_t) T * nout_act * st.M)));
    const bool check = (h_first - st.max_hv) < (long long) st.P - 1;
    sh.steady = (!check && !st.gh_count && p_live == (long long) st.P) ? 1 : 0;
    const long long y_elems = (long long) T * nout_act * st.M;

    const bool defer = allow_defer && st.P > 0 && T == 1 && B < st.M && st.P > 1 && p_live >= 1 && nout_act == mCfg.nout &&
                       nin_act == (mCfg.diag ? mCfg.nout : mCfg.nin);
    const bool have_pre = defer && st.pre_hop == h_first;

    // ---- MAC phase (stream sM)
    MacPlan pl;
    pl.ksplit = 1;
    bool boundary_split = false;
    if (st.P)
    {
        if (have_pre)
        {
            // boundary of a hop whose partitions 1..P-1 were accumulated in the background: whatever slices are still
            // due, their total into slot 0, then partition 0 only
            if (!advance_background(blk, st, true)) return false;
            MacShape s0 = sh;
            s0.P = 1;
            // Partition 0 is a reduction over the inputs only — 64 terms on the 64 x 64 engine — and with ONE k-slice every wave walked
            // them all one after the other: 85 us per boundary, for 4 MB of spectra, twice in the call that carries the boundaries of
            // two stages.  Partition 0 is therefore split up to seven ways, its k-slices going into the slots BEHIND the background
            // slices' (Ypre has kBgSlices + kBoundarySlices of them), and the inverse adds all the slots up as it loads them — no launch
            // for the slices' total either.
            boundary_split = !is_big_fft(st.log2n) && nout_act == mCfg.nout;
            s0.max_ksplit = boundary_split ? kBoundarySlices : 1;
            mac_plan(s0, pl);
            if (!boundary_split) HCV_TRY(launch_reduce_partials(st.Ypre, st.bg_slices, (long long) mCfg.nout * st.M, (long long) mCfg.nout * st.M, sM));
            if (!mac(st, s0, pl, st.Ht(), boundary_split ? st.Ypre + (size_t) st.bg_slices * mCfg.nout * st.M : st.Y, h_first, check, sM)) return false;
        }
        else
        {
            mac_plan(sh, pl);
            if (!begin_event()) return false;
            if (!mac(st, sh, pl, st.Ht(), st.Y, h_first, check, sM)) return false;
            if (ev) HCV_TRY(hipEventRecord(ev->b, sM));
            st.launches++;
            st.hops += (uint64_t) T;
            st.last_ksplit = (uint32_t) pl.ksplit;
            st.last_ot = (uint32_t) pl.ot;
            st.last_tt = (uint32_t) pl.tt;
            st.last_parts = (uint32_t) p_live;
            if (!check && pl.nt) st.steady_launches++;
        }
    }
    if (tail_gate && sj == 0 && mStages.size() > 1 && !mOneStream && st.P && !defer && !have_pre)
    {
        HCV_TRY(rec(st.mac_done[q], sM));
        blk.gate = st.mac_done[q];
    }

    // ---- inverse phase (stream sI): every read-modify-write of this stage's timeline happens on this stream
    if (!blk.emit_first) HCV_TRY(wt(sI, mEvEmit[q]));       // emit(k-2) has cleared the timeline span reused now
    if (head_here)
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
            else if (boundary_split)
                HCV_TRY(launch_rifft_overlap_add(st.log2n, st.Ypre, st.bg_slices + pl.ksplit, y_elems, h_first, 1, (int) nout_act, st.timeline, st.tl_len,
                                                 st.tl_len - 1, st.tw, &st.big, sI));
            else
                HCV_TRY(launch_rifft_overlap_add(st.log2n, st.Y, 2, (long long) (st.Ypre - st.Y), h_first, 1, (int) nout_act, st.timeline, st.tl_len,
                                                 st.tl_len - 1, st.tw, &st.big, sI));
        }
        else
        {
            // the split-K partial sums are added up by the inverse transform's loads when that is cheaper than a launch of its
            // own: few slices, many transforms (each workgroup then reads ksplit spectra instead of one — c4: 6 x 64 KB per
            // output, 14 us less per step; c5's 24 slices over 16 outputs keep the reduction kernel, which uses the whole chip)
            const bool fold = pl.ksplit > 1 && pl.ksplit <= kFoldMax && !is_big_fft(st.log2n) && (long long) T * nout_act >= 16;
            if (!fold) HCV_TRY(launch_reduce_partials(st.Y, pl.ksplit, y_elems, y_elems, sI));
            HCV_TRY(launch_rifft_overlap_add(st.log2n, st.Y, fold ? pl.ksplit : 1, y_elems, h_first, T, (int) nout_act, st.timeline, st.tl_len, st.tl_len - 1,
                                             st.tw, &st.big, sI));
        }
    }
    HCV_TRY(rec(st.done[q], sI));
    // Every stage has one hop of latency: what this chain adds to the timeline starts at sample (h_first + 1) M.  When that is the
    // end of the block — the boundary calls of real-time sizes — this block's emit reads nothing of it (and clears a span the inverse
    // does not touch), so the main stream does not wait for the chain HERE: the call returns when its own samples are out, and the
    // chain runs on into the gap before the next call, whose enqueue starts with the wait (fence_chains; control work likewise).
    const bool late = !serial && !head_here && !mProfiling && (h_first + 1) * (long long) st.M >= n0 + (long long) B;
    if (late)
    {
        st.chain_pending = q;
        blk.late_mask |= 1u << (2 * si + (size_t) q);
    }
    else
        HCV_TRY(wt(mStream, st.done[q]));

    st.pre_hop = -1;
    if (defer)
    {
        // plan the background accumulation for hop h+1: partitions 1 .. min(P-1, h+1) (those that have input), in up to
        // kBgSlices launches that the following calls issue as the hop's samples arrive (advance_background)
        st.bg_parts = (int) std::min<long long>(st.P - 1, h_first + 1);
        st.bg_slices = bg_slice_count(st.bg_parts, st.M, B);
        st.bg_launched = 0;
        st.pre_hop = st.bg_parts > 0 ? h_first + 1 : -1;
    }
    return true;
}

// One block of at most max_block samples, everything device side.  Caller owns the engine (audio_enter).
//
// Stream plan for block k (q = k & 1; every event and the FIR output buffer exist twice, indexed by block parity):
//
//   in stream:    wait readers(k-2) ─ scatter_input ─ record in[q]
//   stage s:      wait in[q], emit[q] (= emit of block k-2) ─ rfft_frames → spectral_mac → reduce → rifft_overlap_add ─ record done_s[q]
//   head stream:  wait in[q], emit[q]                       ─ fir_head → tdout[q]                                   ─ record td[q]
//   main stream:  wait done_s[q] for all s, td[q] ─ emit(tdout[q]) ─ record emit[q]
//                 (small blocks: not for a stage whose chain only adds to samples BEHIND the block — `late`, enqueue_stage: that wait is
//                 taken at the next enqueue, fence_chains — and the emit is enqueued in front of those chains; a plain small call is
//                 one launch on the main stream, fir_head_small_kernel filing the samples, adding the head and emitting)
//
// Whole-hop blocks (made of whole, aligned hops of the last stage) run ONE stage on one stream instead:
//   last stage:   rfft_frames_direct -> spectral_mac over lead + P partitions -> [reduce_partials] -> rifft_emit
// (see enqueue_stage: no scatter, no head chain, no timeline, no emit).
//
// The stages only meet in emit(), so the latency-bound short stages and the FIR head hide under the HBM-bound tail;
// and because block k+1's scatter and FFTs do not wait for block k's emit, consecutive asynchronous calls overlap.
// Ring depths make that safe: the history ring holds three blocks + a frame (a block's readers must be done before
// the block two later is scattered over them), each stage timeline holds two blocks + a hop (emit(k-2) must have
// cleared what block k's hops are added into).
bool Engine::enqueue_chunk(const float *din, int64_t in_stride, float *dout, int64_t out_stride, uint32_t nin_act, uint32_t nout_act, uint32_t B)
{
    RoctxRange range("hcv:block");                                  // (HCV_ROCTX=1; otherwise one load of a flag)
    const long long n0 = mN;
    const long long hmask = mHistLen - 1;
    const uint32_t rows_in = mCfg.diag ? nout_act : nin_act;
    const int q = (int) (mBlockCount & 1);
    if (!fence_chains(/* keep_forward */ true)) return false;       // the main stream behind the boundary chains the previous block left running

    const bool drop_head = mDropHead.load(std::memory_order_relaxed);     // (reference-quirk mode: the head's output is lost, see hcv_engine.h)
    const bool td_any = mCfg.has_td && mTdLpad > 0 && !drop_head;
    const bool td_check = mTdMaxValid > 0 && (n0 - (long long) mTdLpad < mTdMaxValid);
    // Whole-hop mode: the block is made of whole, aligned hops of the last stage.  Every output sample of such a block only
    // needs inputs that the last stage's own frames hold, so IR[0 : its hop) is served by ONE extra zero-latency partition
    // of that stage (emitted in the hop's own slot, like the head-through-FFT of the first stage) and the head and the
    // shorter stages — a dozen launches in latency-bound chains — are not run at all.  Entering the mode drops their
    // pending (now duplicate) results; leaving it rebuilds their input spectra from the history ring and catches up on
    // the one hop whose result is due in the new block (see the stage loop).
    const size_t last = mStages.empty() ? 0 : mPivot;          // the stage whole-hop blocks run on
    const bool rungs = !mStages.empty() && mPivot + 1 < mStages.size();      // the extended ladder's far-tail stages run beside it
    const bool whole_hops = mTailHead && !drop_head && !td_check && (n0 % mStages[last]->M) == 0 && (B % mStages[last]->M) == 0 &&
                            !(mStages[last]->max_hv > n0 / mStages[last]->M);
    const bool entering = whole_hops && !mTailHeadPrev, leaving = !whole_hops && mTailHeadPrev;
    mTailHeadPrev = whole_hops;
    // hop-aligned block of a larger matrix: the head goes through the first stage's FFTs (see init)
    // (not for a block of ONE hop of the first stage: the head through the FFTs is emitted in the hop's own slot, so this block's emit
    // would wait for the stage's whole chain, where beside the time-domain head every chain of a real-time call runs on past the
    // emit — enqueue_stage, `late`.  ns64 at 128 samples per call: p50 0.110 -> 0.075 ms, p99 0.273 -> 0.190; at 256 samples, two
    // hops, the FFT head is ahead, p99 0.29 against 0.31)
    const bool head_fft = !whole_hops && td_any && mHeadFFT && !td_check && (n0 % mStages[0]->M) == 0 && (B % mStages[0]->M) == 0 && B >= 2 * mStages[0]->M;
    const bool td = td_any && !head_fft && !whole_hops;

    // Serial blocks: everything on the main stream, in program order, with no events at all.  A dependency on a pending event
    // of another stream costs the host ~10 us (a kernel launch 2.6, an event record 1.7 — tools/micro/api_cost.hip), so a
    // small engine running one stage per block (whole-hop mode: 26 calls, five such dependencies, 111 us of host time for
    // 60 us of kernels on the 8x1 workload) is bound by its own enqueue; big engines keep their streams (but see below).  HCV_SERIAL = 0 / 1
    // forces the choice for whole-hop blocks; single-stream engines are always serial.
    static const int serial_env = std::getenv("HCV_SERIAL") ? std::atoi(std::getenv("HCV_SERIAL")) : -1;
    constexpr double kSerialMB = 1024.0;
    const bool small_tail = whole_hops && (double) mStages[last]->live_parts * mStages[last]->M * sizeof(float2) < kSerialMB * 1048576.0;
    // (an extended ladder keeps its streams: the rungs' deferred slices run beside the pivot stage's latency-bound chain — c5 on the ladder
    // 0.147 ms per 8192-sample block on one stream, 0.135 on the stages' own)
    const bool serial_small = mOneStream || (whole_hops && (serial_env >= 0 ? serial_env != 0 : (small_tail && !rungs)));
    // ... and big engines keep it for the blocks that do not take the n x m fused block (below): a steady-state one-hop block of a streamed
    // engine through device pointers does take it, on the main stream like a small engine's (round 6, one box, streamed -> this: c5 1.954 ->
    // 1.909 ms per step, ns64 2.470 -> 2.403, c4 0.524 -> 0.504 — the forward transforms run under the previous block's inverse and the
    // partial spectra meet in LDS; in round 5 the block gained nothing there).  Not where a host-pointer call has put the old partitions'
    // multiply-accumulate on the stage's stream already (host_pre_mac).
    const bool pre_pending = mPre.valid && mPre.block == mBlockCount;
    const bool big_candidate = whole_hops && serial_env < 0 && !serial_small && !rungs && !pre_pending;
    const bool full_matrix_c = nout_act == mCfg.nout && nin_act == (mCfg.diag ? mCfg.nout : mCfg.nin);
    // (direct input / output of whole-hop blocks: see below where the block's fields are set)
    const bool direct_in_c = rows_in > 0 && whole_hops && !is_big_fft(mStages[last]->log2n) && ((uintptr_t) din % 8) == 0 && (in_stride % 2) == 0 && !entering;
    const bool direct_out_c = whole_hops && !entering && !rungs && ((uintptr_t) dout % 8) == 0 && (out_stride % 2) == 0;
    bool nxm_take = false;
    FusedNxmPlan nxm_plan_c;
    // The n x m fused block (hcv_fused_nxm.hip): a steady-state one-hop block of a serial engine with several outputs whose last stage
    // is the reference's 16384-point tail with a lead slot — one rank's share of a strong-scaled matrix (64 x 8 of config 4 over 8 GPUs)
    // is the shape it was built for.  Forward transforms on the pipe stream, ONE multiply-accumulate + inverse launch on the main
    // stream, no event between them (HCV_COOP = 0: the separate kernels).
    // Streamed engines (a GB and more of tail spectra) take it too since round 6 (`big_candidate` above; in round 5, before the forward launch
    // was placed under the previous block's inverse, 64 x 64 gained nothing with 2 s IRs — 0.553 -> 0.551 ms per step — and lost 3 % with 10 s).
    // The pivot stage of an extended ladder can take it (HCV_NXM_LADDER = 1; its hop then goes into the stage's timeline, which emit adds to
    // the rungs') but does not by default.  Measured, c5 on the ladder, ms per step: ONE lane 0.147 / 0.153 with it against 0.137 / 0.140
    // without (one workgroup per CU with most of its registers crowds the rungs' slices out while it runs); TWO lanes (enqueue_stage) as the
    // process's only engine 0.121 - 0.129 with it against 0.125 - 0.131 without — inside the boxes' spread — and as the process's SECOND engine
    // (bench.py's extended leg) 0.80 with it: its forward launches sat behind the first engine's streams in a hardware queue they share, every
    // multiply-accumulate waited its full bound and then did the transforms itself.  The stand-down below brings that to 0.148; the two
    // lanes alone run 0.118 - 0.125 there (0.60 - 0.64 of HBM for the ladder's 601 MB per step), so that is the default.
    static const int nxm_ladder_env = std::getenv("HCV_NXM_LADDER") ? std::atoi(std::getenv("HCV_NXM_LADDER")) : 0;
    const bool nxm_ladder = nxm_ladder_env != 0;
    if ((serial_small || big_candidate || (rungs && nxm_ladder)) && whole_hops && direct_in_c && (direct_out_c || (rungs && !entering)) && !mCfg.diag && mCfg.nout > 1 &&
        full_matrix_c && mPipeStream && B == mStages[last]->M)
    {
        const Stage &tl = *mStages[last];
        const long long h = n0 / (long long) tl.M;
        const int Pw = (int) (tl.P + tl.lead);
        const bool wcheck = (h - tl.max_hv) < (long long) Pw - 1;           // (right after a reset the partitions have bounds: the checked kernels)
        // (launches whose wait for the forward transforms ran out report it, hcv_fused_nxm.hip: three of them within 64 blocks and the stage
        // takes the separate kernels for the next 64 .. 4096 blocks — the forward stream is stuck behind another stream in a hardware queue they
        // share, e.g. a second engine of the process: c5 on the ladder as the bench's second engine ran 0.80 ms per step that way, 0.125 without)
        Stage &tw_ = *mStages[last];
        if (tw_.nxm_helped)
        {
            const unsigned now = *reinterpret_cast<volatile unsigned *>(tw_.nxm_helped);
            if (now != tw_.nxm_helped_seen)
            {
                tw_.nxm_helped_seen = now;
                tw_.nxm_strikes = (mBlockCount - tw_.nxm_strike_block <= 64) ? tw_.nxm_strikes + 1 : 1;
                tw_.nxm_strike_block = mBlockCount;
                if (tw_.nxm_strikes >= 3)
                {
                    // (round 6: the stand-down backs off — 64 blocks the first time, four times longer each time it recurs within 1024 blocks of
                    // coming back, 4096 at most — instead of 4096 blocks for any three late launches: a co-tenant's burst costs half a second of
                    // the separate kernels, not 34)
                    if (mBlockCount - tw_.nxm_off_until > 1024) tw_.nxm_backoff = 64;
                    tw_.nxm_off_until = mBlockCount + tw_.nxm_backoff;
                    tw_.nxm_backoff = std::min<uint64_t>(4096, tw_.nxm_backoff * 4);
                    tw_.nxm_stood_down++;
                    tw_.nxm_strikes = 0;
                }
            }
        }
        if (tl.lead && tl.coop_flags && !tl.coop_off && !tl.gh_count && !wcheck && h + 1 >= Pw && mBlockCount >= tl.nxm_off_until)
            nxm_take = fused_block_nxm_plan(tl.log2n, (int) rows_in, (int) nout_act, Pw, tl.y_elems, &nxm_plan_c);
    }
    const bool serial = serial_small || (big_candidate && nxm_take);
    Block blk;
    blk.nxm = nxm_take && (serial || rungs);
    blk.nxm_plan = nxm_plan_c;
    blk.din = din; blk.dout = dout; blk.in_stride = in_stride; blk.out_stride = out_stride;
    blk.nin_act = nin_act; blk.nout_act = nout_act; blk.rows_in = rows_in; blk.B = B;
    blk.n0 = n0; blk.hmask = hmask; blk.q = q; blk.last = last;
    blk.td_any = td_any; blk.td_check = td_check; blk.whole_hops = whole_hops; blk.entering = entering; blk.leaving = leaving;
    blk.head_fft = head_fft; blk.td = td; blk.serial = serial;
    blk.full_matrix = nout_act == mCfg.nout && nin_act == (mCfg.diag ? mCfg.nout : mCfg.nin);
    blk.main = mStream; blk.sIn = serial ? mStream : mInStream; blk.sTd = serial ? mStream : mTdStream;
    blk.src.count = 0;
    const hipStream_t sIn = blk.sIn, sTd = blk.sTd;
    auto rec = [&](hipEvent_t e, hipStream_t s) { return blk.rec(e, s); };
    auto wt = [&](hipStream_t s, hipEvent_t e) { return blk.wt(s, e); };
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
        for (Stage *st : mStages)
        {
            HCV_TRY(hipStreamWaitEvent(st->stream, mEvSerial, 0));
            if (st->stream2) HCV_TRY(hipStreamWaitEvent(st->stream2, mEvSerial, 0));
        }
    }
    mPrevSerial = serial;
    if (mGhostPruneAt >= 0 && n0 >= mGhostPruneAt && !prune_ghosts()) return false;
    // a foreign event the main stream waits for (process_dev `after`) binds a streamed block's stage streams like control work;
    // a serial block runs on the main stream and is behind it already
    if (mExtDirty && !serial) mCtlDirty = true;
    mExtDirty = false;
    // control work queued on the main stream (IR spectra, reset fills, regrown buffers) must land before this block
    const bool ctl_was_dirty = mCtlDirty;
    if (mCtlDirty)
    {
        HCV_TRY(rec(mEvCtl, mStream));
        HCV_TRY(wt(sIn, mEvCtl));
        // (deferred slices are launched without waiting for the block's input: order them after the control work directly)
        for (Stage *st : mStages)
        {
            HCV_TRY(wt(st->stream, mEvCtl));
            if (st->stream2) HCV_TRY(wt(st->stream2, mEvCtl));
        }
        mCtlDirty = false;
    }
    // Direct input: when exactly one FFT stage runs this block and the block is made of whole,
    // aligned hops of it — whole-hop mode, or a single-stage engine — that stage's forward transforms read the caller's block
    // in their first pass and file it in the history ring themselves (launch_rfft_frames_direct).  The scatter launch goes,
    // and with it the cross-stream hand-over in front of the stage's first kernel: on a 64x64 engine with 2 s IRs the next
    // block's scatter used to sit in a hardware queue behind the tail MAC (there are fewer queues than streams), 25-40 us of
    // every 0.58 ms step.  Needs 8-byte aligned input rows (the first pass loads sample pairs).
    const bool direct_in = rows_in > 0 && whole_hops && !is_big_fft(mStages[last]->log2n) && ((uintptr_t) din % 8) == 0 &&
                           (in_stride % 2) == 0 && !entering;
    blk.direct_in = direct_in;
    // Direct output: a whole-hop block past the mode's first has empty timelines and one
    // zero-latency transform per output and hop, so the inverse's last pass writes the caller's block (launch_rifft_emit) and the
    // emit launch — with its wait for the stage's stream — goes too.  Needs 8-byte aligned output rows.
    blk.direct_out = whole_hops && !entering && !rungs && ((uintptr_t) dout % 8) == 0 && (out_stride % 2) == 0;
    if (blk.nxm)
    {
        if (!mPrevNxm || ctl_was_dirty)
        {
            // the pipe stream starts behind everything the main stream holds so far (earlier blocks, control work)
            HCV_TRY(hipEventRecord(mEvSerial, mStream));
            HCV_TRY(hipStreamWaitEvent(mPipeStream, mEvSerial, 0));
            mNxmRun = 0;
        }
        // Back-pressure.  Nothing else holds the forward launches back — an asynchronous caller's whole burst of them would run at once —
        // and the launch of block k overwrites the ring slot of hop h - R, which the multiply-accumulate of block k - (R - P - lead) - 1
        // still reads, and the history ring's samples of hop h - hops, which the helping path of block k - hops + 1 may still read: block
        // k - lag must be through, lag = min(R - P - lead + 1, hops - 1) (R = Pcap + 2 Tmax + 4: seven with 8192-sample blocks).  An event
        // record costs the recording stream 4 - 7 us, so only every (lag - 1)-th block records its end on the main stream, and block k's
        // forward launch waits for the one such end in [k - lag, k - 2]: never for the launch that needs it, nor for the one in front of
        // that (the forward transforms of block k run beside block k - 1's inverse), and in a paced or GPU-bound stream for one long gone.
        {
            const Stage &tl = *mStages[last];
            const long long lag = std::min<long long>((long long) tl.R - (long long) (tl.P + tl.lead) + 1, mHistLen / (long long) tl.M - 1);
            // (two lanes, enqueue_stage: the end recorded on one lane says nothing of the other lane's newer blocks — only, through the emits
            // both lanes' inverses wait for, that every block four and more back is through: the window is four blocks shorter)
            mNxmEvery = (uint32_t) std::max<long long>(1, std::min<long long>(8, lag - 1 - (tl.stream2 ? 4 : 0)));
            if (lag < 2) blk.nxm = false;           // (no room to run ahead at all: the separate kernels)
            else if ((long long) mNxmRun >= lag) HCV_TRY(hipStreamWaitEvent(mPipeStream, mEvNxmEnd[((mNxmRun - 2) / mNxmEvery) & 3], 0));
        }
    }
    if (!blk.nxm && mFwdPending)
    {
        // a block of another kind: everything behind the forward launches of the fused blocks before it (Engine::join_forward_stream)
        if (!join_forward_stream()) return false;
        if (!serial)
        {
            HCV_TRY(hipStreamWaitEvent(mInStream, mEvFwd, 0));
            HCV_TRY(hipStreamWaitEvent(mTdStream, mEvFwd, 0));
            for (Stage *st : mStages)
            {
                HCV_TRY(hipStreamWaitEvent(st->stream, mEvFwd, 0));
                if (st->stream2) HCV_TRY(hipStreamWaitEvent(st->stream2, mEvFwd, 0));
            }
        }
    }
    // A PLAIN small call — it completes no hop of any stage (three calls in four at 32 samples per call) — has nothing to wait
    // for but its head: the head kernel then delivers the block itself, on the main stream (the stages' timelines added and
    // cleared as emit would: launch_fir_head's `emit`), AND files the call's samples in the history ring: the whole call is one launch
    // (scatter || head -> emit, then scatter || head + emit, now head + scatter + emit).
    const bool head_direct = td && fir_head_is_small((int) B, (int) nin_act, (int) mTdLpad, mCfg.diag ? 1 : 0) && !direct_in;
    bool plain = head_direct && !leaving && !blk.direct_out && !serial && rows_in > 0;
    for (size_t si = 0; plain && si < mStages.size(); si++) plain = (n0 + B) / mStages[si]->M == n0 / mStages[si]->M;
    if (!direct_in)
    {
        const hipStream_t sW = plain ? mStream : sIn;           // (who writes the ring)
        // the block two back read the history this scatter may overwrite
        for (Stage *st : mStages) HCV_TRY(wt(sW, st->done[q]));
        HCV_TRY(wt(sW, mEvTd[q]));
        // after direct-input blocks the ring was last written on the last stage's stream: later hops must land behind those
        if (mPrevDirect && !serial) HCV_TRY(wt(sW, mStages[last]->done[q ^ 1]));
        if (plain)
            HCV_TRY(wt(mStream, mEvInput[q ^ 1]));              // (the previous block's samples are filed in front of these)
        else
        {
            // a plain block before this one filed ITS samples from the main stream (fir_head_small_kernel): this scatter — and, through
            // the input event it records, every transform of this block that reads the ring — goes behind that write.  (Blocks that
            // scatter share the input stream, whose order covered it until the plain call took the scatter over.)
            if (mPrevPlain) HCV_TRY(wt(sIn, mEvInput[q ^ 1]));
            HCV_TRY(launch_scatter_input(din, in_stride, (int) B, (int) rows_in, mHist, mHistLen, hmask, n0, sIn));
            HCV_TRY(rec(mEvInput[q], sIn));
        }
    }
    mPrevDirect = direct_in;
    mPrevPlain = plain;
    // Two-stream pipeline of a small engine's whole-hop blocks (enqueue_stage): the NEXT block's forward transforms on a second stream
    // beside the current block's MAC, reduction and inverse.  A cross-stream hand-over is dear on this stack — the wait on the
    // transforms' event plus the end record cost the main stream about 10 us per block, an event record alone 4-8 us — so it pays only
    // where both halves of the chain are longer than that, and the engine records as few events as the ring's margin allows (in a run
    // of single-hop blocks every second block, enqueue_stage).  Measured with the steps timed bare (the MAC's HIP events in the chain
    // had hidden the gain in the first measurement), ms per 8192-sample block, serial -> pipelined:
    //   c3 (8 -> 1; transforms 13.5 us | MAC 7.3 + reduction 4.2 + inverse 8.5)   0.0335 -> 0.0270
    //   c1 (1 x 1, 16384-point stage; 13 | 10)                                     0.0233 -> 0.0212
    //   c2 (1 x 1, 4096-point stage, four hops per block; 9 | 15)                  0.0228 -> 0.0355  (loses: multi-hop blocks keep the
    //                                                                              closer wait and an end record per block)
    // Hence the rule: single-workgroup transforms of 16384 points and up.  Only asynchronous callers gain — a call that waits for its
    // block pays the hand-overs and overlaps nothing (c3 synchronous steps 0.044 -> 0.057 when forced).  HCV_PIPE2 = 0 / 1 forces
    // the choice.
    static const int pipe2_env = std::getenv("HCV_PIPE2") ? std::atoi(std::getenv("HCV_PIPE2")) : -1;
    bool want_pipe2 = pipe2_env > 0;
    if (pipe2_env < 0 && !mCallWaits && !mStages.empty())
    {
        // (not where the residue-split transforms run, hcv_fft_split.hip: spread over 9 CUs instead of one the transforms are no
        // longer the long part of the chain, and the serial chain beats the hand-overs — c1 0.0179 against 0.0200 ms per block)
        const Stage &tl = *mStages[last];
        const int hops = (int) (B / tl.M);
        want_pipe2 = tl.log2n >= 14 && !fft_split_applies(tl.log2n, hops * (int) rows_in);
    }
    blk.pipe2 = want_pipe2 && serial && whole_hops && direct_in && mPipeStream != nullptr && !blk.nxm;
    if (blk.pipe2)
    {
        if (!mPrevPipe2 || ctl_was_dirty)
        {
            // the pipe stream starts behind everything the main stream holds so far (earlier blocks, control work)
            HCV_TRY(hipEventRecord(mEvSerial, mStream));
            HCV_TRY(hipStreamWaitEvent(mPipeStream, mEvSerial, 0));
            mPipeRun = 0;
            mPipeSince = 0;
            mPipeRecA = mPipeRecB = -1;
        }
    }
    else
    {
        mPipeRun = mPipeSince = 0;
        mPipeRecA = mPipeRecB = -1;
    }
    mPrevPipe2 = blk.pipe2;

    if (td)
    {
        // small calls: the head reads the call's own samples from the caller's block, so it starts beside the scatter instead of behind
        // it (one cross-stream hand-over less in the chain scatter -> head -> emit of a plain real-time call) and only needs the ring
        // complete up to the call's first sample — the previous block's input event
        if (plain)
        {
            EmitSources all;
            all.count = 0;
            for (Stage *st : mStages)
            {
                all.timeline[all.count] = st->timeline;
                all.stride[all.count] = st->tl_len;
                all.mask[all.count] = st->tl_len - 1;
                all.count++;
            }
            // (the main stream is behind the previous block's emit and this call's control work already; the ring up to the call's
            // first sample is the previous block's input event)
            HCV_TRY(launch_fir_head(mHist, mHistLen, hmask, mTaps, (int) mTdLpad, 2048, (int) nin_act, (int) mNinAlloc, (int) nout_act, 0, n0, (int) B,
                                    mTdValid, td_check, dout, out_stride, mStream, din, in_stride, &all, mHist));
            HCV_TRY(rec(mEvInput[q], mStream));         // "the block's input is in the ring"
            HCV_TRY(rec(mEvTd[q], mStream));
            blk.emitted = true;
        }
        else
        {
            HCV_TRY(wt(sTd, mEvInput[head_direct ? (q ^ 1) : q]));
            // (the previous block's input event is older than this call's control work — new taps, reset fences, an upload of the
            // block itself, a foreign `after` event — which this block's own input event would have put in front of the head)
            if (head_direct && ctl_was_dirty && sTd != mStream) HCV_TRY(wt(sTd, mEvCtl));
            HCV_TRY(wt(sTd, mEvEmit[q]));      // emit(k-2) has consumed tdout[q]
            const bool check = td_check;
            HCV_TRY(launch_fir_head(mHist, mHistLen, hmask, mTaps, (int) mTdLpad, 2048, (int) nin_act, (int) mNinAlloc, (int) nout_act, mCfg.diag ? 1 : 0,
                                    n0, (int) B, mTdValid, check, mTdOut[q], mMaxBlock, sTd, head_direct ? din : nullptr, in_stride));
            HCV_TRY(rec(mEvTd[q], sTd));
        }
    }

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
    blk.tail_gate = tail_gate;


    // XCD pinning (hcv_kernels.h: xcd_pin_for): a serial whole-hop block whose live spectra and input ring fit ONE XCD's L2 with
    // room to spare keeps its three or four tiny launches on one XCD, so that each kernel finds its predecessor's output in that
    // L2 instead of fetching it from the memory side (c1: 0.0182 -> 0.0169 ms per block; c2, 3.85 MB of spectra, loses 10 %)
    const bool pin_block = serial && whole_hops && !mStages.empty() &&
                           (double) (mStages[last]->live_parts + (uint64_t) rows_in * mStages[last]->R) * mStages[last]->M * sizeof(float2) <= 1.5 * 1048576.0;
    xcd_pin_hint(pin_block, mPinXcd);
    // largest stage first: the tail's spectral_mac is the critical path, the short stages fill in around it.  (A whole-hop block of
    // an extended ladder: the pivot stage first — its transforms file the block's samples in the history ring, which a rung at
    // its hop boundary reads — then the rungs.)
    // A small streamed block enqueues its emit IN FRONT of the chains it does not wait for (enqueue_stage: `late`) and of the deferred
    // slices: there are fewer hardware queues than streams, and a main stream that shares one with a stage's stream had its emit
    // queued behind that stage's chain — calls with the boundaries of two or three stages were done after 0.10-0.125 ms where their
    // enqueue took 0.05-0.06.  What emit needs of such a block is decided here: the stages' timelines, and the stages whose chain's
    // output starts inside the block (the head through the first stage's FFTs) in front of it.
    blk.emit_first = !whole_hops && !serial && !leaving && !mProfiling && B <= 512 && !blk.direct_out && !mStages.empty();
    auto after_emit = [&](size_t si) -> bool
    {
        const Stage &sg = *mStages[si];
        if (head_fft && si == 0) return false;
        const long long h0 = n0 / (long long) sg.M;
        const int hops = (int) ((n0 + B) / sg.M - h0);
        return hops <= 0 || (h0 + 1) * (long long) sg.M >= n0 + (long long) B;
    };
    if (blk.emit_first)
        for (size_t sj = 0; sj < mStages.size(); sj++)
        {
            Stage &sg = *mStages[mStages.size() - 1 - sj];
            blk.src.timeline[blk.src.count] = sg.timeline;
            blk.src.stride[blk.src.count] = sg.tl_len;
            blk.src.mask[blk.src.count] = sg.tl_len - 1;
            blk.src.count++;
            HCV_TRY(wt(sg.stream, mEvEmit[q]));     // emit(k-2) has cleared the timeline span a chain of this block may reuse (before this
                                                    // block's emit takes the event over)
        }
    for (int pass = 0; pass < 2; pass++)
    {
        for (size_t sj = 0; sj < mStages.size(); sj++)
        {
            size_t si = mStages.size() - 1 - sj;
            if (whole_hops && rungs) si = sj == 0 ? last : (sj <= mStages.size() - 1 - last ? mStages.size() - sj : mStages.size() - 1 - sj);
            if ((blk.emit_first && after_emit(si)) != (pass == 1)) continue;
            if (!enqueue_stage(blk, si, sj))
            {
                xcd_pin_hint(false);
                return false;
            }
        }
        if (pass == 1) break;
        if (blk.emitted)
        {
            // (the head kernel delivered the block; the main stream still ends behind the scatter)
            HCV_TRY(wt(mStream, mEvInput[q]));
            HCV_TRY(rec(mEvEmit[q], mStream));
        }
        else if (!blk.direct_out)
        {
            if (td) HCV_TRY(wt(mStream, mEvTd[q]));
            HCV_TRY(wt(mStream, mEvInput[q]));       // a block with no live stage still orders after its scatter
            HCV_TRY(launch_emit(blk.src, n0, (int) B, (int) nout_act, td ? mTdOut[q] : nullptr, mMaxBlock, dout, out_stride, mStream));
            HCV_TRY(rec(mEvEmit[q], mStream));
        }
        if (!blk.emit_first) break;
    }
    xcd_pin_hint(false);
    mEmitFirstEv = (blk.emit_first && !blk.direct_out) ? mEvEmit[q] : nullptr;
    mN += B;
    mBlockCount++;
    mLastNin = rows_in;
    mLastNout = nout_act;
    return true;
}

// Host-pointer calls, one whole hop of a streamed engine (see PreMac in hcv_engine.h).  The block about to be enqueued is classified by the
// same rules enqueue_chunk and enqueue_stage apply (the engine is owned: nothing changes in between; should the two ever disagree, the
// block's own enqueue computes the whole sum again and this launch was wasted, not wrong).  It qualifies when it is a steady-state whole-hop
// block of the full matrix on a lead-slot stage with a GB or more of live spectra, one hop long, with nothing pending that the
// multiply-accumulate would have to be ordered behind.  Then  Y[0 .. ks) = sum over p >= 1  X[h - p] H[p]  goes out on the stage's stream now.
bool Engine::host_pre_mac(uint32_t nin_act, uint32_t nout_act, uint32_t B)
{
    mPre.valid = false;
    static const bool allow = !(std::getenv("HCV_HOST_PRE_MAC") && std::atoi(std::getenv("HCV_HOST_PRE_MAC")) == 0);
    static const int serial_env = std::getenv("HCV_SERIAL") ? std::atoi(std::getenv("HCV_SERIAL")) : -1;
    if (!allow || mStages.empty() || !mTailHead || !mLeadSlot || mOneStream || mProfiling || mCfg.diag) return true;
    if (mCtlDirty || mExtDirty || mFwdPending || mDropHead.load(std::memory_order_relaxed)) return true;
    Stage &st = *mStages[mPivot];
    const bool rungs = mPivot + 1 < mStages.size();
    const long long n0 = mN;
    if (rungs || !st.lead || B != st.M || (n0 % st.M) != 0 || !mTailHeadPrev) return true;                  // one whole hop, not the mode's first block
    if (mTdMaxValid > 0 && (n0 - (long long) mTdLpad < mTdMaxValid)) return true;                             // (td_check)
    if (st.max_hv > n0 / st.M) return true;
    if (nout_act != mCfg.nout || nin_act != mCfg.nin) return true;                                           // the full matrix
    if ((double) st.live_parts * st.M * sizeof(float2) < 1024.0 * 1048576.0 || serial_env == 1) return true; // streamed engines only
    for (Stage *o : mStages)
        if (o->chain_pending >= 0 || o->bg_pending) return true;
    const int Pw = (int) (st.P + st.lead);
    const long long h_mac = n0 / st.M;
    if (Pw < 4 || st.gh_count || (h_mac - st.max_hv) < (long long) Pw - 1 || h_mac + 1 < Pw) return true;     // steady state: every partition live
    const int q = (int) (mBlockCount & 1);
    const size_t ks_cap = std::max<size_t>(1, st.y_elems / ((size_t) nout_act * st.M));
    if (ks_cap < kPreNewSlices + 2) return true;
    MacShape so = mac_shape(st, /* P */ Pw - 1, /* Pcap */ st.hparts(), /* nin */ (int) nin_act, /* nin_alloc */ (int) mNinAlloc, /* nout */ (int) nout_act, /* diag */ 0,
                            /* T */ 1, /* max_ksplit */ (int) (ks_cap - kPreNewSlices));
    MacPlan po;
    mac_plan(so, po);
    if (po.mfma || po.inwg) return true;
    xcd_pin_hint(false);
    if (!mac(st, so, po, st.Hs + st.M, st.Yq[q], h_mac - 1, /* check */ false, st.stream)) return false;
    mPre.valid = true;
    mPre.block = mBlockCount;
    mPre.ksplit = po.ksplit;
    mPre.h_mac = h_mac;
    mPre.nin = nin_act;
    mPre.nout = nout_act;
    return true;
}

// The staging buffer of the host paths (mDevIn) is ONE buffer, rewritten by every call's upload.  The forward launches of n x m blocks read it
// from the pipe stream with no event towards the main stream (hcv_fused_nxm.hip): one that arrives late — its block finished by the helping
// path, the call returned — would otherwise still be reading while the next call's upload lands (the launch itself also stands
// down once its tasks are marked done).  The upload goes behind the pipe stream: two calls, only while such launches are outstanding.
bool Engine::input_behind_forward()
{
    if (!mFwdPending) return true;
    HCV_TRY(hipEventRecord(mEvFwd, mPipeStream));
    HCV_TRY(hipStreamWaitEvent(mStream, mEvFwd, 0));
    return true;
}

// Host-pointer path, first half: stage the inputs, enqueue the block and the download of its result.  Small blocks (the
// real-time sizes) skip both DMA copies: scatter_input reads the pinned staging buffer through its device mapping and emit
// writes the pinned output buffer directly — two copy-engine round trips less per call.
bool Engine::process_begin(const float *const *ins, uint32_t nin_act, uint32_t nout_act, uint32_t B)
{
    DeviceGuard dg(mDevice);
    nout_act = std::min(nout_act, mCfg.nout);
    nin_act = std::min(nin_act, mCfg.nin);
    mHostMuted = false;
    if (!nout_act || !B) return true;
    if (B > mMaxBlock) { mErr = "process_begin: block longer than max_block"; return false; }
    const uint32_t rows_in = mCfg.diag ? nout_act : nin_act;
    if (!audio_enter(B))
    {
        mHostMuted = true;              // (a stream-start collision, hcv_engine.h: this block is silent, nothing of the engine's state is touched)
        return true;
    }
    OwnerGuard own(this);
    if (!fence_chains(/* keep_forward */ true) || !update_active_matrix(rows_in, nout_act) || !apply_pending_resets()) return false;
    static const int zc_limit = std::getenv("HCV_ZERO_COPY") ? std::atoi(std::getenv("HCV_ZERO_COPY")) : 2048;
    const bool zero_copy = mPinInDev && mPinOutDev && (int) B <= zc_limit;
    // (a hop-sized block of a streamed engine: what needs nothing of the caller's samples starts now, beside the staging copy and the upload)
    if (!zero_copy && !host_pre_mac(nin_act, nout_act, B)) return false;
    for (uint32_t i = 0; i < rows_in; i++) std::memcpy(mPinIn + (size_t) i * B, ins[i], sizeof(float) * B);
    if (zero_copy)
    {
        mCallWaits = true;
        if (!enqueue_chunk(mPinInDev, B, mPinOutDev, B, nin_act, nout_act, B)) return false;
    }
    else
    {
        // the upload goes on the main stream and is handed to the block like control work: a serial block scatters on the
        // main stream, a streamed one makes its input stream wait for it (enqueue_chunk, mCtlDirty)
        if (rows_in)
        {
            if (!input_behind_forward()) return false;
            HCV_TRY(hipMemcpyAsync(mDevIn, mPinIn, sizeof(float) * rows_in * B, hipMemcpyHostToDevice, mStream));
            mCtlDirty = true;
        }
        mCallWaits = true;
        if (!enqueue_chunk(mDevIn, B, mDevOut, B, nin_act, nout_act, B)) return false;
        HCV_TRY(hipMemcpyAsync(mPinOut, mDevOut, sizeof(float) * nout_act * B, hipMemcpyDeviceToHost, mStream));
    }
    // (a zero-copy block that enqueued its emit in front of its late chains is waited for through that emit's own event: an event
    // recorded NOW goes to the end of the main stream's hardware queue, behind the chains of whichever stage's stream shares it —
    // host-pointer calls carrying two or three boundaries took 0.14 ms where device-pointer calls took 0.07)
    if (zero_copy && mEmitFirstEv) mHostWait = mEmitFirstEv;
    else
    {
        HCV_TRY(hipEventRecord(mEvHostDone, mStream));
        mHostWait = mEvHostDone;
    }
    own.leave();                        // (a section posted meanwhile, the clock, the ownership back)
    return true;
}

// second half: wait for THIS block's result (an event, not the stream) and deliver it
bool Engine::process_end(float *const *outs, uint32_t nout_act, uint32_t B, bool accumulate)
{
    nout_act = std::min(nout_act, mCfg.nout);
    if (!nout_act || !B) return true;
    if (mHostMuted)
    {
        if (!accumulate)
            for (uint32_t o = 0; o < nout_act; o++) std::memset(outs[o], 0, sizeof(float) * B);
        return true;
    }
    DeviceGuard dg(mDevice);
    HCV_TRY(hipEventSynchronize(mHostWait ? mHostWait : mEvHostDone));
    for (uint32_t o = 0; o < nout_act; o++)
    {
        float *dst = outs[o];
        const float *src = mPinOut + (size_t) o * B;
        if (accumulate)
            for (uint32_t j = 0; j < B; j++) dst[j] += src[j];
        else
            std::memcpy(dst, src, sizeof(float) * B);
    }
    return true;
}

bool Engine::process(const float *const *ins, float *const *outs, uint32_t nin_act, uint32_t nout_act, uint64_t n, bool accumulate)
{
    nout_act = std::min(nout_act, mCfg.nout);
    nin_act = std::min(nin_act, mCfg.nin);
    if (!nout_act || !n) return true;
    const uint32_t rows_in = mCfg.diag ? nout_act : nin_act;
    std::vector<const float *> ip(std::max<uint32_t>(rows_in, 1));
    std::vector<float *> op(nout_act);
    for (uint64_t pos = 0; pos < n; pos += mMaxBlock)
    {
        const uint32_t B = (uint32_t) std::min<uint64_t>(mMaxBlock, n - pos);
        for (uint32_t i = 0; i < rows_in; i++) ip[i] = ins[i] + pos;
        for (uint32_t o = 0; o < nout_act; o++) op[o] = outs[o] + pos;
        if (!process_begin(ip.data(), nin_act, nout_act, B) || !process_end(op.data(), nout_act, B, accumulate)) return false;
    }
    if (mProfiling) collect_events();
    return true;
}

bool Engine::process_pinned(const float *ins_host, const float *ins_map, int64_t in_stride, float *outs_host, float *outs_map, int64_t out_stride,
                            uint32_t nin_act, uint32_t nout_act, uint64_t n)
{
    static const int zc_limit = std::getenv("HCV_ZERO_COPY") ? std::atoi(std::getenv("HCV_ZERO_COPY")) : 2048;
    if (n <= (uint64_t) zc_limit) return process_dev(ins_map, in_stride, outs_map, out_stride, nin_act, nout_act, n, true);
    DeviceGuard dg(mDevice);
    nout_act = std::min(nout_act, mCfg.nout);
    nin_act = std::min(nin_act, mCfg.nin);
    if (!nout_act || !n) return true;
    const uint32_t rows_in = mCfg.diag ? nout_act : nin_act;
    for (uint64_t pos = 0; pos < n; pos += mMaxBlock)
    {
        const uint32_t B = (uint32_t) std::min<uint64_t>(mMaxBlock, n - pos);
        if (!audio_enter(n))
        {
            for (uint32_t o = 0; o < nout_act; o++) std::memset(outs_host + (size_t) o * out_stride + pos, 0, sizeof(float) * B);
            continue;
        }
        OwnerGuard own(this);
        if (!fence_chains(/* keep_forward */ true) || !update_active_matrix(rows_in, nout_act) || !apply_pending_resets()) return false;
        if (!host_pre_mac(nin_act, nout_act, B)) return false;
        if (rows_in)
        {
            if (!input_behind_forward()) return false;
            // (a packed block is ONE transfer; a pitched one is moved row by row by the copy engine, a few microseconds per row)
            if (in_stride == (int64_t) B)
                HCV_TRY(hipMemcpyAsync(mDevIn, ins_host + pos, sizeof(float) * (size_t) B * rows_in, hipMemcpyHostToDevice, mStream));
            else
                HCV_TRY(hipMemcpy2DAsync(mDevIn, sizeof(float) * B, ins_host + pos, sizeof(float) * (size_t) in_stride, sizeof(float) * B, rows_in,
                                         hipMemcpyHostToDevice, mStream));
            mCtlDirty = true;
        }
        mCallWaits = true;
        if (!enqueue_chunk(mDevIn, B, mDevOut, B, nin_act, nout_act, B)) return false;
        if (out_stride == (int64_t) B)
            HCV_TRY(hipMemcpyAsync(outs_host + pos, mDevOut, sizeof(float) * (size_t) B * nout_act, hipMemcpyDeviceToHost, mStream));
        else
            HCV_TRY(hipMemcpy2DAsync(outs_host + pos, sizeof(float) * (size_t) out_stride, mDevOut, sizeof(float) * B, sizeof(float) * B, nout_act,
                                     hipMemcpyDeviceToHost, mStream));
        HCV_TRY(hipEventRecord(mEvHostDone, mStream));
        own.leave();
        HCV_TRY(hipEventSynchronize(mEvHostDone));
    }
    if (mProfiling) collect_events();
    return true;
}

bool Engine::process_dev(const float *ins, int64_t in_stride, float *outs, int64_t out_stride, uint32_t nin_act, uint32_t nout_act, uint64_t n,
                         bool sync, hipEvent_t after)
{
    DeviceGuard dg(mDevice);
    nout_act = std::min(nout_act, mCfg.nout);
    nin_act = std::min(nin_act, mCfg.nin);
    if (!nout_act || !n) return true;
    {
        if (!audio_enter(n))
        {
            // a stream-start collision (hcv_engine.h): silence for this call; nothing of the engine's state is touched
            HCV_TRY(hipMemset2DAsync(outs, sizeof(float) * (size_t) out_stride, 0, sizeof(float) * n, nout_act, mStream));
            if (sync) HCV_TRY(hipStreamSynchronize(mStream));
            return true;
        }
        OwnerGuard own(this);
        if (!fence_chains(/* keep_forward */ true) || !update_active_matrix(mCfg.diag ? nout_act : nin_act, nout_act) || !apply_pending_resets()) return false;
        if (after)
        {
            // serial blocks write `outs` from the main stream; a streamed block's writer (a stage stream's inverse, or emit behind
            // it) is reached through the control-work fan-out of enqueue_chunk
            HCV_TRY(hipStreamWaitEvent(mStream, after, 0));
            mExtDirty = true;
        }
        mCallWaits = sync;
        for (uint64_t pos = 0; pos < n; pos += mMaxBlock)
        {
            const uint32_t B = (uint32_t) std::min<uint64_t>(mMaxBlock, n - pos);
            if (!enqueue_chunk(ins + pos, in_stride, outs + pos, out_stride, nin_act, nout_act, B)) return false;
        }
        // (a call that waits: the main stream goes behind the boundary chains the block left running here, inside the ownership the call
        // holds anyway, and the wait below needs none)
        if (sync && !fence_chains()) return false;
        own.leave();
    }
    if (sync)
    {
        HCV_TRY(hipStreamSynchronize(mStream));
        if (mProfiling) collect_events();
    }
    return true;
}

} // namespace hcv
