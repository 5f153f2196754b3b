// Exact per-pair restart, host side (kernels and the scheme: hcv_ghost.hip): ghost spectra of the input before a restart,
// the pair's pending output retired from the timelines and from the deferred slices, and the reset bookkeeping.
// All of it is control work on the engine's main stream.

#include "hcv_engine_impl.h"

namespace hcv
{

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
        // whole-hop mode: every block delivers all it computes, no stage up to the pivot has anything pending (the extended ladder's
        // rungs behind it keep their one hop of latency)
        if (mTailHeadPrev && si <= mPivot) continue;
        MacShape sh = mac_shape(st, /* P */ (int) P, /* Pcap */ st.hparts(),
                                /* nin */ 1, /* nin_alloc */ 1, /* nout */ 1, /* diag */ 0,
                                /* T */ 1, /* max_ksplit */ (int) std::max<size_t>(1, st.y_elems / st.M));
        MacPlan pl;
        mac_plan(sh, pl);
        float2 *Y = st.Yq[0];
        const float2 *H = st.Ht() + pair * st.hstride();
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
        MacShape sh = mac_shape(st, /* P */ (int) covered, /* Pcap */ st.hparts(),
                                /* nin */ 1, /* nin_alloc */ 1, /* nout */ 1, /* diag */ 0,
                                /* T */ 1, /* max_ksplit */ (int) std::max<size_t>(1, st.y_elems / st.M));
        MacPlan pl;
        mac_plan(sh, pl);
        float2 *Y = st.Yq[0];
        const float2 *H = st.Ht() + pair * st.hstride() + st.M;                  // partitions 1 .. covered at hop pre_hop - 1
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


// The caller changed how many inputs / outputs it passes (Convolver.cpp:148-153, NToMonoConvolve.cpp:41 process only the
// active channels).  A pair that drops out is muted at the sample — what it still had to deliver is retired — and a pair that
// comes (back) in restarts from silence like any restarted pair; the rows of the history and of the timelines that were not
// maintained while a channel was inactive are thereby never looked at (the ghost spectra are taken from the same ring contents
// as the frames, so whatever sits there cancels exactly), and the timeline rows of a returning output are cleared.
// (The reference freezes an inactive pair's private state instead and resumes it later as if no time had passed.)
bool Engine::update_active_matrix(uint32_t rows_in, uint32_t nout_act)
{
    if (mN <= 0 || (rows_in == mLastNin && nout_act == mLastNout)) return true;
    if (!fence_background(exact_restart())) return false;
    for (uint32_t o = 0; o < mCfg.nout; o++)
        for (uint32_t c = 0; c < mNinAlloc; c++)
        {
            const size_t p = (size_t) o * mNinAlloc + c;
            const uint32_t row = mCfg.diag ? o : c;
            const bool was = o < mLastNout && row < mLastNin, now = o < nout_act && row < rows_in;
            if (!mLoaded[p] || was == now) continue;
            if (was && !__atomic_load_n(&mPending[p], __ATOMIC_ACQUIRE) && !mRetired[p])
            {
                if (!retire_pair(p)) return false;
                mRetired[p] = 1;
            }
            if (now) __atomic_store_n(&mPending[p], (uint8_t) 1, __ATOMIC_RELEASE);
        }
    for (uint32_t o = mLastNout; o < nout_act && o < mCfg.nout; o++)
        for (Stage *st : mStages) HCV_TRY(hipMemsetAsync(st->timeline + (size_t) o * st->tl_len, 0, sizeof(float) * st->tl_len, mStream));
    mCtlDirty = true;
    return true;
}

bool Engine::apply_pending_resets()
{
    // (reset_pair / reset_all write the flags from other threads without the lock: a cheap look first, then every flag is TAKEN —
    // exchanged with 0 — so that one raised after this point waits for the next block instead of being wiped)
    const uint32_t gen = mResetAllGen.load(std::memory_order_acquire);
    const bool reset_all_seen = gen != mResetAllSeen;
    bool any = reset_all_seen;
    for (size_t p = 0; p < mPending.size() && !any; p++) any = __atomic_load_n(&mPending[p], __ATOMIC_ACQUIRE) != 0;
    if (!any) return true;
    mResetAllSeen = gen;
    std::vector<uint8_t> &taken = mTaken;
    if (taken.size() != mPending.size()) taken.assign(mPending.size(), 0);
    bool all = true;
    for (size_t p = 0; p < mPending.size(); p++)
    {
        taken[p] = __atomic_exchange_n(&mPending[p], (uint8_t) 0, __ATOMIC_ACQ_REL);
        if (reset_all_seen) taken[p] = 1;           // (reset_all: every pair restarts at this sample)
        if (mLoaded[p] && !taken[p]) all = false;
    }
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
        std::vector<size_t> &restart = mRestartScratch;
        restart.clear();
        // (the pairs' fences — first hop / first sample each may see — go out in launches of kSwapFills values, not one launch per value)
        SwapPlan fences;
        auto fence = [&](long long *where, long long v) -> bool
        {
            if (fences.nfill == kSwapFills)
            {
                HCV_TRY(launch_swap_in(fences, mStream));
                fences.nfill = 0;
            }
            return fences.set(where, v);
        };
        for (size_t p = 0; p < taken.size(); p++)
        {
            if (!taken[p]) continue;
            if (!mRetired[p] && !retire_pair(p)) return false;
            if (mLoaded[p]) restart.push_back(p);
            else release_ghost(p);
            for (Stage *st : mStages)
            {
                const long long hvv = mN / st->M;
                if (!fence(st->hv + p, hvv)) return false;
                st->max_hv = std::max(st->max_hv, hvv);
                if (hvv != 0) st->hv_zero = false;
            }
            if (mTdValid)
            {
                if (!fence(mTdValid + p, mN)) return false;
                mTdMaxValid = std::max(mTdMaxValid, mN);
            }
        }
        HCV_TRY(launch_swap_in(fences, mStream));
        if (!make_ghost_event(restart)) return false;
        if (!rebuild_ghost_tables()) return false;
    }
    std::fill(mRetired.begin(), mRetired.end(), 0);
    return true;
}

} // namespace hcv
