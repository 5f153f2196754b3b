// In-launch hand-overs between the workgroups of ONE launch, with forward progress by construction: the pieces the fused kernels share
// (hcv_fft_split.hip: one-launch blocks of one-output engines; hcv_kernels.hip: the fused stage boundary of real-time calls).
#pragma once

#include "hcv_engine.h"

#include <cstdlib>

namespace hcv
{

// AGENT = the value is handed to other workgroups of the SAME launch (the fused block kernel): written through with
// agent-scope relaxed atomics instead of plain stores (MI355X_MICROARCH.md: 8-byte agent atomics on both sides)
template <bool AGENT> __device__ __forceinline__ void put2(float2 *p, float2 v)
{
    if constexpr (AGENT)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), ((unsigned long long) __float_as_uint(v.y) << 32) | __float_as_uint(v.x),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}
template <bool AGENT> __device__ __forceinline__ void put1(float *p, float v)
{
    if constexpr (AGENT) __hip_atomic_store(reinterpret_cast<unsigned *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
// 16 bytes written through at agent scope in ONE store (a wave's stores then cover whole lines: pairs of 8-byte write-through stores 16 bytes
// apart are partial-line writes, each a read-modify-write at the memory side — 2 MB of them took a launch's last workgroups 50 us)
typedef float hcv_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void put4_agent(float4 *p, float4 v)
{
    const hcv_v4f d = { v.x, v.y, v.z, v.w };
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(d) : "memory");
}
template <bool AGENT> __device__ __forceinline__ float2 get2(const float2 *p)
{
    if constexpr (AGENT)
    {
        const unsigned long long a = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_float2(__uint_as_float((unsigned) a), __uint_as_float((unsigned) (a >> 32)));
    }
    else
        return *p;
}

// The scheme, as first used by the fused 1 x 1 block:
//
// One whole-hop block of a 1 x 1 engine — forward transform, multiply-accumulate over the P live partitions, inverse transform,
// PartitionedConvolve::process for one hop (PartitionedConvolve.cpp:243-385) — as ONE launch with in-launch hand-overs in place of
// two kernel boundaries.  What crosses workgroups inside the launch (X[h], then Y) is written and read with agent-scope 8-byte
// atomics; every thread drains its stores before its workgroup is counted in, and the consumers poll the counter from one lane
// with relaxed agent loads (`bar[k]` counts arrivals over all launches and the host keeps the running totals: no reset).
//
// Forward progress BY CONSTRUCTION: no workgroup ever waits without bound.  A workgroup that spins on a counter holds its CU, so a
// consumer placed while one of its producers is not yet resident could keep that producer off the chip for good (other engines'
// launches, a CU mask or a 32-CU partition take the rest; the dispatcher's block order is per XCD and promises nothing across
// them).  Two things keep that from ever mattering:
//   * producers have the LOW block indices (forward transforms, then multiply-accumulate, whose first workgroups go on to the
//     inverse), so in the ordinary case everything a workgroup waits for was dispatched before it;
//   * every wait is BOUNDED (`spin` polls, ~1 us each), and a workgroup whose wait runs out does the missing work ITSELF: every
//     task of the launch (residue class r of a forward transform, a multiply-accumulate bin range) is a pure function of data
//     that is complete before the launch, or of tasks it can in turn complete itself, and writes the same values whoever runs
//     it and however often — so a waiter walks the unfinished tasks (a per-task flag holds the sequence number of the launch that
//     last completed it), runs them, and goes on.  Nobody then depends on a workgroup that is not running: the launch
//     completes on one CU as on 256, under any mask, beside any number of other engines (work stealing, in effect — a consumer
//     that got onto the chip early does the producers' work instead of idling).  The task's own workgroup, placed late, repeats
//     it (same values) and is the only one counted in `bar`, so the counters' running totals stay exact.
// HCV_COOP_SPIN = polls before helping (default 2048: about a millisecond of ~0.5 us polls; 0 = help at once: the tests run the whole parity suite that way).
struct FusedSync
{
    unsigned *bar;                       // [2] arrival counters: forward transforms, multiply-accumulate
    unsigned long long *flagF, *flagM;   // per task: sequence number of the launch that last completed it
    unsigned long long seq;              // this launch
    unsigned targetA, targetB;           // what the two counters read when this launch's producers have all arrived
    int spin;
};

// (every helper takes the thread index from its caller: a workitem-id read inside the out-of-line slow path would make the kernel
// keep the packed ids alive in a register of their own up to the call — one register too many for the 128 of the multi-hop kernel)
// thread 0's value, to every thread of the workgroup
__device__ __forceinline__ int wg_broadcast(int tid, int v, int *slot)
{
    __syncthreads();
    if (tid == 0) *slot = v;
    __syncthreads();
    return *slot;
}
// publish a finished task: every thread's agent-scope stores have been written through, then one lane sets the task's flag and
// — the task's own workgroup only (counter != nullptr) — counts the workgroup in
__device__ __forceinline__ void grid_publish(int tid, unsigned long long *flag, unsigned long long seq, unsigned *counter)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
    {
        __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (counter) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// bounded wait: true when the counter reached `target` within `spin` polls.  FRESH: `slot` has not been read since the
// workgroup's last barrier (the ordinary path gives each of its two waits a slot of its own), which saves the barrier in front
template <bool FRESH = false> __device__ __forceinline__ bool grid_wait_bounded(int tid, unsigned *counter, unsigned target, int spin, int *slot)
{
    int ok = 0;
    if (tid == 0)
        for (int k = 0; k < spin; k++)
        {
            ok = (int) (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0;
            if (ok) break;
            // (a few quick polls for the hand-overs of a ten-microsecond block, then half a microsecond between them: a launch with
            // hundreds of waiting workgroups must not keep the producers' arrivals queueing behind its polls)
            if (k < 32) __builtin_amdgcn_s_sleep(1);
            else __builtin_amdgcn_s_sleep(20);
        }
    if constexpr (FRESH)
    {
        if (tid == 0) *slot = ok;
        __syncthreads();
        return *slot != 0;
    }
    else
        return wg_broadcast(tid, ok, slot) != 0;
}
__device__ __forceinline__ bool task_done(int tid, const unsigned long long *flag, unsigned long long seq, int *slot)
{
    int d = 0;
    if (tid == 0) d = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == seq;
    return wg_broadcast(tid, d, slot) != 0;
}

// What a multiply-accumulate workgroup does once one of its waits has run out (`from_mac_wait`: the second one).  Kept OUT OF
// LINE and entered as the last thing the kernel does, so that the ordinary path — straight-line code, the bodies inlined once —
// pays nothing for it: no registers held across it, no scratch, no extra copies of the transforms in its instruction stream (a
// helping loop around the inlined bodies was measured against exactly that: hoisted loop invariants took the 1 x 1 kernel from
// 98 to 222 registers and put the other two into scratch).  `Bodies` provides forward(task), mac_old(m) — the part of
// multiply-accumulate task m that needs nothing of this launch —, mac_new(m) — the rest, the reduction and the store of Y — and
// inverse(j).  Helpers start at different tasks and skip what has been completed meanwhile: they share the work instead of
// repeating it.
// the first task at or behind position k (in the rotation starting at `first`) that this launch has not completed, or `count`: the
// first wave looks at 64 flags at a time
__device__ __forceinline__ int next_undone(int tid, const unsigned long long *flags, unsigned long long seq, int count, int first, int k, int *slot)
{
    for (; k < count; k += 64)
    {
        int found = -1;
        if (tid < 64)
        {
            const int kk = k + tid;
            bool undone = false;
            if (kk < count)
            {
                int task = first + kk;
                if (task >= count) task -= count;
                undone = __hip_atomic_load(flags + task, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq;
            }
            const unsigned long long mask = __ballot(undone);
            if (mask) found = k + __ffsll((long long) mask) - 1;
        }
        found = wg_broadcast(tid, found, slot);
        if (found >= 0) return found;
    }
    return count;
}

template <class Bodies>
__device__ __forceinline__ void fused_slow_path(Bodies &b, const FusedSync &sy, int m, int nfwd, int nmac, int ninv, bool from_mac_wait, int *slot)
{
    if (!from_mac_wait)
    {
        // the forward tasks nobody has completed, then this workgroup's own multiply-accumulate from the start (nothing was kept)
        const int first = (int) ((long long) m * nfwd / nmac);
        for (int k = next_undone(b.tid, sy.flagF, sy.seq, nfwd, first, 0, slot); k < nfwd; k = next_undone(b.tid, sy.flagF, sy.seq, nfwd, first, k + 1, slot))
        {
            int task = first + k;
            if (task >= nfwd) task -= nfwd;
            b.forward(task);
            grid_publish(b.tid, sy.flagF + task, sy.seq, nullptr);
        }
        b.mac_old(m);
        b.mac_new(m);
        grid_publish(b.tid, sy.flagM + m, sy.seq, sy.bar + 1);
        if (m >= ninv) return;
        if (grid_wait_bounded(b.tid, sy.bar + 1, sy.targetB, sy.spin, slot))
        {
            b.inverse(m);
            return;
        }
    }
    // the multiply-accumulate tasks nobody has completed (the forward transforms are known to be in), then the inverse
    const int first = m * (nmac / ninv) + 1;
    for (int k = next_undone(b.tid, sy.flagM, sy.seq, nmac, first, 0, slot); k < nmac; k = next_undone(b.tid, sy.flagM, sy.seq, nmac, first, k + 1, slot))
    {
        int task = first + k;
        if (task >= nmac) task -= nmac;
        b.mac_old(task);
        b.mac_new(task);
        grid_publish(b.tid, sy.flagM + task, sy.seq, nullptr);
    }
    b.inverse(m);
}

// The ordinary path of one workgroup of a fused launch: workgroup w < nfwd runs forward task w; workgroup nfwd + m runs
// multiply-accumulate task m and, for m < ninv, the inverse.  `Slow` = the kernel's out-of-line entry to fused_slow_path.
template <class Bodies, class Slow>
__device__ __forceinline__ void fused_roles(Bodies &b, const FusedSync &sy, int w, int nfwd, int ninv, bool mac_needs_fwd, int *slot, const Slow &slow)
{
    if (w < nfwd)
    {
        b.forward(w);
        grid_publish(b.tid, sy.flagF + w, sy.seq, sy.bar);
        return;
    }
    const int m = w - nfwd;
    b.mac_old(m);
    if (mac_needs_fwd && !grid_wait_bounded<true>(b.tid, sy.bar, sy.targetA, sy.spin, slot))
    {
        slow(m, false);
        return;
    }
    b.mac_new(m);
    grid_publish(b.tid, sy.flagM + m, sy.seq, sy.bar + 1);
    if (m >= ninv) return;
    if (!grid_wait_bounded<true>(b.tid, sy.bar + 1, sy.targetB, sy.spin, slot + 1))
    {
        slow(m, true);
        return;
    }
    b.inverse(m);
}

// ---- sharded arrival counters (the n x m block, hcv_fused_nxm.hip) ----
// An agent-scope atomic is performed at the memory side, one at a time per line, and so is an agent-scope load: the arrivals of a
// launch and the polls of the workgroups waiting for them queue up at the counter's line.  The n x m block's forward launch counts in on
// kShards counters 128 bytes apart (task t on shard t mod kShards) and a waiting workgroup's first 32 lanes poll one shard each; the
// host keeps a running total per shard.
constexpr int kShards = 32, kShardStride = 32;          // (unsigned per shard: 128 bytes)
struct FusedSyncSharded
{
    unsigned *bar;                       // [kShards * kShardStride]: the forward transforms' arrivals
    unsigned long long *flagF;           // per forward task: sequence number of the launch that last completed it
    unsigned long long seq;
    unsigned targetA[kShards];
    int spin;
};
__device__ __forceinline__ void grid_publish_sharded(int tid, unsigned long long *flag, unsigned long long seq, unsigned *counters, int task)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
    {
        __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (counters) __hip_atomic_fetch_add(counters + (task % kShards) * kShardStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// bounded wait for ALL shards; `targets` = the launch's kShards totals (kernel arguments).  `slot` has not been read since the
// workgroup's last barrier
__device__ __forceinline__ bool grid_wait_bounded_sharded(int tid, unsigned *counters, const unsigned *targets, int spin, int *slot)
{
    int ok = 0;
    if (tid < 64)
    {
        const unsigned want = tid < kShards ? targets[tid] : 0u;
        const unsigned *c = counters + (tid < kShards ? tid : 0) * kShardStride;
        for (int k = 0; k < spin; k++)
        {
            const bool reached = tid >= kShards || (int) (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0;
            ok = __all(reached);
            if (ok) break;
            if (k < 32) __builtin_amdgcn_s_sleep(1);
            else __builtin_amdgcn_s_sleep(20);
        }
    }
    if (tid == 0) *slot = ok;
    __syncthreads();
    return *slot != 0;
}

// The host's running totals of the two counters and the launch sequence number move only once the runtime has accepted the
// launch: a refused launch leaves the device counters where they were, and totals that ran ahead of them would make every later
// launch wait (boundedly — and then redo everything by helping) for arrivals that never come.
struct FusedHostState
{
    unsigned *arrived;          // [2]
    unsigned long long *seq;
};
inline int fused_spin()
{
    static const int s = std::getenv("HCV_COOP_SPIN") ? std::max(0, std::atoi(std::getenv("HCV_COOP_SPIN"))) : 2048;
    return s;
}
inline FusedSync fused_sync(unsigned *bar, unsigned long long *flags, const unsigned *arrived, unsigned long long seq, unsigned producers, unsigned macs,
                            unsigned mac_slots = kFusedMacTasks)
{
    FusedSync sy;
    sy.bar = bar;
    sy.flagM = flags;                       // [kFusedMacTasks]
    sy.flagF = flags + mac_slots;           // [kFusedFwdTasks]
    sy.seq = seq + 1;
    sy.targetA = arrived[0] + producers;
    sy.targetB = arrived[1] + macs;
    sy.spin = fused_spin();
    return sy;
}
// host side of the sharded counters: `arrived` = kShards running totals; a launch of n tasks adds n / kShards (+ 1 for the first n mod kShards shards)
inline unsigned shard_share(unsigned n, int shard) { return n / kShards + ((unsigned) shard < n % kShards ? 1u : 0u); }
inline FusedSyncSharded fused_sync_sharded(unsigned *bar, unsigned long long *flags, const unsigned *arrived, unsigned long long seq, unsigned producers)
{
    FusedSyncSharded sy;
    sy.bar = bar;
    sy.flagF = flags;
    sy.seq = seq + 1;
    for (int c = 0; c < kShards; c++) sy.targetA[c] = arrived[c] + shard_share(producers, c);
    sy.spin = fused_spin();
    return sy;
}
inline void fused_arrivals_sharded(unsigned *arrived, unsigned n)
{
    for (int c = 0; c < kShards; c++) arrived[c] += shard_share(n, c);
}
inline hipError_t fused_launched(unsigned *arrived, unsigned long long *seq, unsigned producers, unsigned macs)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    arrived[0] += producers;
    arrived[1] += macs;
    *seq += 1;
    return hipSuccess;
}

} // namespace hcv
