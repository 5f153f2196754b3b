// Device-side pieces of the whole-frame forward transform (hcv_kernels.hip: rfft_frames_direct_kernel), shared with the n x m fused block
// (hcv_fused_nxm.hip), whose forward launch and helping path run the same body.
#pragma once

#include "hcv_fft_device.h"
#include "hcv_fused_sync.h"

namespace hcv
{

// Real post-pass of the forward transform (the maths of pass_real_trig_table<false>,
// HISSTools_FFT_Core.h:934-988): s holds Z = FFT_M(x_even + i x_odd); writes the packed, doubled half
// spectrum to dst[0..M).
// (AGENT: the spectrum is handed to other workgroups of the same launch — written through, hcv_fused_sync.h: put2)
template <int LOG2M, int TG, bool AGENT = false>
__device__ __forceinline__ void real_post_store(LdsBuf<float2> s, int tid, const float2 *__restrict__ tw, float2 *__restrict__ dst)
{
    constexpr int M = 1 << LOG2M;
    for (int k = tid; k <= M / 2; k += TG)
    {
        if (k == 0)
        {
            float2 z = s[0];
            float t1 = z.x + z.y, t2 = z.x - z.y;
            put2<AGENT>(dst, make_float2(t1 + t1, t2 + t2));
        }
        else
        {
            int m = M - k;
            float2 w = tw[k];                            // exp(-i pi k / M)
            float2 z1 = s[k], z2 = s[m];
            float r3 = z1.x + z2.x, i3 = z1.y + z2.y, r4 = z1.x - z2.x, i4 = z1.y - z2.y;
            float u1 = (w.x * i3) + (w.y * r4);
            float u2 = (w.y * i3) - (w.x * r4);
            put2<AGENT>(dst + k, make_float2(r3 + u1, u2 + i4));
            put2<AGENT>(dst + m, make_float2(r3 - u1, u2 - i4));
        }
    }
}

// The same frame with its NEW hop taken straight from the caller's block (the framing copy of PartitionedConvolve.cpp:304-307
// fused into the transform's first-pass loads): positions >= n0 are read from in[pos - n0], older ones from the history ring.
// The transform of hop h also files hop h's samples (the second half of its frame) in the ring, for the next block's first
// frame and for every reader of the history — so a hop-aligned block needs no separate scatter launch, and the stage's stream
// no cross-stream wait before its first kernel.
// (AGENT: the ring's new hop is read by launches that may have started before this one ends — written through, like the spectrum)
template <bool AGENT = false> struct DirectFrameLoadT
{
    static constexpr bool is_lds = false;
    float *hist_row;
    const float *in_row;
    long long base, mask, n0;
    int half;                           // float2 index of the frame's own (second) half: k >= half <=> position >= h * M
    bool live;
    __device__ __forceinline__ float2 operator()(int k) const
    {
        if (!live) return make_float2(0.f, 0.f);
        const long long pos = base + 2LL * k;
        // ONE load through a selected pointer (a branch per element would make the compiler wait for each load in turn
        // instead of keeping a thread's sixteen in flight); the store is uniform per call: k >= half for a whole butterfly row
        const float *src = pos < n0 ? hist_row + (pos & mask) : in_row + (pos - n0);
        const float2 v = *reinterpret_cast<const float2 *>(src);
        if (k >= half) put2<AGENT>(reinterpret_cast<float2 *>(hist_row + (pos & mask)), v);
        return v;
    }
};
using DirectFrameLoad = DirectFrameLoadT<false>;

} // namespace hcv
