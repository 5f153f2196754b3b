// Launch interface of the gfx950 kernels (hcv_kernels.hip).  Internal to the library.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace hcv
{
    constexpr int kMinFFTLog2 = 5;        // PartitionedConvolve.h:18
    constexpr int kMaxFFTLog2 = 20;       // PartitionedConvolve.h:19
    constexpr int kMaxLdsFFTLog2 = 15;    // N = 32768 -> 16384 complex points = 128 KiB of the CU's 160 KiB LDS

    // ---- FFT family (tw = N-th roots of unity, N/2 entries).  N <= 32768 runs inside LDS; larger sizes take the
    //      four-step path of hcv_bigfft.hip and need a BigFFTWork (two scratch buffers + sub-transform tables).
    struct BigFFTWork
    {
        float2 *a = nullptr, *b = nullptr;      // scratch, `elems` float2 each
        size_t elems = 0;
        const float2 *tw1 = nullptr, *tw2 = nullptr;   // (2*M1)-th and (2*M2)-th roots for the column / row pieces
    };
    inline bool is_big_fft(int log2n) { return log2n > kMaxLdsFFTLog2; }
    void big_fft_split(int log2n, int &l1, int &l2);      // N/2 = 2^l1 * 2^l2

    hipError_t launch_rfft_frames(int log2n, const float *hist, long long hist_stride, long long hist_mask, long long h_first, int T, int nin,
                                  float2 *X, int R, const float2 *tw, const BigFFTWork *big, hipStream_t st);
    // hop-aligned block: the new hops come straight from the caller's block `in` (first sample = position n0) and are filed in
    // the history ring by the transform itself (no scatter launch); LDS sizes only
    hipError_t launch_rfft_frames_direct(int log2n, float *hist, long long hist_stride, long long hist_mask, const float *in, long long in_stride,
                                         long long n0, long long h_first, int T, int nin, float2 *X, int R, const float2 *tw, hipStream_t st);
    hipError_t launch_rfft_ir(int log2n, const float *src, long long count, int P, float2 *dst, const float2 *tw, const BigFFTWork *big, hipStream_t st);
    hipError_t launch_rfft_rows(int log2n, const float *src, long long src_stride, long long in_len, int batch, float2 *dst, const float2 *tw,
                                const BigFFTWork *big, hipStream_t st);
    hipError_t launch_rifft_rows(int log2n, const float2 *src, int batch, float *dst, const float2 *tw, const BigFFTWork *big, hipStream_t st);
    hipError_t launch_rifft_overlap_add(int log2n, const float2 *Y, int ksplit, long long ks_stride, long long h_first, int T, int nout,
                                        float *timeline, long long tl_stride, long long tl_mask, const float2 *tw, const BigFFTWork *big,
                                        hipStream_t st);

    // the inverse of a whole-hop block: the valid half of every transform goes straight to the caller's block (row o, hop t at
    // out[o * out_stride + t * M]); no timeline, no emit.  LDS sizes only; rows 8-byte aligned.
    hipError_t launch_rifft_emit(int log2n, const float2 *Y, int ksplit, long long ks_stride, int T, int nout, float *out, long long out_stride,
                                 const float2 *tw, hipStream_t st);

    // ---- XCD pinning of tiny launches (HCV_XCD_PIN): a chain of kernels of a few workgroups each hands its data from kernel to
    //      kernel through memory; workgroup b of a launch runs on XCD b % 8 (observed, never relied on for correctness), so a
    //      launch of 8 x as many workgroups of which only those with b % 8 == 0 work keeps the whole chain on ONE XCD and its
    //      hand-overs in that XCD's L2.  Returns the XCD to pin a launch of `workgroups` to (one-dimensional grids), or -1.
    int xcd_pin_for(long long workgroups);
    // set by the engine around a block's enqueue (thread-local): "this block's chain is tiny and its data fits one L2"; xcd = the
    // XCD this engine's chains go to (engines of one process are spread over the eight)
    void xcd_pin_hint(bool on, int xcd = 0);

    // ---- residue-split transforms (hcv_fft_split.hip): one hop transform over several workgroups that share nothing, for blocks
    //      of a few transforms.  `applies` = the rule (HCV_FFT_SPLIT = 0 / 1 forces it); `prepare` uploads the sub-transform tables
    //      of the current device ahead of the first launch.
    bool fft_split_applies(int log2n, int transforms);
    void fft_split_prepare(int log2n);
    hipError_t launch_rfft_frames_direct_split(int log2n, float *hist, long long hist_stride, long long hist_mask, const float *in, long long in_stride,
                                               long long n0, long long h_first, int T, int nin, float2 *X, int R, const float2 *tw, hipStream_t st);
    // (started / started_marks / started_seq: the launch's first workgroups write `started_seq` to marks 128 bytes apart as they begin)
    hipError_t launch_rifft_emit_split(int log2n, const float2 *Y, int ksplit, long long ks_stride, int T, int nout, float *out, long long out_stride,
                                       const float2 *tw, hipStream_t st, unsigned long long *started = nullptr, int started_marks = 0,
                                       unsigned long long started_seq = 0);

    // ---- the fused 1 x 1 block (hcv_fft_split.hip): one hop of a one-input, one-output engine in ONE launch.  h = the hop, h_mac =
    //      the hop partition 0 reads (h with a lead slot, h - 1 for a lone stage), H = the pair's P live partitions.  Hand-over state,
    //      touched by these launches only, all zero-initialised: bar = two arrival counters, flags = kFusedMacTasks + kFusedFwdTasks
    //      per-task completion marks; arrived / seq = the host's running totals of the counters and the launch sequence number
    //      (advanced only by launches the runtime accepted).
    constexpr int kFusedMacTasks = 64, kFusedFwdTasks = 2048 * 9;
    bool fused_block_1x1_applies(int log2n);
    hipError_t launch_fused_block_1x1(int log2n, float *hist, long long hist_mask, const float *in, long long n0, long long h, float2 *X, int Rring,
                                      const float2 *H, int P, long long h_mac, float2 *Y, float *out, const float2 *tw, unsigned *bar, unsigned long long *flags,
                                      unsigned *arrived, unsigned long long *seq, hipStream_t st);

    // several hops of a short stage per block (1 x 1, 4096-point partitions, T = 2 .. 4 hops: BASELINE config 2 at 8192-sample calls);
    // h = the block's first hop, Y = room for T spectra, out = the block's T hops of output
    bool fused_block_hops_applies(int log2n, int T);
    hipError_t launch_fused_block_hops(int log2n, float *hist, long long hist_mask, const float *in, long long n0, long long h, int T, float2 *X, int Rring,
                                       const float2 *H, int P, long long h_mac, float2 *Y, float *out, const float2 *tw, unsigned *bar, unsigned long long *flags,
                                       unsigned *arrived, unsigned long long *seq, hipStream_t st);

    // nin inputs -> one output, or 1 x 1 with a long reduction (K = nin P split over the waves of 1024-thread workgroups); H = output 0's
    // pairs, `hstride` float2 between two inputs' spectra
    hipError_t launch_fused_block_nx1(int log2n, float *hist, long long hist_stride, long long hist_mask, const float *in, long long in_stride, long long n0,
                                      long long h, int nin, float2 *X, int Rring, const float2 *H, long long hstride, int P, long long h_mac, float2 *Y,
                                      float *out, const float2 *tw, unsigned *bar, unsigned long long *flags, unsigned *arrived, unsigned long long *seq,
                                      hipStream_t st);

    // ---- the block of an n x m matrix (hcv_fused_nxm.hip): one hop of the last stage of an engine with several outputs as a forward
    //      launch on `fwd_stream` and a multiply-accumulate launch on `st` that meet through the stage's sharded arrival counters, no
    //      event between them, then the inverse on `st`.  Lead-slot stages of 16384 points only; P = live partitions, lead slot included.
    //      bar = kFusedShards x kFusedShardStride unsigned, flags = kFusedFwdTasks marks (both zero-initialised, touched by these launches
    //      only); arrived = kFusedShards running totals of the host, seq = the launch sequence number; chained = the previous launch on this
    //      state was the block right before this one (its forward launch then waits for that block's multiply-accumulate to end: a
    //      scheduling hint); ev_begin / ev_end (optional) are recorded around the multiply-accumulate launch; helped (optional, host-mapped) counts the
    //      launches whose wait for the forward transforms ran out.  `plan` says whether the shape is taken.
    constexpr int kFusedShards = 32, kFusedShardStride = 32;
    struct FusedNxmPlan { int ms, tiles, kper_old, nmac, nfwd; };
    bool fused_block_nxm_plan(int log2n, int nin, int nout, int P, size_t y_elems, FusedNxmPlan *pl);
    hipError_t launch_fused_block_nxm(const FusedNxmPlan &pl, float *hist, long long hist_stride, long long hist_mask, const float *in, long long in_stride,
                                      long long n0, long long h, int nin, int nin_alloc, int nout, float2 *X, int Rring, const float2 *H, int hparts, int P,
                                      float2 *Y, float *out, long long out_stride, const float2 *tw, unsigned *bar, unsigned long long *flags, unsigned *arrived,
                                      unsigned long long *seq, hipStream_t fwd_stream, hipStream_t st, bool chained, hipEvent_t ev_begin = nullptr,
                                      hipEvent_t ev_end = nullptr, unsigned *helped = nullptr);
    const float2 *fft_split_sub_table(int log2s);          // the (2 S)-th roots of the residue-split transforms' sub-transform, current device

    hipError_t big_rfft_frames(int log2n, const float *hist, long long hist_stride, long long hist_mask, long long h_first, int T, int nin, float2 *X,
                               int R, const float2 *tw, const BigFFTWork &w, hipStream_t st);
    hipError_t big_rfft_ir(int log2n, const float *src, long long count, int P, float2 *dst, const float2 *tw, const BigFFTWork &w, hipStream_t st);
    hipError_t big_rfft_rows(int log2n, const float *src, long long src_stride, long long in_len, int batch, float2 *dst, const float2 *tw,
                             const BigFFTWork &w, hipStream_t st);
    hipError_t big_rifft_rows(int log2n, const float2 *src, int batch, float *dst, const float2 *tw, const BigFFTWork &w, hipStream_t st);
    hipError_t big_rifft_overlap_add(int log2n, const float2 *Y, long long h_first, int T, int nout, float *timeline, long long tl_stride,
                                     long long tl_mask, const float2 *tw, const BigFFTWork &w, hipStream_t st);

    // ---- spectral multiply-accumulate ----
    struct MacShape
    {
        int M;              // bins per spectrum (= N/2)
        int R;              // input-spectrum ring slots
        int P, Pcap;        // live partitions, allocated partitions per pair
        int nin, nin_alloc; // live inputs, allocated inputs per output row
        int nout;           // live outputs
        int diag;           // parallel (diagonal) mode
        int T;              // hops in this launch
        int max_ksplit;     // 0 = unlimited (bounded by the Y partial buffer)
        int target_blocks;  // 0 = default; > 0 = aim for this many workgroups (background work keeps a small footprint)
        int ot_cap;         // 0 = none; > 0 = at most this many outputs per thread (more, smaller workgroups for launches without k-slices)
        int steady = 0;     // the launch will run unchecked (every pair sees all P partitions): the offline shapes may take the matrix cores
        long long hop_min = -(1LL << 62);   // matrix-core launches only: input hops before this one count as silence (the ramp-up after a global
                                            // reset: every pair's first hop is 0, and what the ring holds from before the reset must not be seen)
    };
    struct MacPlan
    {
        int ot, tt;                 // register tile: outputs x hops per thread
        int bx, by, tz;             // threads along bins, hop tiles per workgroup, hop-tile blocks
        int binblocks, outtiles, ksplit, kper;
        int nt;                     // stream H with nontemporal loads
        int inwg;                   // > 0: the k-slices are the waves of ONE workgroup (this many, 64 lanes = 128 bins each) and are added
                                    // up in LDS — ksplit = 1, no partial sums in memory, no reduce_partials launch (small engines)
        int mfma = 0;               // > 0: the offline kernel on the matrix cores (hcv_mac_mfma.hip), tiles of 32 x this many hops; ot = 16
    };
    // hcv_mac_mfma.hip: offline calls (>= 32 hops per launch, steady state, a matrix) as one dense contraction per bin on the f32 MFMA
    bool mac_mfma_applies(const MacShape &s);
    void mac_mfma_plan(const MacShape &s, MacPlan &pl);
    void mac_plan(const MacShape &s, MacPlan &pl);
    // (dst: where the sum goes; nullptr = slice 0 of Y itself)
    hipError_t launch_reduce_partials(float2 *Y, int ksplit, long long ks_stride, long long elems, hipStream_t st, float2 *dst = nullptr);
    hipError_t launch_spectral_mac(const MacShape &s, const MacPlan &pl, const float2 *X, const float2 *H, float2 *Y, const long long *hv,
                                   long long h_first, bool check, hipStream_t st);

    // ---- exact per-pair restart (hcv_ghost.hip) ----
    struct GhostEntry
    {
        long long h_r;              // first frame the restarted pair may see (= the hop the restart fell into)
        const float4 *g0, *g1;      // spectra of the pre-restart part of frames h_r and h_r + 1
        int i, pad;                 // the pair's input column in H
    };
    // `rows` is a host array (passed to the kernel by value)
    hipError_t launch_ghost_hist(const float *hist, long long hist_stride, long long hist_mask, const int *rows, int nrows, float *ghost, long long Lg,
                                 long long t0, hipStream_t st);
    // follows a spectral_mac launch of the same shape: subtracts the ghost products from slice 0 of its partial sums.  Entries
    // come from a per-output table (start[o] .. start[o + 1]) or, for a one-pair launch, from `single`
    hipError_t launch_ghost_mac(const MacShape &s, const float2 *H, float2 *Y, long long h_first, const int *start, const GhostEntry *ent,
                                const GhostEntry *single, hipStream_t st);
    hipError_t launch_timeline_sub(float *row, long long mask, long long base, const float *tmp, int n, float scale, long long t_min, hipStream_t st);

    // ---- time-domain head ----
    // calls of <= 256 samples take fir_head_small_kernel, which can read the call's own samples from the caller's block (din) instead
    // of the ring: it then needs the ring complete only up to the call's first sample
    bool fir_head_is_small(int B, int nin, int Lpad, int diag);
    struct EmitSources;
    hipError_t launch_fir_head(const float *hist, long long hist_stride, long long hist_mask, const float *taps, int Lpad, int tap_stride, int nin,
                               int nin_alloc, int nout, int diag, long long n0, int B, const long long *valid_from, bool check, float *out,
                               long long out_stride, hipStream_t st, const float *din = nullptr, long long in_stride = 0, const struct EmitSources *emit = nullptr,
                               float *ring = nullptr);

    // ---- ring bookkeeping ----
    hipError_t launch_scatter_input(const float *in, long long in_stride, int B, int nin, float *hist, long long hist_stride, long long hist_mask,
                                    long long n0, hipStream_t st);
    constexpr int kMaxStages = 8;         // MonoConvolve builds at most 4 FFT stages (MonoConvolve.cpp:235-252); the extended
                                          // tail ladder (hcv_api.hip) adds up to 3 more
    struct EmitSources
    {
        float *timeline[kMaxStages];
        long long stride[kMaxStages];
        long long mask[kMaxStages];
        int count;
    };
    hipError_t launch_emit(const EmitSources &src, long long n0, int B, int nout, const float *td, long long td_stride, float *out,
                           long long out_stride, hipStream_t st);
    // ---- sharded matrices: the sum of the input groups' partial output blocks ----
    constexpr int kMaxParts = 16;
    struct PartSources
    {
        const float *part[kMaxParts];
        int count;
    };
    hipError_t launch_sum_parts(const PartSources &src, long long part_stride, int B, int nout, float *out, long long out_stride, hipStream_t st);
    // ---- one-shot spectral convolution / correlation ----
    hipError_t launch_spectral_pointwise(float2 *a, const float2 *b, int M, float scale, int correlate, hipStream_t st);
    hipError_t launch_segment_op(float *out, const float *t, long long o_off, long long off, long long n, int op, hipStream_t st);
    hipError_t launch_fold_copy(float *dst, const float *in, long long n, long long fold, int off, hipStream_t st);
    hipError_t launch_fill_i64(long long *p, long long n, long long v, hipStream_t st);
    // a control call's swap section in one launch (pointers and byte counts multiples of 16)
    constexpr int kSwapSegs = 12, kSwapFills = 12;
    struct SwapSeg { void *dst; const void *src; long long copy, zero; };
    struct SwapFill { long long *p; long long v; };
    struct SwapPlan
    {
        SwapSeg seg[kSwapSegs];
        SwapFill fill[kSwapFills];
        int nseg = 0, nfill = 0;
        bool add(void *dst, const void *src, long long copy, long long zero)
        {
            if (nseg >= kSwapSegs || ((copy | zero) & 15) || (reinterpret_cast<uintptr_t>(dst) & 15) || (copy && (reinterpret_cast<uintptr_t>(src) & 15))) return false;
            seg[nseg++] = { dst, src, copy, zero };
            return true;
        }
        bool set(long long *p, long long v)
        {
            if (nfill >= kSwapFills) return false;
            fill[nfill++] = { p, v };
            return true;
        }
    };
    hipError_t launch_swap_in(const SwapPlan &pl, hipStream_t st);
    hipError_t launch_regrow_spectra(const float2 *src, float2 *dst, long long pairs, int Pold, int Pnew, int M, hipStream_t st);
    hipError_t launch_regrow_ring(const float2 *src, float2 *dst, int nin, int Rold, int Rnew, int M, long long h_last, int live, hipStream_t st);

    // each translation unit's code object loaded on the current device now, not at the first launch of one of its kernels (see hcv_kernels.hip)
    void preload_kernels();
    void preload_mac();
    void preload_mac_tiled();
    void preload_mac_mfma();
    void preload_ghost();
    void preload_bigfft();
    void preload_fft_split();
    void preload_fused_nxm();
}
