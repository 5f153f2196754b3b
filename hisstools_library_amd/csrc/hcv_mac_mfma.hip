// K1c: the spectral multiply-accumulate of OFFLINE calls (one process() call spans 32 hops or more of the stage) on the matrix cores.
//
//   Y[ks][t][o][b] = sum over (i, p) in this block's k-slice of  X[i][(h_t - p) mod R][b] * H[o][i][p][b],   h_t = h_first + t
//
// — the same sum as spectral_mac_kernel (hcv_mac.hip; PartitionedConvolve.cpp:387-426 per partition and pair, looped over the hops of
// a long call by PartitionedConvolve.cpp:298-299, 321-348, summed over the inputs by NToMonoConvolve.cpp:39-42).
//
// Why another kernel.  The register-tiled kernels share an IR spectrum over at most 8 hops: at 8 hops per read c5's 11.8 GB of spectra
// stream at 5.8 TB/s while the FMA pipe sits at 0.29 of its rate (hcv_mac_tiled.hip), and a thread cannot hold more hops.  With 32 - 64
// hops per read the bound moves to the arithmetic — and that arithmetic is a dense contraction: FOR ONE BIN b
//
//     Y_b [hop t][output o]  =  sum_k  A_b [t][k] * B_b [k][o],     k = (input i, partition p),   A_b [t][(i, p)] = X[i][h_t - p][b]   (Toeplitz in t, p)
//
// a complex (hops x K) x (K x outputs) product per bin.  Written over the reals, [Yr Yi] = Xr [Hr Hi] + Xi [-Hi Hr]: one
// v_mfma_f32_32x32x2_f32 per (bin, partition, input) is exactly ONE complex rank-1 update of 32 hops x 16 outputs — rows = hops, the two
// k values = (re, im) of the input spectrum, columns = (output, re / im) — 4096 flops, all of them needed (8 real flops per complex
// multiply-add, as the scalar kernels spend).  The f32 MFMA is an exact f32 fmaf chain at the f32 vector peak (157 TFLOP/s), but one
// operand register per lane feeds 2048 multiply-adds where a v_fma_f32 needs its operands per lane: the matrix core is how the FMA rate
// is reached at all here (cdna_hip_programming.md section 3: 122 - 147 TFLOP/s against 52 for a packed-VALU kernel).
//
// Workgroup = 256 threads = 4 waves; tile = 16 bins x 16 outputs x (32 MT) hops x one k-slice.  Wave w owns bins 4 w .. 4 w + 3 of the
// tile: 4 x MT accumulator tiles of 16 registers.  Per chunk of PC = 16 partitions of one input the workgroup stages, bin-contiguous
// 128-byte runs from HBM (nontemporal: every IR element is read once per launch and hop tile):
//     Hs[pp][2 o + c][bin]   the chunk's IR spectra, component-major so that a lane's B operand for 4 bins is ONE ds_read_b128
//     Xs[c][hop & (W-1)][bin]   a ring over the hop axis of the input's spectra: chunk after chunk the window slides down by PC hops, so
//                               each chunk loads only the PC hops it gains (X traffic = H traffic / 16)
// and every wave then issues, per partition, 1 + MT LDS reads of 16 bytes and 4 MT matrix instructions.  The next chunk's global loads
// are in flight (in registers) under the current chunk's arithmetic.  Bin 0 carries (DC, Nyquist) — two real products — through a
// B operand of its own in the one wave that owns it: B[k][2 o + c] = (c == k) ? H.c : 0.
// The RAMP-UP of a stream (the hops after a global reset, while partitions still reach back before the stream's first hop) runs here too:
// every pair's bound is the same then, hop 0, so the staged input hops before `hop_min` are simply zeros — no per-pair checks in the loop.
// (An offline convolution of a file shorter than the impulse response is ramp-up from its first sample to its last.)
//
// Row strides of 20 floats keep every 16-byte LDS access of 8 consecutive lanes on 8 different 4-bank groups.

#include "hcv_kernels.h"
#include "hcv_fft_device.h"
#include "hcv_mac_params.h"

#include <algorithm>
#include <cstdlib>

namespace hcv
{

namespace
{
    typedef float f32x16 __attribute__((ext_vector_type(16)));

    constexpr int kNB = 16;             // bins per workgroup
    constexpr int kPC = 16;             // partitions per staged chunk
    constexpr int kBS = 20;             // floats between two LDS rows of 16 bins
    constexpr int kHRow = 32 * kBS;     // floats per partition of Hs
}

template <int MT, bool NT>
__global__ __launch_bounds__(256, 2) void spectral_mac_mfma_kernel(MacParams a)
{
    constexpr int TH = 32 * MT;                     // hops of the tile
    constexpr int W = MT == 2 ? 128 : 64;           // hop ring of Xs: >= TH + PC - 1 slots
    __shared__ __attribute__((aligned(16))) float Hs[kPC * kHRow];
    __shared__ __attribute__((aligned(16))) float Xs[2 * W * kBS];

    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63;
    const int k = l >> 5, j = l & 31, c = j & 1;

    const int bb = blockIdx.x % a.binblocks;        // 16-bin block of the spectrum
    const int ks = blockIdx.x / a.binblocks;
    const int o0 = blockIdx.y * 16;
    const int t0 = blockIdx.z * TH;
    const int live_t = min(TH, a.T - t0);
    const long long h0 = a.h_first + t0;
    const int hmod = (int) (h0 % a.R);
    const int h0w = (int) (h0 & (W - 1));           // the ring index only needs h0 mod W

    const int K = a.nin * a.P;
    const int kb0 = ks * a.kper;
    const int kb1 = min(K, kb0 + a.kper);

    f32x16 acc[4][MT];
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[b][m][r] = 0.f;

    const long long pair_stride4 = (long long) a.Pcap * a.M2;
    const long long out_stride4 = (long long) a.nin_alloc * pair_stride4;
    const unsigned boff4 = (unsigned) bb * 8u;      // float4 offset of the bin block inside a spectrum

    // staging roles.  IR spectra: 256 rows (pp, o) of 8 float4 per chunk = 8 float4 per thread; 64 consecutive threads take 8 rows of
    // ONE partition and 8 consecutive outputs (the LDS rows of consecutive outputs start 8 banks apart: a conflict-free write).
    const int hq = tid & 7;
    const int hrow0 = tid >> 3;                     // + 32 r: row = o + 16 pp
    // input spectra: the PC hops a chunk gains, 8 float4 each: threads 0 .. 127
    const int xq = tid & 7, xr = tid >> 3;

    const bool fix0 = (bb == 0 && w == 0);          // this wave owns bin 0 = (DC, Nyquist)
    const unsigned sign = (k == 1 && c == 0) ? 0x80000000u : 0u;
    const int brow = (j ^ k) * kBS + 4 * w;         // B operand: component c ^ k of output j >> 1
    const int brow0 = ((j & ~1) + k) * kBS;         // bin 0: component k of output j >> 1, used where c == k
    const int arow = k * W * kBS + 4 * w;

    float4 hreg[8];
    float4 xreg;

    if (kb0 < kb1)
    {
        const int i_first = kb0 / a.P, i_last = (kb1 - 1) / a.P;
        for (int i = i_first; i <= i_last; i++)
        {
            const int pa = (i == i_first) ? kb0 - i_first * a.P : 0;
            const int pb = (i == i_last) ? kb1 - i_last * a.P : a.P;
            const float4 *xrow = a.X + (long long) i * a.R * a.M2 + boff4;
            const float4 *hbase = a.H + (long long) i * pair_stride4 + boff4;

            auto load_chunk = [&](int pc)
            {
#pragma unroll
                for (int r = 0; r < 8; r++)
                {
                    const int row = hrow0 + 32 * r;
                    const int o = row & 15, pp = row >> 4;
                    const int p = min(pc + pp, pb - 1);                 // (rows past the slice re-read its last partition; never used)
                    const float4 *src = hbase + (long long) min(o0 + o, a.nout - 1) * out_stride4 + (unsigned) p * (unsigned) a.M2 + hq;
                    hreg[r] = NT ? load_nt(src) : *src;
                }
                if (tid < 128)
                {
                    int slot = hmod - pc - xr;                          // hop h0 - pc - xr
                    slot %= a.R;
                    if (slot < 0) slot += a.R;
                    xreg = xrow[(unsigned) slot * (unsigned) a.M2 + xq];
                    if (h0 - pc - xr < a.hop_min) xreg = make_float4(0.f, 0.f, 0.f, 0.f);     // (before the stream's first hop: silence, whatever the ring holds)
                }
            };
            auto store_chunk = [&](int pc)
            {
#pragma unroll
                for (int r = 0; r < 8; r++)
                {
                    const int row = hrow0 + 32 * r;
                    const int o = row & 15, pp = row >> 4;
                    float *d = Hs + pp * kHRow + (2 * o) * kBS + 2 * hq;
                    *reinterpret_cast<float2 *>(d) = make_float2(hreg[r].x, hreg[r].z);
                    *reinterpret_cast<float2 *>(d + kBS) = make_float2(hreg[r].y, hreg[r].w);
                }
                if (tid < 128)
                {
                    const int s = (h0w - pc - xr) & (W - 1);
                    float *d = Xs + s * kBS + 2 * xq;
                    *reinterpret_cast<float2 *>(d) = make_float2(xreg.x, xreg.z);
                    *reinterpret_cast<float2 *>(d + W * kBS) = make_float2(xreg.y, xreg.w);
                }
            };

            // a new input: the hops above the first chunk's own, h0 - pa + 1 .. h0 - pa + TH - 1 (clamped to the launch's last hop:
            // rows of a ragged tile that do not exist are computed on it and never stored)
            __syncthreads();                        // (the previous input's last chunk has been read)
            for (int e = tid; e < (TH - 1) * 8; e += 256)
            {
                const int u = 1 + (e >> 3), q = e & 7;
                const int uc = min(u, max(live_t - 1, 0));
                int slot = (hmod - pa + uc) % a.R;
                if (slot < 0) slot += a.R;
                float4 v = xrow[(unsigned) slot * (unsigned) a.M2 + q];
                if (h0 - pa + uc < a.hop_min) v = make_float4(0.f, 0.f, 0.f, 0.f);
                const int s = (h0w - pa + u) & (W - 1);
                float *d = Xs + s * kBS + 2 * q;
                *reinterpret_cast<float2 *>(d) = make_float2(v.x, v.z);
                *reinterpret_cast<float2 *>(d + W * kBS) = make_float2(v.y, v.w);
            }
            load_chunk(pa);
            store_chunk(pa);
            __syncthreads();

            for (int pc = pa; pc < pb; pc += kPC)
            {
                const bool more = pc + kPC < pb;
                if (more) load_chunk(pc + kPC);     // in flight under this chunk's arithmetic

                const int n = min(kPC, pb - pc);
                for (int pp = 0; pp < n; pp++)
                {
                    float4 bv = *reinterpret_cast<const float4 *>(Hs + pp * kHRow + brow);
                    bv.x = __uint_as_float(__float_as_uint(bv.x) ^ sign);
                    bv.y = __uint_as_float(__float_as_uint(bv.y) ^ sign);
                    bv.z = __uint_as_float(__float_as_uint(bv.z) ^ sign);
                    bv.w = __uint_as_float(__float_as_uint(bv.w) ^ sign);
                    if (fix0)
                    {
                        const float b0 = Hs[pp * kHRow + brow0];
                        bv.x = (c == k) ? b0 : 0.f;
                    }
#pragma unroll
                    for (int m = 0; m < MT; m++)
                    {
                        const int s = (h0w + m * 32 + j - pc - pp) & (W - 1);
                        const float4 av = *reinterpret_cast<const float4 *>(Xs + arow + s * kBS);
                        acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[0][m], 0, 0, 0);
                        acc[1][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[1][m], 0, 0, 0);
                        acc[2][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[2][m], 0, 0, 0);
                        acc[3][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[3][m], 0, 0, 0);
                    }
                }
                if (more)
                {
                    __syncthreads();
                    store_chunk(pc + kPC);
                    __syncthreads();
                }
            }
        }
    }

    // C / D of the 32 x 32 forms: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int o = o0 + (j >> 1);
    if (o < a.nout)
    {
        float *y = reinterpret_cast<float *>(a.Y + (long long) ks * a.ks_stride4) + ((long long) o * a.M2 * 4 + (long long) (bb * kNB + 4 * w) * 2 + c);
        const long long tstride = (long long) a.nout * a.M2 * 4;
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int r = 0; r < 16; r++)
            {
                const int t = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * k;
                if (t < live_t)
                {
                    float *d = y + (long long) (t0 + t) * tstride;
                    d[0] = acc[0][m][r];
                    d[2] = acc[1][m][r];
                    d[4] = acc[2][m][r];
                    d[6] = acc[3][m][r];
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------ plan and launch

// the offline shapes: every partition live (the caller's `steady`), 32 hops or more, a matrix (the parallel mode's outputs share no input)
bool mac_mfma_applies(const MacShape &s)
{
    static const int mode = std::getenv("HCV_MAC_MFMA") ? std::atoi(std::getenv("HCV_MAC_MFMA")) : 1;      // 0 = off (A/B against the register tiles)
    if (!mode || !s.steady || s.diag) return false;
    return s.T >= 32 && s.M >= kNB && s.M % kNB == 0 && (long long) s.nin * s.P >= 16 && s.nout >= 2;
}

void mac_mfma_plan(const MacShape &s, MacPlan &pl)
{
    const int mt = s.T > 32 ? 2 : 1;
    const int th = 32 * mt;
    pl.mfma = mt;
    pl.ot = 16;
    pl.tt = th;
    pl.bx = 256;
    pl.by = 1;
    pl.tz = (s.T + th - 1) / th;
    pl.binblocks = s.M / kNB;
    pl.outtiles = (s.nout + 15) / 16;
    pl.inwg = 0;
    pl.nt = 1;
    // two workgroups of 60 KB of LDS per CU are resident at once: 512 per round.  k-slices only to fill that round (each slice is
    // another set of partial spectra through memory), and never shorter than two staged chunks
    const long long K = (long long) s.nin * s.P;
    const long long base = (long long) pl.binblocks * pl.outtiles * pl.tz;
    long long want = std::max<long long>(1, 512 / base);
    want = std::min<long long>(want, std::max<long long>(1, K / (2 * kPC)));
    if (s.max_ksplit > 0) want = std::min<long long>(want, s.max_ksplit);
    pl.kper = (int) ((K + want - 1) / want);
    pl.ksplit = (int) ((K + pl.kper - 1) / pl.kper);
}

hipError_t launch_mac_mfma(const MacPlan &pl, const MacParams &a, hipStream_t st)
{
    dim3 grid(pl.binblocks * pl.ksplit, pl.outtiles, pl.tz), block(256);
    if (pl.mfma == 2)
        hipLaunchKernelGGL((spectral_mac_mfma_kernel<2, true>), grid, block, 0, st, a);
    else
        hipLaunchKernelGGL((spectral_mac_mfma_kernel<1, true>), grid, block, 0, st, a);
    return hipGetLastError();
}

// (Engine::init, once per device: HIP loads a translation unit's code object at the first launch of one of its kernels — 0.3 - 0.8 ms on the
// calling thread, which for the kernels of a control section or a restart is the audio thread in mid-stream; asking for a kernel's attributes loads it now)
void preload_mac_mfma()
{
    hipFuncAttributes fa;
    (void) hipFuncGetAttributes(&fa, reinterpret_cast<const void *>((spectral_mac_mfma_kernel<2, true>)));
    (void) hipGetLastError();
}

} // namespace hcv
