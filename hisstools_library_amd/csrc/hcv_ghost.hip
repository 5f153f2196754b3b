// Exact per-pair restart while the other pairs keep running (MonoConvolve::reset / set on one (in,out) pair mid-stream,
// MonoConvolve.cpp:139-150 -> PartitionedConvolve.cpp:262-292, TimeDomainConvolve.cpp:91-98).
//
// The reference gives every pair private input buffers and a private output accumulator; restarting a pair at sample t0
// means (1) it never sees input older than t0 and (2) whatever it had still to deliver after t0 is dropped.  Here all pairs
// of an input share one ring of input spectra and all pairs of an output share one timeline, so a restart is made exact
// by two corrections instead:
//
//   ghost spectra   G0, G1 = transforms of the PRE-t0 part of the two frames that straddle t0 (frame h_r = floor(t0 / hop)
//                   and h_r + 1; a frame holds hops h-1 and h).  Linearity: what the restarted pair must not see is
//                   exactly H[p] * G, so every spectral_mac launch that reaches those frames is followed by
//                   ghost_mac_kernel, which subtracts  H[o][i][h - h_r] * G0 + H[o][i][h - h_r - 1] * G1  from the launch's
//                   accumulated spectra.  Two products per restarted pair and hop, for one IR length after the restart.
//   retiring        at t0 the timeline still holds the hop computed last (its result is emitted one hop later): the pair's
//                   share of it — a one-pair spectral_mac with the OLD spectra, before a set() overwrites them — is
//                   inverse-transformed and subtracted from the timeline for samples >= t0 (timeline_sub_kernel).
#include "hcv_kernels.h"

namespace hcv
{

// ring position j of the ghost history holds sample s in [t0 - Lg/2, t0 + Lg/2), s = j (mod Lg): the input's own samples
// for s < t0, zeros from t0 on
struct GhostRows
{
    int row[128];
};

__global__ __launch_bounds__(256) void ghost_hist_kernel(const float *__restrict__ hist, long long hist_stride, long long hist_mask, GhostRows rows,
                                                         float *__restrict__ ghost, long long Lg, long long t0)
{
    const long long j = blockIdx.x * 256LL + threadIdx.x;
    if (j >= Lg) return;
    const int r = blockIdx.y;
    const long long lo = t0 - Lg / 2;
    const long long s = lo + ((j - lo) & (Lg - 1));
    float v = 0.f;
    if (s >= 0 && s < t0) v = hist[(long long) rows.row[r] * hist_stride + (s & hist_mask)];
    ghost[(long long) r * Lg + j] = v;
}

hipError_t launch_ghost_hist(const float *hist, long long hist_stride, long long hist_mask, const int *rows, int nrows, float *ghost, long long Lg,
                             long long t0, hipStream_t st)
{
    for (int r0 = 0; r0 < nrows; r0 += 128)
    {
        GhostRows gr;
        const int n = nrows - r0 < 128 ? nrows - r0 : 128;
        for (int k = 0; k < n; k++) gr.row[k] = rows[r0 + k];
        dim3 grid((unsigned) ((Lg + 255) / 256), (unsigned) n);
        hipLaunchKernelGGL(ghost_hist_kernel, grid, dim3(256), 0, st, hist, hist_stride, hist_mask, gr, ghost + (long long) r0 * Lg, Lg, t0);
    }
    return hipGetLastError();
}

struct GhostMacParams
{
    const float4 *H;
    float4 *Y;
    const int *start;
    const GhostEntry *ent;
    GhostEntry single;
    int use_single;
    long long h_first;
    int M2, P, Pcap, T, nin, nin_alloc, nout;
};

// bins 2 b4 and 2 b4 + 1; bin 0 carries (DC, Nyquist): two real products (PartitionedConvolve.cpp:398-406)
__device__ __forceinline__ void ghost_cmac(float4 &acc, float &ny, const float4 x, const float4 h, bool bin0)
{
    acc.x += x.x * h.x - x.y * h.y;
    acc.y += x.x * h.y + x.y * h.x;
    acc.z += x.z * h.z - x.w * h.w;
    acc.w += x.z * h.w + x.w * h.z;
    if (bin0) ny += x.y * h.y;
}

__global__ __launch_bounds__(256) void ghost_mac_kernel(GhostMacParams a)
{
    const int b4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (b4 >= a.M2) return;
    const int o = blockIdx.y, t = blockIdx.z;
    const long long h = a.h_first + t;
    const int e0 = a.use_single ? 0 : a.start[o], e1 = a.use_single ? 1 : a.start[o + 1];
    const bool bin0 = b4 == 0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float ny = 0.f;
    bool any = false;
    for (int e = e0; e < e1; e++)
    {
        const GhostEntry en = a.use_single ? a.single : a.ent[e];
        if (en.i >= a.nin) continue;
        const float4 *hp = a.H + ((long long) o * a.nin_alloc + en.i) * a.Pcap * a.M2 + b4;
        const long long p0 = h - en.h_r, p1 = p0 - 1;
        if (p0 >= 0 && p0 < a.P)
        {
            ghost_cmac(acc, ny, en.g0[b4], hp[p0 * a.M2], bin0);
            any = true;
        }
        if (p1 >= 0 && p1 < a.P)
        {
            ghost_cmac(acc, ny, en.g1[b4], hp[p1 * a.M2], bin0);
            any = true;
        }
    }
    if (!any) return;
    if (bin0)
    {
        acc.x += ny;
        acc.y = ny;
    }
    float4 *y = a.Y + ((long long) t * a.nout + o) * a.M2 + b4;
    float4 v = *y;
    v.x -= acc.x;
    v.y -= acc.y;
    v.z -= acc.z;
    v.w -= acc.w;
    *y = v;
}

hipError_t launch_ghost_mac(const MacShape &s, const float2 *H, float2 *Y, long long h_first, const int *start, const GhostEntry *ent,
                            const GhostEntry *single, hipStream_t st)
{
    if (s.T <= 0 || s.nout <= 0) return hipSuccess;
    GhostMacParams a;
    a.H = reinterpret_cast<const float4 *>(H);
    a.Y = reinterpret_cast<float4 *>(Y);
    a.start = start;
    a.ent = ent;
    a.use_single = single ? 1 : 0;
    if (single) a.single = *single;
    else a.single = GhostEntry{ 0, nullptr, nullptr, 0, 0 };
    a.h_first = h_first;
    a.M2 = s.M / 2;
    a.P = s.P;
    a.Pcap = s.Pcap;
    a.T = s.T;
    a.nin = s.diag ? 1 : s.nin;
    a.nin_alloc = s.nin_alloc;
    a.nout = s.nout;
    const int bx = a.M2 < 256 ? a.M2 : 256;
    dim3 grid((unsigned) ((a.M2 + bx - 1) / bx), (unsigned) s.nout, (unsigned) s.T);
    hipLaunchKernelGGL(ghost_mac_kernel, grid, dim3(bx), 0, st, a);
    return hipGetLastError();
}

// row[(base + j) & mask] -= tmp[j] * scale for the samples at or after t_min
__global__ __launch_bounds__(256) void timeline_sub_kernel(float *__restrict__ row, long long mask, long long base, const float *__restrict__ tmp, int n,
                                                           float scale, long long t_min)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const long long s = base + j;
    if (s >= t_min) row[s & mask] -= tmp[j] * scale;
}

hipError_t launch_timeline_sub(float *row, long long mask, long long base, const float *tmp, int n, float scale, long long t_min, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(timeline_sub_kernel, dim3((n + 255) / 256), dim3(256), 0, st, row, mask, base, tmp, n, scale, t_min);
    return hipGetLastError();
}

// (Engine::init, once per device: HIP loads a translation unit's code object at the first launch of one of its kernels — 0.3 - 0.8 ms on the
// calling thread, which for the kernels of a control section or a restart is the audio thread in mid-stream; asking for a kernel's attributes loads it now)
void preload_ghost()
{
    hipFuncAttributes fa;
    (void) hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(ghost_mac_kernel));
    (void) hipGetLastError();
}

} // namespace hcv
