// Launch parameters of the spectral multiply-accumulate kernels (hcv_mac.hip, hcv_mac_tiled.hip).  Internal.
#pragma once

#include <hip/hip_runtime.h>

#include "hcv_kernels.h"

namespace hcv
{

struct MacParams
{
    const float4 *X;        // [nin][R][M/2] float4
    const float4 *H;        // [nout][nin_alloc][Pcap][M/2] float4
    float4 *Y;              // [ksplit][T][nout][M/2] float4
    const long long *hv;    // [nout][nin_alloc]
    long long h_first;
    int M2;                 // float4 per spectrum = M/2
    int R, P, Pcap, T;
    int nin, nin_alloc, nout;
    int diag;               // parallel mode: output o reads input o only (nin == 1 logically)
    int ksplit, kper;       // k-slices over blockIdx.x and their length
    int binblocks;
    long long ks_stride4;   // float4 stride between k-slices of Y
    int pin;                // >= 0: grid.x is 8 x the workgroups and only blockIdx.x % 8 == pin work (hcv_kernels.h: xcd_pin_for)
    long long hop_min;      // (hcv_mac_mfma.hip) input hops before this one are staged as zeros
};

// the software-pipelined hop-tiled kernel (hcv_mac_tiled.hip): (OT, TT) in {1, 4} x {2, 4, 8}
hipError_t launch_mac_tiled(int ot, int tt, bool nt, dim3 grid, dim3 block, const MacParams &a, hipStream_t st);

// the offline kernel on the matrix cores (hcv_mac_mfma.hip); a.binblocks = 16-bin blocks per spectrum
hipError_t launch_mac_mfma(const MacPlan &pl, const MacParams &a, hipStream_t st);

} // namespace hcv
