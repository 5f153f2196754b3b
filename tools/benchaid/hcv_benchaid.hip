// BENCH AID, not part of the product: what THIS box streams from HBM when it does nothing else — a grid-stride sum over `bytes` of device
// memory with the loads the multiply-accumulate kernel streams its IR spectra with (nontemporal 16-byte loads, 768 workgroups of 256
// threads, eight loads in flight per thread), timed with HIP events, best of `reps`.  bench.py prints it beside the roofline fraction
// (boxes of the pool differ by several per cent).  Built by __graft_entry__.build() into tools/benchaid/libhcv_benchaid.so; nothing under
// hisstools_library_amd/ or include/ knows it.
#include <hip/hip_runtime.h>

namespace
{
    __global__ __launch_bounds__(256) void stream_read_kernel(const float4 *__restrict__ p, size_t n4, float *__restrict__ sink)
    {
        typedef float v4 __attribute__((ext_vector_type(4)));
        const size_t stride = (size_t) gridDim.x * blockDim.x;
        size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
        float acc = 0.f;
        for (; i + 7 * stride < n4; i += 8 * stride)
        {
            v4 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(p + i + k * stride));
#pragma unroll
            for (int k = 0; k < 8; k++) acc += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        }
        for (; i < n4; i += stride)
        {
            const v4 v = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(p + i));
            acc += (v.x + v.y) + (v.z + v.w);
        }
        if (acc == 123456.789f) *sink = acc;           // (keeps the loads alive; never true for a zeroed buffer)
    }
}

// GB/s into *gbs; 0 on success
extern "C" int hcv_benchaid_box_read_rate(int device, size_t bytes, int reps, double *gbs)
{
    if (!gbs || bytes < (1u << 20) || reps < 1 || device < 0) return -1;
    int prev_dev = 0;
    (void) hipGetDevice(&prev_dev);
    if (hipSetDevice(device) != hipSuccess)
    {
        (void) hipGetLastError();
        return -1;
    }
    void *buf = nullptr;
    float *sink = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    int rc = -1;
    float best = 1e30f;
    const size_t n4 = bytes / 16;
    if (hipMalloc(&buf, n4 * 16) != hipSuccess || hipMalloc(&sink, sizeof(float)) != hipSuccess) goto out;
    if (hipMemset(buf, 0, n4 * 16) != hipSuccess || hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) goto out;
    for (int r = 0; r < reps + 1; r++)
    {
        if (hipEventRecord(a, nullptr) != hipSuccess) goto out;
        hipLaunchKernelGGL(stream_read_kernel, dim3(768), dim3(256), 0, nullptr, static_cast<const float4 *>(buf), n4, sink);
        if (hipEventRecord(b, nullptr) != hipSuccess || hipEventSynchronize(b) != hipSuccess) goto out;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, a, b) != hipSuccess) goto out;
        if (r > 0 && ms < best) best = ms;              // (the first pass warms the TLBs)
    }
    *gbs = (double) (n4 * 16) / ((double) best * 1e-3) / 1e9;
    rc = 0;
out:
    if (a) (void) hipEventDestroy(a);
    if (b) (void) hipEventDestroy(b);
    if (buf) (void) hipFree(buf);
    if (sink) (void) hipFree(sink);
    if (rc != 0) (void) hipGetLastError();
    (void) hipSetDevice(prev_dev);
    return rc;
}
