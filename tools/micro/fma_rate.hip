// Issue rate of v_fma_f32 against v_pk_fma_f32 on gfx950: 16 independent accumulator chains per lane, 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/fma_rate.hip -o tools/micro/build/fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_scalar(float *out, float a, float b, int iters)
{
    float c[16];
    for (int j = 0; j < 16; j++) c[j] = threadIdx.x * 1e-3f + j;
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) c[j] = __builtin_fmaf(c[j], a, b);
    float s = 0;
    for (int j = 0; j < 16; j++) s += c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_packed(float *out, float a, float b, int iters)
{
    v2f c[8];
    for (int j = 0; j < 8; j++) c[j] = v2f{ threadIdx.x * 1e-3f + j, threadIdx.x * 2e-3f + j };
    const v2f va = { a, a * 1.0001f }, vb = { b, b * 0.9999f };
    for (int i = 0; i < iters; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) c[j] = __builtin_elementwise_fma(c[j], va, vb);
    float s = 0;
    for (int j = 0; j < 8; j++) s += c[j].x + c[j].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    float *out;
    hipMalloc(&out, sizeof(float) * 256 * 2048);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000, blocks = 2048;
    for (int which = 0; which < 2; which++)
        for (int rep = 0; rep < 2; rep++)
        {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(256), 0, 0, out, 0.999f, 0.001f, iters);
            else hipLaunchKernelGGL(k_packed, dim3(blocks), dim3(256), 0, 0, out, 0.999f, 0.001f, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flops = 2.0 * 16 * (double) iters * 256.0 * blocks;
            std::printf("%s: %.3f ms  %.1f TFLOP/s\n", which ? "v_pk_fma_f32" : "v_fma_f32   ", ms, flops / (ms * 1e-3) / 1e12);
        }
    return 0;
}
