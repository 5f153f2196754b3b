// Host-side cost of the HIP calls the engine issues per block (build: hipcc --offload-arch=gfx950 -O2 api_cost.hip -o api_cost)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void nop(int *p) { if (p) *p = 1; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    hipEvent_t e, f;
    hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipEventCreateWithFlags(&f, hipEventDisableTiming);
    const int N = 2000;
    for (int k = 0; k < 100; k++) hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, a, nullptr);
    hipDeviceSynchronize();
    double t0 = now();
    for (int k = 0; k < N; k++) hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, a, nullptr);
    double t1 = now();
    hipDeviceSynchronize();
    std::printf("launch (one stream)              %.2f us/call (gpu drain %.2f us/launch)\n", (t1 - t0) / N, (now() - t0) / N);
    t0 = now();
    for (int k = 0; k < N; k++) hipEventRecord(e, a);
    t1 = now();
    hipDeviceSynchronize();
    std::printf("event record                     %.2f us/call\n", (t1 - t0) / N);
    t0 = now();
    for (int k = 0; k < N; k++) hipStreamWaitEvent(a, e, 0);
    t1 = now();
    hipDeviceSynchronize();
    std::printf("wait, same stream, done event    %.2f us/call\n", (t1 - t0) / N);
    t0 = now();
    for (int k = 0; k < N; k++) hipStreamWaitEvent(b, e, 0);
    t1 = now();
    hipDeviceSynchronize();
    std::printf("wait, other stream, done event   %.2f us/call\n", (t1 - t0) / N);
    t0 = now();
    for (int k = 0; k < N; k++)
    {
        hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, a, nullptr);
        hipEventRecord(e, a);
        hipStreamWaitEvent(b, e, 0);
        hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, b, nullptr);
        hipEventRecord(f, b);
        hipStreamWaitEvent(a, f, 0);
    }
    t1 = now();
    hipDeviceSynchronize();
    std::printf("ping-pong a->b->a (2 launches, 2 records, 2 waits)  %.2f us/round host, %.2f us/round total\n", (t1 - t0) / N, (now() - t0) / N);
    t0 = now();
    for (int k = 0; k < N; k++)
    {
        hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, a, nullptr);
        hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, a, nullptr);
    }
    t1 = now();
    hipDeviceSynchronize();
    std::printf("same two launches on one stream  %.2f us/round host, %.2f us/round total\n", (t1 - t0) / N, (now() - t0) / N);
    return 0;
}
