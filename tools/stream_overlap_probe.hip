// stream_overlap_probe.hip -- do two HIP streams of one process run small kernels CONCURRENTLY on this box?
// (decides whether the multi-rank apply may put the boundary chains + exchange on a second stream)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_spin(long long cycles, int *sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (cycles < 0) sink[0] = 1;
}
static double run(hipStream_t a, hipStream_t b, int grid_a, int grid_b, long long cyc, int *sink)
{
    hipDeviceSynchronize();
    hipEvent_t e0, e1, ej;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreateWithFlags(&ej, hipEventDisableTiming);
    hipEventRecord(e0, a);
    hipStreamWaitEvent(b, e0, 0);
    hipLaunchKernelGGL(k_spin, dim3(grid_b), dim3(256), 0, b, cyc, sink);
    hipEventRecord(ej, b);
    hipLaunchKernelGGL(k_spin, dim3(grid_a), dim3(256), 0, a, cyc, sink);
    hipStreamWaitEvent(a, ej, 0);
    hipEventRecord(e1, a);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3;
}
int main()
{
    for (const char *v : {"GPU_MAX_HW_QUEUES", "HIP_LAUNCH_BLOCKING", "AMD_SERIALIZE_KERNEL", "HSA_ENABLE_SDMA", "ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "HIP_FORCE_DEV_KERNARG", "DEBUG_HIP_GRAPH_DOT_PRINT"}) {
        const char *e = getenv(v);
        printf("%s=%s\n", v, e ? e : "(unset)");
    }
    int *sink; hipMalloc((void **)&sink, 64);
    const long long cyc = 10000;   // wall_clock64 ticks at 100 MHz: 100 us
    hipStream_t s1, s2, sh;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithPriority(&sh, hipStreamNonBlocking, hi);
    printf("priority range least=%d greatest=%d\n", lo, hi);
    for (int rep = 0; rep < 2; ++rep) {
        printf("one kernel alone (2 WG)            : %7.1f us\n", run(s1, s1, 2, 2, cyc, sink) / 2);
        printf("null stream + nonblocking stream    : %7.1f us (100 = concurrent, 200 = serial)\n", run(nullptr, s1, 126, 2, cyc, sink));
        printf("two nonblocking streams             : %7.1f us\n", run(s1, s2, 126, 2, cyc, sink));
        printf("nonblocking + high-priority stream  : %7.1f us\n", run(s1, sh, 126, 2, cyc, sink));
        printf("null + high-priority stream         : %7.1f us\n", run(nullptr, sh, 126, 2, cyc, sink));
        printf("two streams, both 256 WG            : %7.1f us\n", run(s1, s2, 256, 256, cyc, sink));
    }
    return 0;
}
