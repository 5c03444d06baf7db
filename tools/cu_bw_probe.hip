// cu_bw_probe.hip -- how much HBM read bandwidth ONE compute unit can pull, as a function of the bytes it keeps in
// flight and of how many CUs stream at the same time.  Decides the prefetch depth of the sweep kernels when a rank owns
// fewer chains than the device has CUs (strong scaling: N/G rows per GPU).
//   hipcc -O3 --offload-arch=gfx950 tools/cu_bw_probe.hip -o gpurun_out/cu_bw_probe && gpurun_out/cu_bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

// every workgroup (NWV waves) streams one contiguous region front to back, DEPTH 16-byte loads in flight per lane
template <int DEPTH, int NWV>
__global__ __launch_bounds__(NWV * 64) void k_stream(const d2 *src, int64_t per_wg2, double *sink)
{
    const d2 *p = src + (int64_t)blockIdx.x * per_wg2 + threadIdx.x;
    constexpr int64_t STRIDE = NWV * 64;
    d2 a[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) a[k] = __builtin_nontemporal_load(p + k * STRIDE);
    double t0 = 0, t1 = 0;
    for (int64_t i = DEPTH * STRIDE; i + DEPTH * STRIDE <= per_wg2; i += DEPTH * STRIDE) {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            t0 += a[k].x; t1 += a[k].y;
            a[k] = __builtin_nontemporal_load(p + i + k * STRIDE);
        }
    }
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) { t0 += a[k].x; t1 += a[k].y; }
    if (t0 + t1 == 1.2345e300) sink[0] = t0;
}

template <int DEPTH, int NWV>
static double run(const d2 *buf, int64_t total2, int nwg, double *sink)
{
    int64_t per = (total2 / nwg) / (DEPTH * NWV * 64) * (DEPTH * NWV * 64);
    // same bytes per WORKGROUP for every grid size would make small grids short; keep each WG at <= 16 MiB
    const int64_t cap = (int64_t)(16 << 20) / 16;
    if (per > cap) per = cap / (DEPTH * NWV * 64) * (DEPTH * NWV * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_stream<DEPTH, NWV>), dim3(nwg), dim3(NWV * 64), 0, 0, buf, per, sink);
    hipEventRecord(e0, 0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_stream<DEPTH, NWV>), dim3(nwg), dim3(NWV * 64), 0, 0, buf, per, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)per * 16.0 * nwg * reps / (ms * 1e-3) / 1e9;
}

int main()
{
    const int64_t bytes = (int64_t)8 << 30;
    d2 *buf = nullptr;
    double *sink = nullptr;
    if (hipMalloc((void **)&buf, bytes) != hipSuccess || hipMalloc((void **)&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    printf("device: %s, %d CUs\n", pr.name, pr.multiProcessorCount);
    const int grids[] = {32, 64, 128, 192, 256, 384, 512, 1024};
    printf("%-28s", "waves/WG x loads in flight");
    for (int g : grids) printf(" %8d", g);
    printf("   (GB/s total; workgroups across)\n");
#define ROW(D, W)                                                                         \
    do {                                                                                  \
        printf("%2d waves x %2d x16B = %4d KiB ", W, D, W * 64 * D * 16 / 1024);         \
        for (int g : grids) printf(" %8.0f", run<D, W>(buf, bytes / 16, g, sink));        \
        printf("\n");                                                                     \
        fflush(stdout);                                                                   \
    } while (0)
    ROW(8, 4);
    ROW(16, 4);
    ROW(32, 4);
    ROW(48, 4);
    ROW(64, 4);
    ROW(16, 8);
    ROW(32, 8);
    ROW(16, 16);
    ROW(32, 16);
    ROW(8, 1);
    ROW(32, 1);
    hipFree(buf); hipFree(sink);
    return 0;
}
