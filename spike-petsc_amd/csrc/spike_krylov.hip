// spike_krylov.hip -- device pieces of the Krylov caller of PCApply.
//
// Reference: the outer KSP of /root/reference/src/testbed2.c:125-128 with the options of
// src/makefile:18 (-ksp_type gmres).  In the reference these are PETSc Vec operations
// (VecMDot / VecMAXPY / VecNorm); here they are fused multi-vector kernels so that one
// Gram-Schmidt pass reads the new vector once instead of once per basis vector.
// All kernels are HBM-bound streaming kernels: lane = row, 8-byte coalesced accesses.
#include "spike_internal.h"

namespace spike {

// ---- deterministic grid reductions ---------------------------------------------------------------------------------
// Every workgroup writes its NV partial sums to part[block][NV]; a one-workgroup kernel then adds the partials of all
// workgroups in a fixed order and writes out[].  No floating-point atomics: a dot product has the same bits in every
// run (the Krylov iteration is reproducible), and one launch pair serves the whole Gram-Schmidt column.  The kernel
// boundary is the only synchronisation: a "last workgroup finishes" ticket needs device-scope fences, which on this
// chip write back / invalidate the XCD's L2 once per workgroup (measured: 2-3x slower kernels).
constexpr int RED_BLOCKS = 2048;  // grid of the reduction kernels (8 workgroups = 32 waves per CU: every wave resident)
constexpr int RED_MAXV = 32;      // vectors per launch

// rows per workgroup when `grid` workgroups of 256 threads split n rows into contiguous runs (a multiple of 256)
__host__ __device__ inline int64_t block_chunk(int64_t n, int64_t grid) { return (((n + grid - 1) / grid) + 255) & ~(int64_t)255; }

template <int NV>
__device__ __forceinline__ void block_partials(double (&acc)[NV], double *part)
{
    __shared__ double red[4][NV];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double a = acc[i];
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o);
        if (lane == 0) red[wv][i] = a;
    }
    __syncthreads();
    if (threadIdx.x < NV)
        part[(int64_t)blockIdx.x * NV + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// all NV sums at once: thread t owns vector t % NV and every (1024/NV)-th workgroup's partial (coalesced reads, fixed
// order), then a fixed tree over the 1024/NV chunks
template <int NV>
__global__ __launch_bounds__(1024) void k_reduce_final(const double *part, int nblocks, int nout, double *out)
{
    __shared__ double fin[1024];
    constexpr int NCH = 1024 / NV;
    const int i = threadIdx.x % NV, c = threadIdx.x / NV;
    // fixed order, eight loads in flight: a = (((p0 + p1) + p2) + ...) exactly as a rolled loop would add them
    double a = 0.0;
    int b = c;
    for (; b + 7 * NCH < nblocks; b += 8 * NCH) {
        double v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = part[(int64_t)(b + e * NCH) * NV + i];
#pragma unroll
        for (int e = 0; e < 8; ++e) a += v[e];
    }
    for (; b < nblocks; b += NCH) a += part[(int64_t)b * NV + i];
    fin[threadIdx.x] = a;
    __syncthreads();
#pragma unroll
    for (int o = NCH / 2; o > 0; o >>= 1) {
        if (c < o) fin[threadIdx.x] += fin[threadIdx.x + o * NV];
        __syncthreads();
    }
    if (c == 0 && i < nout) out[i] = fin[i];
}

// out[i] = V_i . w   for i < nvec <= NV; w is read once for all vectors
template <int NV>
__global__ __launch_bounds__(256) void k_dots(const double *V, int64_t ldv, int nvec, const double *w, int64_t n,
                                              double *part)
{
    double acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.0;
    // a workgroup owns ONE contiguous run of rows of every vector (not a grid-stride comb): consecutive trips touch
    // adjacent 2-KiB segments, so a CU works inside a few pages per vector instead of a new page per vector per trip
    constexpr int U = NV <= 8 ? 4 : (NV <= 16 ? 2 : 1);  // rows per lane and trip: keeps >= 32 loads in flight per lane
    const int64_t chunk = block_chunk(n, gridDim.x);
    const int64_t end = (blockIdx.x + 1) * chunk < n ? (blockIdx.x + 1) * chunk : n;
    // every load is unconditional (out-of-range rows / vectors are clamped to a valid address and weighted by zero, or
    // land in accumulators nobody reads): loads under a branch are not batched by the compiler, and a lane that waits for
    // each one separately leaves the memory system idle
    for (int64_t r0 = blockIdx.x * chunk + threadIdx.x; r0 < end; r0 += U * 256) {
        double wv[U];
        int64_t rr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool in = r0 + u * 256 < end;
            rr[u] = in ? r0 + u * 256 : r0;
            wv[u] = w[rr[u]];
            if (!in) wv[u] = 0.0;
        }
        double xv[NV][U];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int64_t base = (int64_t)(i < nvec ? i : 0) * ldv;
#pragma unroll
            for (int u = 0; u < U; ++u) xv[i][u] = V[base + rr[u]];
        }
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int u = 0; u < U; ++u) acc[i] = fma(xv[i][u], wv[u], acc[i]);
    }
    block_partials<NV>(acc, part);
}

static int red_grid(int64_t n)
{
    int grid = (int)((n + 511) / 512);
    return grid > RED_BLOCKS ? RED_BLOCKS : (grid < 1 ? 1 : grid);
}

// ws: RED_BLOCKS*RED_MAXV doubles of partials
size_t red_workspace_doubles() { return (size_t)RED_BLOCKS * RED_MAXV; }

template <int NV>
static void launch_dots_t(const double *V, int64_t ldv, int nv, const double *w, int64_t n, double *out, double *ws,
                          hipStream_t st)
{
    const int grid = red_grid(n);
    hipLaunchKernelGGL((k_dots<NV>), dim3(grid), dim3(256), 0, st, V, ldv, nv, w, n, ws);
    hipLaunchKernelGGL((k_reduce_final<NV>), dim3(1), dim3(1024), 0, st, ws, grid, nv, out);
}

hipError_t launch_dots(const double *V, int64_t ldv, int nvec, const double *w, int64_t n, double *out, double *ws,
                       hipStream_t st)
{
    for (int v0 = 0; v0 < nvec; v0 += RED_MAXV) {
        const int nv = nvec - v0 < RED_MAXV ? nvec - v0 : RED_MAXV;
        const double *Vp = V + (int64_t)v0 * ldv;
        if (nv <= 4) launch_dots_t<4>(Vp, ldv, nv, w, n, out + v0, ws, st);
        else if (nv <= 8) launch_dots_t<8>(Vp, ldv, nv, w, n, out + v0, ws, st);
        else if (nv <= 16) launch_dots_t<16>(Vp, ldv, nv, w, n, out + v0, ws, st);
        else launch_dots_t<32>(Vp, ldv, nv, w, n, out + v0, ws, st);
    }
    return hipGetLastError();
}

// w += sign * sum_i coef[i] V_i ; NORM: also norm2_out[0] = |w_new|^2 (deterministic grid reduction), so the norm of
// the orthogonalised Krylov vector costs no extra pass over it (PETSc: VecMAXPY followed by VecNorm)
template <bool NORM>
__global__ __launch_bounds__(256) void k_axpys(const double *V, int64_t ldv, int nvec, const double *coef, double *w,
                                               int64_t n, double sign, double *part)
{
    __shared__ double cf[64];
    for (int i = threadIdx.x; i < nvec && i < 64; i += blockDim.x) cf[i] = coef[i];
    __syncthreads();
    double acc[1] = {0.0};
    const int64_t chunk = block_chunk(n, gridDim.x);
    const int64_t end = (blockIdx.x + 1) * chunk < n ? (blockIdx.x + 1) * chunk : n;
    for (int64_t r = blockIdx.x * chunk + threadIdx.x; r < end; r += 512) {
        const bool two = r + 256 < end;
        const int64_t r1 = two ? r + 256 : r;   // clamped: the loads below stay unconditional
        double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
        for (int i = 0; i < nvec; ++i) {
            s0 = fma(cf[i], V[(int64_t)i * ldv + r], s0);
            s1 = fma(cf[i], V[(int64_t)i * ldv + r1], s1);
        }
        const double v0 = w[r] + sign * s0;
        const double v1 = w[r1] + sign * s1;
        w[r] = v0;
        if (NORM) acc[0] = fma(v0, v0, acc[0]);
        if (two) {
            w[r1] = v1;
            if (NORM) acc[0] = fma(v1, v1, acc[0]);
        }
    }
    if (NORM) block_partials<1>(acc, part);
}

hipError_t launch_axpys(const double *V, int64_t ldv, int nvec, const double *coef, double *w, int64_t n, double sign,
                        hipStream_t st)
{
    if (nvec <= 0) return hipSuccess;
    if (nvec > 64) return hipErrorInvalidValue;
    const int grid = red_grid(n);
    hipLaunchKernelGGL((k_axpys<false>), dim3(grid), dim3(256), 0, st, V, ldv, nvec, coef, w, n, sign, (double *)nullptr);
    return hipGetLastError();
}

hipError_t launch_axpys_norm(const double *V, int64_t ldv, int nvec, const double *coef, double *w, int64_t n, double sign,
                             double *norm2_out, double *ws, hipStream_t st)
{
    if (nvec <= 0 || nvec > 64) return hipErrorInvalidValue;
    const int grid = red_grid(n);
    hipLaunchKernelGGL((k_axpys<true>), dim3(grid), dim3(256), 0, st, V, ldv, nvec, coef, w, n, sign, ws);
    hipLaunchKernelGGL((k_reduce_final<1>), dim3(1), dim3(1024), 0, st, ws, grid, 1, norm2_out);
    return hipGetLastError();
}

hipError_t launch_lincomb(const double *V, int64_t ldv, int nvec, const double *y_dev, double *x, int64_t n,
                          hipStream_t st)
{
    return launch_axpys(V, ldv, nvec, y_dev, x, n, 1.0, st);
}

// out = w / sqrt(scal[0])   (invert != 0)   or   out = w * scal[0]
__global__ void k_scale_copy(const double *w, const double *scal, int invert, double *out, int64_t n)
{
    const double s = invert ? 1.0 / sqrt(scal[0]) : scal[0];
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x)
        out[r] = w[r] * s;
}

hipError_t launch_scale_copy(const double *w, const double *scal_dev, int invert, double *out, int64_t n,
                             hipStream_t st)
{
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_scale_copy, dim3(grid), dim3(256), 0, st, w, scal_dev, invert, out, n);
    return hipGetLastError();
}

// w *= s, or (norm2 != nullptr) w *= 1/sqrt(norm2[0]) with the squared norm read on the device (0 -> no scaling)
__global__ void k_scale_value(double *w, double s, const double *norm2, int64_t n)
{
    if (norm2 != nullptr) { const double q = norm2[0]; s = q > 0.0 ? 1.0 / sqrt(q) : 1.0; }
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) w[r] *= s;
}

hipError_t launch_scale_value(double *w, double s, int64_t n, hipStream_t st, const double *norm2_dev)
{
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_scale_value, dim3(grid), dim3(256), 0, st, w, s, norm2_dev, n);
    return hipGetLastError();
}

__global__ void k_residual(const double *b, const double *ax, double *r, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        r[i] = b[i] - ax[i];
}

hipError_t launch_residual(const double *b, const double *ax, double *r, int64_t n, hipStream_t st)
{
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_residual, dim3(grid), dim3(256), 0, st, b, ax, r, n);
    return hipGetLastError();
}

// y = A x for a CSR operator (the un-banded A of /root/reference/src/testbed2.c:125-128).  TPR lanes share a row
// (TPR = 1, 4, 16, 64 picked from the mean row length) so that short circuit-matrix rows do not idle a wave.
template <int TPR>
__global__ __launch_bounds__(256) void k_csr_matvec(int64_t n, const int64_t *ia, const int32_t *ja, const double *a,
                                                    const double *x, double *y)
{
    const int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / TPR;
    const int sub = threadIdx.x % TPR;
    double s = 0.0;
    if (row < n) {
        const int64_t e = ia[row + 1];
        for (int64_t k = ia[row] + sub; k < e; k += TPR) s = fma(a[k], x[ja[k]], s);
    }
#pragma unroll
    for (int o = TPR / 2; o > 0; o >>= 1) s += __shfl_down(s, o, TPR);
    if (row < n && sub == 0) y[row] = s;
}

hipError_t launch_csr_matvec(int64_t n, const int64_t *ia, const int32_t *ja, const double *a, int tpr, const double *x,
                             double *y, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((n * tpr + 255) / 256);
    switch (tpr) {
    case 1: hipLaunchKernelGGL((k_csr_matvec<1>), dim3(grid), dim3(256), 0, st, n, ia, ja, a, x, y); break;
    case 4: hipLaunchKernelGGL((k_csr_matvec<4>), dim3(grid), dim3(256), 0, st, n, ia, ja, a, x, y); break;
    case 16: hipLaunchKernelGGL((k_csr_matvec<16>), dim3(grid), dim3(256), 0, st, n, ia, ja, a, x, y); break;
    default: hipLaunchKernelGGL((k_csr_matvec<64>), dim3(grid), dim3(256), 0, st, n, ia, ja, a, x, y); break;
    }
    return hipGetLastError();
}

// CSR -> diagonal-major band on the device (spike_setup_csr): lane = nonzero within a row group, rows by TPR lanes;
// entries outside |c - r| <= K are dropped, duplicates add (as MatSetValues(ADD) would).
// Local rows [row0, row0 + n) of a row-block-distributed matrix.  ja holds the column RELATIVE to the rank's first row
// (global column - row0, clamped into int32 by the host: an in-band entry lies in [-K, n + K], whatever n_global is).
// One thread per row, entries in storage order, plain stores: a repeated (row, column) pair keeps the LAST value -- the
// INSERT_VALUES semantics of the reference's MatSetValues (src/matbanded.c:98) -- and the result is the same bits every run.
__global__ __launch_bounds__(256) void k_csr_to_band(int64_t n, const int64_t *ia, const int32_t *ja, const double *a, int K,
                                                     double *band, int64_t ld)
{
    const int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (row >= n) return;
    for (int64_t k = ia[row]; k < ia[row + 1]; ++k) {
        const int64_t d = (int64_t)ja[k] - row + K;
        if (d >= 0 && d <= 2 * K) band[d * ld + row] = a[k];
    }
}

hipError_t launch_csr_to_band(int64_t n, const int64_t *ia, const int32_t *ja, const double *a, int K, double *band,
                              int64_t ld, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(band, 0, sizeof(double) * (size_t)(2 * K + 1) * (size_t)ld, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_csr_to_band, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, ia, ja, a, K, band, ld);
    return hipGetLastError();
}

}  // namespace spike
