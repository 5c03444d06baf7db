// spike_krylov.hip -- device pieces of the Krylov caller of PCApply.
//
// Reference: the outer KSP of /root/reference/src/testbed2.c:125-128 with the options of
// src/makefile:18 (-ksp_type gmres).  In the reference these are PETSc Vec operations
// (VecMDot / VecMAXPY / VecNorm); here they are fused multi-vector kernels so that one
// Gram-Schmidt pass reads the new vector once instead of once per basis vector.
// All kernels are HBM-bound streaming kernels: lane = row, 8-byte coalesced accesses.
#include "spike_internal.h"

namespace spike {

// out[v0+i] += V_{v0+i} . w   for i < NV (masked by nvec); one atomic per wave per vector
template <int NV>
__global__ __launch_bounds__(256) void k_dots(const double *V, int64_t ldv, int v0, int nvec, const double *w, int64_t n,
                                              double *out)
{
    double acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.0;
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const double wv = w[r];
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (v0 + i < nvec) acc[i] = fma(V[(int64_t)(v0 + i) * ldv + r], wv, acc[i]);
    }
    __shared__ double red[4][NV];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double a = acc[i];
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o);
        if (lane == 0) red[wv][i] = a;
    }
    __syncthreads();
    if (threadIdx.x < NV && v0 + (int)threadIdx.x < nvec) {
        const double s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(out + v0 + threadIdx.x, s);
    }
}

hipError_t launch_dots(const double *V, int64_t ldv, int nvec, const double *w, int64_t n, double *out, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(out, 0, sizeof(double) * nvec, st);
    if (e != hipSuccess) return e;
    int grid = (int)((n + 1023) / 1024);
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    for (int v0 = 0; v0 < nvec; v0 += 8)
        hipLaunchKernelGGL((k_dots<8>), dim3(grid), dim3(256), 0, st, V, ldv, v0, nvec, w, n, out);
    return hipGetLastError();
}

// w += sign * sum_i coef[i] V_i
__global__ __launch_bounds__(256) void k_axpys(const double *V, int64_t ldv, int nvec, const double *coef, double *w,
                                               int64_t n, double sign)
{
    __shared__ double cf[64];
    for (int i = threadIdx.x; i < nvec && i < 64; i += blockDim.x) cf[i] = coef[i];
    __syncthreads();
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int i = 0; i < nvec; ++i) s = fma(cf[i], V[(int64_t)i * ldv + r], s);
        w[r] += sign * s;
    }
}

hipError_t launch_axpys(const double *V, int64_t ldv, int nvec, const double *coef, double *w, int64_t n, double sign,
                        hipStream_t st)
{
    if (nvec <= 0) return hipSuccess;
    if (nvec > 64) return hipErrorInvalidValue;
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_axpys, dim3(grid), dim3(256), 0, st, V, ldv, nvec, coef, w, n, sign);
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

__global__ void k_scale_value(double *w, double s, int64_t n)
{
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) w[r] *= s;
}

hipError_t launch_scale_value(double *w, double s, int64_t n, hipStream_t st)
{
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_scale_value, dim3(grid), dim3(256), 0, st, w, s, n);
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
__global__ __launch_bounds__(256) void k_csr_to_band(int64_t n, const int64_t *ia, const int32_t *ja, const double *a, int K,
                                                     double *band, int64_t ld)
{
    const int64_t row = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / 4;
    const int sub = threadIdx.x % 4;
    if (row >= n) return;
    for (int64_t k = ia[row] + sub; k < ia[row + 1]; k += 4) {
        const int64_t d = (int64_t)ja[k] - row + K;
        if (d >= 0 && d <= 2 * K) atomicAdd(band + d * ld + row, a[k]);
    }
}

hipError_t launch_csr_to_band(int64_t n, const int64_t *ia, const int32_t *ja, const double *a, int K, double *band,
                              int64_t ld, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(band, 0, sizeof(double) * (size_t)(2 * K + 1) * (size_t)ld, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_csr_to_band, dim3((unsigned)((n * 4 + 255) / 256)), dim3(256), 0, st, n, ia, ja, a, K, band, ld);
    return hipGetLastError();
}

}  // namespace spike
