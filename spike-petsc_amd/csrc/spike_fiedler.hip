// spike_fiedler.hip -- the floating-point part of the Fiedler ordering on the device: LOBPCG refinement of one multilevel
// level (reference slot MatGetOrdering_Fiedler, /root/reference/src/petsc_mat_fiedler.c:11-58, whose
// HSL_MC73 is absent: the build publishes its own spec, csrc/host/fiedler.c).
//
// BIT-EXACT WITH THE HOST: this file and fiedler.c:refine() execute the same IEEE fp64 operations in the same order --
//   * element-wise updates: one multiply, one add/subtract, one divide at a time (no fused multiply-add: contraction is
//     switched off here; the host file is compiled with -ffp-contract=off);
//   * the Laplacian product: one thread per row, the row's terms subtracted in storage order, as the host loop;
//   * every sum (dot products, the mean of deflate, the residual norm) in THE REDUCTION ORDER of the spec: the index range
//     is cut into chunks of 1024; inside a chunk 256 slots, slot t = ((v[t] + v[t+256]) + v[t+512]) + v[t+768]
//     (absent elements count as +0.0), then the binary tree s[t] += s[t+o], o = 128, 64, ..., 1; the chunk sums are added
//     in chunk order.  The device computes the chunk sums (one workgroup per chunk), the host adds them -- in both builds;
//   * square roots, the 3 x 3 Rayleigh-Ritz eigenproblem and every branch are evaluated on the host from those scalars,
//     by the same code (fiedler.c:refine_core), whichever side holds the vectors.
// So the permutation does not depend on where a level is refined (tests/test_host_gpu.py compares them bit for bit).
#include "../../include/spike_mi355.h"
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace {

constexpr int CH = 1024;   // chunk of the reduction order
constexpr int MAXD = 6;    // dot products per launch

struct DotArgs {
    const double *a[MAXD];
    const double *b[MAXD];
    int nd;
};

// chunk sums of up to MAXD dot products: out[j * nchunks + chunk]
__global__ __launch_bounds__(256) void k_fd_dots(DotArgs d, int64_t n, int64_t nchunks, double *out)
{
#pragma clang fp contract(off)
    __shared__ double s[256];
    const int t = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * CH;
    for (int j = 0; j < d.nd; ++j) {
        double v = 0.0;
        for (int q = 0; q < 4; ++q) {
            const int64_t i = c0 + t + 256 * q;
            const double p = (i < n) ? d.a[j][i] * d.b[j][i] : 0.0;
            v = (q == 0) ? p : v + p;
        }
        s[t] = v;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (t < o) s[t] += s[t + o];
            __syncthreads();
        }
        if (t == 0) out[(int64_t)j * nchunks + blockIdx.x] = s[0];
        __syncthreads();
    }
}

// w = Lx - rho x (element-wise), chunk sums of w_i^2
__global__ __launch_bounds__(256) void k_fd_resid(int64_t n, double rho, const double *x, const double *Lx, double *w, double *out)
{
#pragma clang fp contract(off)
    __shared__ double s[256];
    const int t = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * CH;
    double v = 0.0;
    for (int q = 0; q < 4; ++q) {
        const int64_t i = c0 + t + 256 * q;
        double p = 0.0;
        if (i < n) {
            const double r = Lx[i] - rho * x[i];
            w[i] = r;
            p = r * r;
        }
        v = (q == 0) ? p : v + p;
    }
    s[t] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) s[t] += s[t + o];
        __syncthreads();
    }
    if (t == 0) out[blockIdx.x] = s[0];
}

// w_i /= (deg_i > 0 ? deg_i : 1), chunk sums of the result (the mean deflate() subtracts)
__global__ __launch_bounds__(256) void k_fd_precond(int64_t n, const double *deg, double *w, double *out)
{
#pragma clang fp contract(off)
    __shared__ double s[256];
    const int t = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * CH;
    double v = 0.0;
    for (int q = 0; q < 4; ++q) {
        const int64_t i = c0 + t + 256 * q;
        double p = 0.0;
        if (i < n) {
            const double dg = deg[i];
            p = w[i] / (dg > 0 ? dg : 1.0);
            w[i] = p;
        }
        v = (q == 0) ? p : v + p;
    }
    s[t] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) s[t] += s[t + o];
        __syncthreads();
    }
    if (t == 0) out[blockIdx.x] = s[0];
}

__global__ void k_fd_shift(int64_t n, double m, double *w)
{
#pragma clang fp contract(off)
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) w[i] -= m;
}

// y -= a x  (and, when y2 != null, y2 -= a x2)
__global__ void k_fd_axpy(int64_t n, double a, const double *x, double *y, const double *x2, double *y2)
{
#pragma clang fp contract(off)
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    y[i] -= a * x[i];
    if (y2 != nullptr) y2[i] -= a * x2[i];
}

// y /= s (and y2 /= s)
__global__ void k_fd_div(int64_t n, double sc, double *y, double *y2)
{
#pragma clang fp contract(off)
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    y[i] /= sc;
    if (y2 != nullptr) y2[i] /= sc;
}

__global__ void k_fd_lap(int64_t n, const int64_t *xadj, const int32_t *adj, const double *w, const double *deg, const double *x,
                         double *y)
{
#pragma clang fp contract(off)
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = deg[i] * x[i];
    for (int64_t k = xadj[i]; k < xadj[i + 1]; ++k) s -= w[k] * x[adj[k]];
    y[i] = s;
}

// the Rayleigh-Ritz update of fiedler.c:refine_core: x <- c0 x + c1 w + c2 p, p <- c1 w + c2 p, same for the L-images
__global__ void k_fd_update(int64_t n, double c0, double c1, double c2, int havep, double *x, double *Lx, const double *w,
                            const double *Lw, double *p, double *Lp)
{
#pragma clang fp contract(off)
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double pn = c1 * w[i] + (havep ? c2 * p[i] : 0.0);
    const double Lpn = c1 * Lw[i] + (havep ? c2 * Lp[i] : 0.0);
    x[i] = c0 * x[i] + pn;
    Lx[i] = c0 * Lx[i] + Lpn;
    p[i] = pn;
    Lp[i] = Lpn;
}

__global__ void k_fd_fill_one(int64_t n, double *x)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) x[i] = 1.0;
}

__global__ void k_fd_fill_alt(int64_t n, double *x)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) x[i] = (double)(i % 2 ? 1 : -1);
}

}  // namespace

// ---- one level's vectors on the device; operations named after the host loop's statements ---------------------------------
struct spike_fd_ctx {
    int64_t n = 0, nchunks = 0;
    int64_t *xadj = nullptr;
    int32_t *adj = nullptr;
    double *w_e = nullptr, *deg = nullptr;
    double *v[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // x, Lx, w, Lw, p, Lp, the constant 1
    double *dchunk = nullptr;
    double *hchunk = nullptr;   // pinned
    hipStream_t st = nullptr;
};

#define FDCHK(call) do { if ((call) != hipSuccess) return SPIKE_ERR_HIP; } while (0)

extern "C" int spike_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

extern "C" int spike_fd_destroy(spike_fd_ctx *c)
{
    if (!c) return SPIKE_OK;
    (void)hipStreamSynchronize(c->st);
    (void)hipFree(c->xadj); (void)hipFree(c->adj); (void)hipFree(c->w_e); (void)hipFree(c->deg); (void)hipFree(c->dchunk);
    for (double *q : c->v) (void)hipFree(q);
    if (c->hchunk) (void)hipHostFree(c->hchunk);
    delete c;
    return SPIKE_OK;
}

// graph (host CSR of the level's Laplacian: adjacency, edge weights, weighted degrees) and the start vector x
extern "C" int spike_fd_create(int64_t n, const int64_t *xadj, const int64_t *adj, const double *w, const double *deg,
                               const double *x0, spike_fd_ctx **out)
{
    if (!out || n <= 0 || n > 2000000000LL || !xadj || !adj || !w || !deg || !x0) return SPIKE_ERR_ARG;
    *out = nullptr;
    spike_fd_ctx *c = new spike_fd_ctx();
    c->n = n;
    c->nchunks = (n + CH - 1) / CH;
    const int64_t ne = xadj[n];
    int32_t *a32 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ne > 0 ? ne : 1));
    if (!a32) { delete c; return SPIKE_ERR_NOMEM; }
    for (int64_t k = 0; k < ne; ++k) a32[k] = (int32_t)adj[k];
    bool ok = hipMalloc((void **)&c->xadj, sizeof(int64_t) * (size_t)(n + 1)) == hipSuccess &&
              hipMalloc((void **)&c->adj, sizeof(int32_t) * (size_t)(ne > 0 ? ne : 1)) == hipSuccess &&
              hipMalloc((void **)&c->w_e, sizeof(double) * (size_t)(ne > 0 ? ne : 1)) == hipSuccess &&
              hipMalloc((void **)&c->deg, sizeof(double) * (size_t)n) == hipSuccess &&
              hipMalloc((void **)&c->dchunk, sizeof(double) * (size_t)(MAXD * c->nchunks)) == hipSuccess &&
              hipHostMalloc((void **)&c->hchunk, sizeof(double) * (size_t)(MAXD * c->nchunks), hipHostMallocDefault) == hipSuccess;
    for (int q = 0; q < 7 && ok; ++q) ok = hipMalloc((void **)&c->v[q], sizeof(double) * (size_t)n) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_fd_fill_one, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, n, c->v[6]);
        ok = hipGetLastError() == hipSuccess;
    }
    if (ok) {
        ok = hipMemcpy(c->xadj, xadj, sizeof(int64_t) * (size_t)(n + 1), hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(c->adj, a32, sizeof(int32_t) * (size_t)ne, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(c->w_e, w, sizeof(double) * (size_t)ne, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(c->deg, deg, sizeof(double) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(c->v[0], x0, sizeof(double) * (size_t)n, hipMemcpyHostToDevice) == hipSuccess;
    }
    free(a32);
    if (!ok) { (void)hipGetLastError(); spike_fd_destroy(c); return SPIKE_ERR_HIP; }
    *out = c;
    return SPIKE_OK;
}

static inline dim3 grid1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

// sums[j] = sum_i a_j[i] * b_j[i] in the reduction order of the spec; vector ids: 0 x, 1 Lx, 2 w, 3 Lw, 4 p, 5 Lp
extern "C" int spike_fd_dots(spike_fd_ctx *c, int nd, const int *ia, const int *ib, double *sums)
{
    if (!c || nd < 1 || nd > MAXD) return SPIKE_ERR_ARG;
    DotArgs d;
    d.nd = nd;
    for (int j = 0; j < nd; ++j) { d.a[j] = c->v[ia[j]]; d.b[j] = c->v[ib[j]]; }
    hipLaunchKernelGGL(k_fd_dots, dim3((unsigned)c->nchunks), dim3(256), 0, c->st, d, c->n, c->nchunks, c->dchunk);
    FDCHK(hipMemcpyAsync(c->hchunk, c->dchunk, sizeof(double) * (size_t)(nd * c->nchunks), hipMemcpyDeviceToHost, c->st));
    FDCHK(hipStreamSynchronize(c->st));
    for (int j = 0; j < nd; ++j) {
        double tot = 0.0;
        for (int64_t q = 0; q < c->nchunks; ++q) tot += c->hchunk[(int64_t)j * c->nchunks + q];
        sums[j] = tot;
    }
    return SPIKE_OK;
}

static int chunk_total(spike_fd_ctx *c, double *tot)
{
    FDCHK(hipMemcpyAsync(c->hchunk, c->dchunk, sizeof(double) * (size_t)c->nchunks, hipMemcpyDeviceToHost, c->st));
    FDCHK(hipStreamSynchronize(c->st));
    double t = 0.0;
    for (int64_t q = 0; q < c->nchunks; ++q) t += c->hchunk[q];
    *tot = t;
    return SPIKE_OK;
}

extern "C" int spike_fd_lap(spike_fd_ctx *c, int src, int dst)   // v[dst] = L v[src]
{
    hipLaunchKernelGGL(k_fd_lap, grid1(c->n), dim3(256), 0, c->st, c->n, c->xadj, c->adj, c->w_e, c->deg, c->v[src], c->v[dst]);
    return hipGetLastError() == hipSuccess ? SPIKE_OK : SPIKE_ERR_HIP;
}
extern "C" int spike_fd_resid(spike_fd_ctx *c, double rho, double *rn2)   // w = Lx - rho x; rn2 = sum w^2
{
    hipLaunchKernelGGL(k_fd_resid, dim3((unsigned)c->nchunks), dim3(256), 0, c->st, c->n, rho, c->v[0], c->v[1], c->v[2], c->dchunk);
    return chunk_total(c, rn2);
}
extern "C" int spike_fd_precond(spike_fd_ctx *c, double *sum)   // w /= deg; sum = sum w
{
    hipLaunchKernelGGL(k_fd_precond, dim3((unsigned)c->nchunks), dim3(256), 0, c->st, c->n, c->deg, c->v[2], c->dchunk);
    return chunk_total(c, sum);
}
extern "C" int spike_fd_shift(spike_fd_ctx *c, int vec, double m)   // v -= m
{
    hipLaunchKernelGGL(k_fd_shift, grid1(c->n), dim3(256), 0, c->st, c->n, m, c->v[vec]);
    return hipGetLastError() == hipSuccess ? SPIKE_OK : SPIKE_ERR_HIP;
}
extern "C" int spike_fd_axpy(spike_fd_ctx *c, double a, int x, int y, int x2, int y2)   // y -= a x [, y2 -= a x2]; x2 < 0: none
{
    hipLaunchKernelGGL(k_fd_axpy, grid1(c->n), dim3(256), 0, c->st, c->n, a, c->v[x], c->v[y], x2 >= 0 ? c->v[x2] : nullptr,
                       y2 >= 0 ? c->v[y2] : nullptr);
    return hipGetLastError() == hipSuccess ? SPIKE_OK : SPIKE_ERR_HIP;
}
extern "C" int spike_fd_div(spike_fd_ctx *c, double s, int y, int y2)   // y /= s [, y2 /= s]
{
    hipLaunchKernelGGL(k_fd_div, grid1(c->n), dim3(256), 0, c->st, c->n, s, c->v[y], y2 >= 0 ? c->v[y2] : nullptr);
    return hipGetLastError() == hipSuccess ? SPIKE_OK : SPIKE_ERR_HIP;
}
extern "C" int spike_fd_update(spike_fd_ctx *c, double c0, double c1, double c2, int havep)
{
    hipLaunchKernelGGL(k_fd_update, grid1(c->n), dim3(256), 0, c->st, c->n, c0, c1, c2, havep, c->v[0], c->v[1], c->v[2], c->v[3],
                       c->v[4], c->v[5]);
    return hipGetLastError() == hipSuccess ? SPIKE_OK : SPIKE_ERR_HIP;
}
extern "C" int spike_fd_fill_alternating(spike_fd_ctx *c)   // x_i = (i odd ? 1 : -1)
{
    hipLaunchKernelGGL(k_fd_fill_alt, grid1(c->n), dim3(256), 0, c->st, c->n, c->v[0]);
    return hipGetLastError() == hipSuccess ? SPIKE_OK : SPIKE_ERR_HIP;
}
extern "C" int spike_fd_download_x(spike_fd_ctx *c, double *x)
{
    FDCHK(hipMemcpyAsync(x, c->v[0], sizeof(double) * (size_t)c->n, hipMemcpyDeviceToHost, c->st));
    FDCHK(hipStreamSynchronize(c->st));
    return SPIKE_OK;
}
