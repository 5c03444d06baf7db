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

__global__ void k_fd_shift(int64_t n, double m, double *w)
{
#pragma clang fp contract(off)
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) w[i] -= m;
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

// ---- fused steps of refine_core (round 2: nine host round trips per iteration -> six) ------------------------------------
// Each kernel performs the element-wise statements of one step and accumulates that step's sums over the UPDATED values,
// chunk by chunk in the reduction order of the spec; the host implementation runs the same statements over all elements
// and then takes the same sums.
struct FusedOut { double *o0, *o1; };

__device__ __forceinline__ void chunk_reduce2(double v0, double v1, double *s0, double *s1, int t, FusedOut out, bool two)
{
    s0[t] = v0;
    if (two) s1[t] = v1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) { s0[t] += s0[t + o]; if (two) s1[t] += s1[t + o]; }
        __syncthreads();
    }
    if (t == 0) { out.o0[blockIdx.x] = s0[0]; if (two) out.o1[blockIdx.x] = s1[0]; }
}

// [x /= xn, Lx /= xn]; w = Lx - rho x; sums: w^2 ; then w /= deg; sums: w
__global__ __launch_bounds__(256) void k_fd_resid_precond(int64_t n, int scale, double xn, double rho, double *x, double *Lx,
                                                          const double *deg, double *w, FusedOut out)
{
#pragma clang fp contract(off)
    __shared__ double s0[256], s1[256];
    const int t = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * CH;
    double v0 = 0.0, v1 = 0.0;
    for (int q = 0; q < 4; ++q) {
        const int64_t i = c0 + t + 256 * q;
        double p0 = 0.0, p1 = 0.0;
        if (i < n) {
            double xi = x[i], li = Lx[i];
            if (scale) { xi /= xn; li /= xn; x[i] = xi; Lx[i] = li; }
            const double r = li - rho * xi;
            p0 = r * r;
            const double dg = deg[i];
            p1 = r / (dg > 0 ? dg : 1.0);
            w[i] = p1;
        }
        v0 = (q == 0) ? p0 : v0 + p0;
        v1 = (q == 0) ? p1 : v1 + p1;
    }
    chunk_reduce2(v0, v1, s0, s1, t, out, true);
}

// w -= m; sums: w.x [, p.x]
__global__ __launch_bounds__(256) void k_fd_shift_dots(int64_t n, double m, int havep, double *w, const double *x, const double *p,
                                                       FusedOut out)
{
#pragma clang fp contract(off)
    __shared__ double s0[256], s1[256];
    const int t = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * CH;
    double v0 = 0.0, v1 = 0.0;
    for (int q = 0; q < 4; ++q) {
        const int64_t i = c0 + t + 256 * q;
        double p0 = 0.0, p1 = 0.0;
        if (i < n) {
            const double wi = w[i] - m;
            w[i] = wi;
            p0 = wi * x[i];
            if (havep) p1 = p[i] * x[i];
        }
        v0 = (q == 0) ? p0 : v0 + p0;
        v1 = (q == 0) ? p1 : v1 + p1;
    }
    chunk_reduce2(v0, v1, s0, s1, t, out, havep != 0);
}

// w -= a x; [p -= b x; Lp -= b Lx; sums: p.p, w.p]
__global__ __launch_bounds__(256) void k_fd_orth_p(int64_t n, double a, double b, int havep, double *w, const double *x,
                                                   const double *Lx, double *p, double *Lp, FusedOut out)
{
#pragma clang fp contract(off)
    __shared__ double s0[256], s1[256];
    const int t = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * CH;
    double v0 = 0.0, v1 = 0.0;
    for (int q = 0; q < 4; ++q) {
        const int64_t i = c0 + t + 256 * q;
        double p0 = 0.0, p1 = 0.0;
        if (i < n) {
            const double wi = w[i] - a * x[i];
            w[i] = wi;
            if (havep) {
                const double pi = p[i] - b * x[i];
                p[i] = pi;
                Lp[i] = Lp[i] - b * Lx[i];
                p0 = pi * pi;
                p1 = wi * pi;
            }
        }
        v0 = (q == 0) ? p0 : v0 + p0;
        v1 = (q == 0) ? p1 : v1 + p1;
    }
    if (havep) chunk_reduce2(v0, v1, s0, s1, t, out, true);
}

// [p /= pn; Lp /= pn; w -= a2 p]; sums: w.w
__global__ __launch_bounds__(256) void k_fd_orth_w(int64_t n, double pn, double a2, int havep, double *w, double *p, double *Lp,
                                                   FusedOut out)
{
#pragma clang fp contract(off)
    __shared__ double s0[256], s1[256];
    const int t = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * CH;
    double v0 = 0.0;
    for (int q = 0; q < 4; ++q) {
        const int64_t i = c0 + t + 256 * q;
        double p0 = 0.0;
        if (i < n) {
            double wi = w[i];
            if (havep) {
                const double pi = p[i] / pn;
                p[i] = pi;
                Lp[i] = Lp[i] / pn;
                wi = wi - a2 * pi;
                w[i] = wi;
            }
            p0 = wi * wi;
        }
        v0 = (q == 0) ? p0 : v0 + p0;
    }
    chunk_reduce2(v0, 0.0, s0, s1, t, out, false);
}

// Rayleigh-Ritz update; sums: x.x of the new x
__global__ __launch_bounds__(256) void k_fd_update_xx(int64_t n, double c0_, double c1, double c2, int havep, double *x, double *Lx,
                                                      const double *w, const double *Lw, double *p, double *Lp, FusedOut out)
{
#pragma clang fp contract(off)
    __shared__ double s0[256], s1[256];
    const int t = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * CH;
    double v0 = 0.0;
    for (int q = 0; q < 4; ++q) {
        const int64_t i = c0 + t + 256 * q;
        double p0 = 0.0;
        if (i < n) {
            const double pn = c1 * w[i] + (havep ? c2 * p[i] : 0.0);
            const double Lpn = c1 * Lw[i] + (havep ? c2 * Lp[i] : 0.0);
            const double xi = c0_ * x[i] + pn;
            x[i] = xi;
            Lx[i] = c0_ * Lx[i] + Lpn;
            p[i] = pn;
            Lp[i] = Lpn;
            p0 = xi * xi;
        }
        v0 = (q == 0) ? p0 : v0 + p0;
    }
    chunk_reduce2(v0, 0.0, s0, s1, t, out, false);
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


extern "C" int spike_fd_lap(spike_fd_ctx *c, int src, int dst)   // v[dst] = L v[src]
{
    hipLaunchKernelGGL(k_fd_lap, grid1(c->n), dim3(256), 0, c->st, c->n, c->xadj, c->adj, c->w_e, c->deg, c->v[src], c->v[dst]);
    return hipGetLastError() == hipSuccess ? SPIKE_OK : SPIKE_ERR_HIP;
}
extern "C" int spike_fd_shift(spike_fd_ctx *c, int vec, double m)   // v -= m
{
    hipLaunchKernelGGL(k_fd_shift, grid1(c->n), dim3(256), 0, c->st, c->n, m, c->v[vec]);
    return hipGetLastError() == hipSuccess ? SPIKE_OK : SPIKE_ERR_HIP;
}
extern "C" int spike_fd_div(spike_fd_ctx *c, double s, int y, int y2)   // y /= s [, y2 /= s]
{
    hipLaunchKernelGGL(k_fd_div, grid1(c->n), dim3(256), 0, c->st, c->n, s, c->v[y], y2 >= 0 ? c->v[y2] : nullptr);
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

// ---- fused steps (see the kernels): sums[0..1] in the reduction order of the spec ------------------------------------------
static int two_totals(spike_fd_ctx *c, int nsum, double *sums)
{
    FDCHK(hipMemcpyAsync(c->hchunk, c->dchunk, sizeof(double) * (size_t)(nsum * c->nchunks), hipMemcpyDeviceToHost, c->st));
    FDCHK(hipStreamSynchronize(c->st));
    for (int j = 0; j < nsum; ++j) {
        double t = 0.0;
        for (int64_t q = 0; q < c->nchunks; ++q) t += c->hchunk[(int64_t)j * c->nchunks + q];
        sums[j] = t;
    }
    return SPIKE_OK;
}
#define FD_OUT(c) FusedOut{(c)->dchunk, (c)->dchunk + (c)->nchunks}

extern "C" int spike_fd_resid_precond(spike_fd_ctx *c, int scale, double xn, double rho, double *sums /* rn2, sum w */)
{
    hipLaunchKernelGGL(k_fd_resid_precond, dim3((unsigned)c->nchunks), dim3(256), 0, c->st, c->n, scale, xn, rho, c->v[0], c->v[1],
                       c->deg, c->v[2], FD_OUT(c));
    return two_totals(c, 2, sums);
}
extern "C" int spike_fd_shift_dots(spike_fd_ctx *c, double m, int havep, double *sums /* w.x, p.x */)
{
    hipLaunchKernelGGL(k_fd_shift_dots, dim3((unsigned)c->nchunks), dim3(256), 0, c->st, c->n, m, havep, c->v[2], c->v[0], c->v[4],
                       FD_OUT(c));
    sums[1] = 0.0;
    return two_totals(c, havep ? 2 : 1, sums);
}
extern "C" int spike_fd_orth_p(spike_fd_ctx *c, double a, double b, int havep, double *sums /* p.p, w.p */)
{
    hipLaunchKernelGGL(k_fd_orth_p, dim3((unsigned)c->nchunks), dim3(256), 0, c->st, c->n, a, b, havep, c->v[2], c->v[0], c->v[1],
                       c->v[4], c->v[5], FD_OUT(c));
    sums[0] = sums[1] = 0.0;
    if (!havep) return hipGetLastError() == hipSuccess ? SPIKE_OK : SPIKE_ERR_HIP;
    return two_totals(c, 2, sums);
}
extern "C" int spike_fd_orth_w(spike_fd_ctx *c, double pn, double a2, int havep, double *ww)
{
    hipLaunchKernelGGL(k_fd_orth_w, dim3((unsigned)c->nchunks), dim3(256), 0, c->st, c->n, pn, a2, havep, c->v[2], c->v[4], c->v[5],
                       FD_OUT(c));
    return two_totals(c, 1, ww);
}
extern "C" int spike_fd_update_xx(spike_fd_ctx *c, double c0, double c1, double c2, int havep, double *xx)
{
    hipLaunchKernelGGL(k_fd_update_xx, dim3((unsigned)c->nchunks), dim3(256), 0, c->st, c->n, c0, c1, c2, havep, c->v[0], c->v[1],
                       c->v[2], c->v[3], c->v[4], c->v[5], FD_OUT(c));
    return two_totals(c, 1, xx);
}
