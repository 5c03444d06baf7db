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
#include <cmath>

#define FD_HD __host__ __device__ static inline __attribute__((always_inline))
#include "host/fiedler_steer.h"

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

// ---- the six vector steps of one iteration (fiedler.c:refine_core) --------------------------------------------------------
// Each kernel performs the element-wise statements of one step and that step's sums over the UPDATED values, chunk by chunk
// in the reduction order of the spec (host: h_* in fiedler.c, the same statements over all elements, then the same sums).
// Scalars come from the iteration state in device memory; k_fd_steer<STEP> (one workgroup) adds the chunk sums in chunk order
// and runs the step's scalar epilogue (fiedler_steer.h, compiled for both sides) on one thread.  After `done` every kernel
// returns at once, so a level's maxit iterations are launched blind, without a host round trip.
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
__global__ __launch_bounds__(256) void k_fd_resid_precond(int64_t n, const fd_state *st, double *x, double *Lx, const double *deg,
                                                          double *w, FusedOut out)
{
#pragma clang fp contract(off)
    __shared__ double s0[256], s1[256];
    if (st->done) return;
    const int scale = st->scale;
    const double xn = st->xn, rho = st->rho;
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
__global__ __launch_bounds__(256) void k_fd_shift_dots(int64_t n, const fd_state *st, double *w, const double *x, const double *p,
                                                       FusedOut out)
{
#pragma clang fp contract(off)
    __shared__ double s0[256], s1[256];
    if (st->done) return;
    const double m = st->m;
    const int havep = st->havep;
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
__global__ __launch_bounds__(256) void k_fd_orth_p(int64_t n, const fd_state *st, double *w, const double *x, const double *Lx,
                                                   double *p, double *Lp, FusedOut out)
{
#pragma clang fp contract(off)
    __shared__ double s0[256], s1[256];
    if (st->done) return;
    const double a = st->a, b = st->b;
    const int havep = st->havep;
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
__global__ __launch_bounds__(256) void k_fd_orth_w(int64_t n, const fd_state *st, double *w, double *p, double *Lp, FusedOut out)
{
#pragma clang fp contract(off)
    __shared__ double s0[256], s1[256];
    if (st->done) return;
    const double pn = st->pn, a2 = st->a2;
    const int havep = st->havep;
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

// w /= wn (one thread per element)
__global__ void k_fd_div_w(int64_t n, const fd_state *st, double *w)
{
#pragma clang fp contract(off)
    if (st->done) return;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) w[i] /= st->wn;
}

// Lw = L w (one thread per row, the row's terms in storage order)
__global__ void k_fd_lap_st(int64_t n, const fd_state *st, const int64_t *xadj, const int32_t *adj, const double *we, const double *deg,
                            const double *x, double *y)
{
#pragma clang fp contract(off)
    if (st->done) return;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    // the row's terms are subtracted one by one in storage order (spec); their LOADS are batched eight at a time from
    // clamped addresses -- a plain loop waits two dependent memory round trips per entry, and the coarse levels have rows of
    // 30-60 entries (15.5 -> 5 us per call averaged over the levels of the n = 3.2e5 case)
    double s = deg[i] * x[i];
    const int64_t kb = xadj[i], ke = xadj[i + 1];
    for (int64_t k = kb; k < ke; k += 8) {
        int32_t a[8];
        double ww[8], xx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int64_t kk = (k + u < ke) ? k + u : ke - 1; a[u] = adj[kk]; ww[u] = we[kk]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) xx[u] = x[a[u]];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (k + u < ke) s -= ww[u] * xx[u];
    }
    y[i] = s;
}

// the Rayleigh-Ritz products x.Lx, x.Lw, x.Lp, w.Lw, w.Lp, p.Lp -> out[j * nchunks + chunk] (p entries only when p exists)
__global__ __launch_bounds__(256) void k_fd_rr_dots(int64_t n, const fd_state *st, const double *x, const double *Lx, const double *w,
                                                    const double *Lw, const double *p, const double *Lp, int64_t nchunks, double *out)
{
#pragma clang fp contract(off)
    __shared__ double s[6][256];
    if (st->done) return;
    const int havep = st->havep;
    const int t = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * CH;
    double v[6];
    for (int q = 0; q < 4; ++q) {
        const int64_t i = c0 + t + 256 * q;
        double pr[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (i < n) {
            const double xi = x[i], wi = w[i], lwi = Lw[i];
            pr[0] = xi * Lx[i]; pr[1] = xi * lwi; pr[3] = wi * lwi;
            if (havep) { const double pi = p[i], lpi = Lp[i]; pr[2] = xi * lpi; pr[4] = wi * lpi; pr[5] = pi * lpi; }
        }
        for (int j = 0; j < 6; ++j) v[j] = (q == 0) ? pr[j] : v[j] + pr[j];
    }
    for (int j = 0; j < 6; ++j) s[j][t] = v[j];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) for (int j = 0; j < 6; ++j) s[j][t] += s[j][t + o];
        __syncthreads();
    }
    if (t < 6) out[(int64_t)t * nchunks + blockIdx.x] = s[t][0];
}

// Rayleigh-Ritz update; sums: x.x of the new x
__global__ __launch_bounds__(256) void k_fd_update_xx(int64_t n, const fd_state *st, double *x, double *Lx, const double *w,
                                                      const double *Lw, double *p, double *Lp, FusedOut out)
{
#pragma clang fp contract(off)
    __shared__ double s0[256], s1[256];
    if (st->done) return;
    const double c0_ = st->c0, c1 = st->c1, c2 = st->c2;
    const int havep = st->havep;
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

// after the last iteration: x /= xn, Lx /= xn if the iterate is still unscaled
__global__ void k_fd_final_scale(int64_t n, const fd_state *st, double *x, double *Lx)
{
#pragma clang fp contract(off)
    if (!st->scale) return;
    const double xn = st->xn;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) { x[i] /= xn; Lx[i] /= xn; }
}

// The scalar step after vector step STEP: totals = chunk sums added IN CHUNK ORDER (one lane per sum, the chunk sums staged
// through LDS so that the serial additions do not wait for memory), then the epilogue of fiedler_steer.h on one thread.
constexpr int STEER_TILE = 1024;
template <int STEP, int NSUM>
__global__ __launch_bounds__(256) void k_fd_steer(fd_state *st, const double *chunk, int64_t nchunks)
{
#pragma clang fp contract(off)
    __shared__ double buf[NSUM][STEER_TILE];
    __shared__ double tot[NSUM];
    if (st->done) return;
    const int t = threadIdx.x;
    double acc = 0.0;
    for (int64_t q0 = 0; q0 < nchunks; q0 += STEER_TILE) {
        const int len = (int)((nchunks - q0 < STEER_TILE) ? nchunks - q0 : STEER_TILE);
        for (int j = 0; j < NSUM; ++j)
            for (int q = t; q < len; q += 256) buf[j][q] = chunk[(int64_t)j * nchunks + q0 + q];
        __syncthreads();
        if (t < NSUM)
            for (int q = 0; q < len; ++q) acc += buf[t][q];
        __syncthreads();
    }
    if (t < NSUM) tot[t] = acc;
    __syncthreads();
    if (t != 0) return;
    double sums[NSUM];
    for (int j = 0; j < NSUM; ++j) sums[j] = tot[j];
    if (STEP == 0) fd_after_resid(st, sums);
    if (STEP == 1) fd_after_shift(st, sums);
    if (STEP == 2) fd_after_orth_p(st, sums);
    if (STEP == 3) fd_after_orth_w(st, sums);
    if (STEP == 4) fd_after_rr_dots(st, sums);
    if (STEP == 5) fd_after_update(st, sums);
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
    fd_state *dstate = nullptr, *hstate = nullptr;   // iteration state (device) and its pinned read-back
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
    (void)hipFree(c->xadj); (void)hipFree(c->adj); (void)hipFree(c->w_e); (void)hipFree(c->deg); (void)hipFree(c->dchunk); (void)hipFree(c->dstate);
    if (c->hstate) (void)hipHostFree(c->hstate);
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
              hipHostMalloc((void **)&c->hchunk, sizeof(double) * (size_t)(MAXD * c->nchunks), hipHostMallocDefault) == hipSuccess &&
              hipMalloc((void **)&c->dstate, sizeof(fd_state)) == hipSuccess &&
              hipHostMalloc((void **)&c->hstate, 2 * sizeof(fd_state), hipHostMallocDefault) == hipSuccess;
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

// ---- a level's whole refinement loop without a host round trip per iteration: (six vector steps, each with its scalar
// step) per iteration, the state (fiedler_steer.h) in device memory, then the final scaling.  The iterations are enqueued in
// CHUNKS of 32; after each chunk the state is copied to one of two pinned slots, and the host looks at chunk c's copy only
// after chunk c+1 is in the queue -- the GPU never waits for the host, and a level whose stopping test fires early costs at
// most two chunks of no-op launches instead of the rest of maxit (up to 1000 x 14).  The no-op steps change nothing, so the
// result is the same bits whatever the chunking.
extern "C" int spike_fd_refine(spike_fd_ctx *c, double dmax, double rho, int maxit, int *its)
{
    if (!c || maxit < 0) return SPIKE_ERR_ARG;
    fd_state h;
    fd_init(&h, (double)c->n, dmax, rho);
    FDCHK(hipMemcpyAsync(c->dstate, &h, sizeof h, hipMemcpyHostToDevice, c->st));
    const dim3 gc((unsigned)c->nchunks), ge = grid1(c->n), b(256);
    const FusedOut out{c->dchunk, c->dchunk + c->nchunks};
    double **v = c->v;
    fd_state *st = c->dstate;
    const int64_t n = c->n, nch = c->nchunks;
    constexpr int CHUNK = 32;
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; ++i) FDCHK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    auto iteration = [&]() {
        hipLaunchKernelGGL(k_fd_resid_precond, gc, b, 0, c->st, n, st, v[0], v[1], c->deg, v[2], out);
        hipLaunchKernelGGL((k_fd_steer<0, 2>), dim3(1), b, 0, c->st, st, c->dchunk, nch);
        hipLaunchKernelGGL(k_fd_shift_dots, gc, b, 0, c->st, n, st, v[2], v[0], v[4], out);
        hipLaunchKernelGGL((k_fd_steer<1, 2>), dim3(1), b, 0, c->st, st, c->dchunk, nch);
        hipLaunchKernelGGL(k_fd_orth_p, gc, b, 0, c->st, n, st, v[2], v[0], v[1], v[4], v[5], out);
        hipLaunchKernelGGL((k_fd_steer<2, 2>), dim3(1), b, 0, c->st, st, c->dchunk, nch);
        hipLaunchKernelGGL(k_fd_orth_w, gc, b, 0, c->st, n, st, v[2], v[4], v[5], out);
        hipLaunchKernelGGL((k_fd_steer<3, 1>), dim3(1), b, 0, c->st, st, c->dchunk, nch);
        hipLaunchKernelGGL(k_fd_div_w, ge, b, 0, c->st, n, st, v[2]);
        hipLaunchKernelGGL(k_fd_lap_st, ge, b, 0, c->st, n, st, c->xadj, c->adj, c->w_e, c->deg, v[2], v[3]);
        hipLaunchKernelGGL(k_fd_rr_dots, gc, b, 0, c->st, n, st, v[0], v[1], v[2], v[3], v[4], v[5], nch, c->dchunk);
        hipLaunchKernelGGL((k_fd_steer<4, 6>), dim3(1), b, 0, c->st, st, c->dchunk, nch);
        hipLaunchKernelGGL(k_fd_update_xx, gc, b, 0, c->st, n, st, v[0], v[1], v[2], v[3], v[4], v[5], out);
        hipLaunchKernelGGL((k_fd_steer<5, 1>), dim3(1), b, 0, c->st, st, c->dchunk, nch);
    };
    int rc = SPIKE_OK;
    int q = 0;   // chunks enqueued
    for (int it0 = 0; it0 < maxit && rc == SPIKE_OK; it0 += CHUNK, ++q) {
        const int it1 = it0 + CHUNK < maxit ? it0 + CHUNK : maxit;
        for (int it = it0; it < it1; ++it) iteration();
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&c->hstate[q & 1], st, sizeof h, hipMemcpyDeviceToHost, c->st) != hipSuccess ||
            hipEventRecord(ev[q & 1], c->st) != hipSuccess) { rc = SPIKE_ERR_HIP; break; }
        if (q >= 1) {   // chunk q is in the queue: now the host may wait for chunk q-1's state
            if (hipEventSynchronize(ev[(q - 1) & 1]) != hipSuccess) { rc = SPIKE_ERR_HIP; break; }
            if (c->hstate[(q - 1) & 1].done) { ++q; break; }   // the stopping test fired: chunk q is all no-ops, nothing more to enqueue
        }
    }
    for (int i = 0; i < 2; ++i) (void)hipEventDestroy(ev[i]);
    if (rc != SPIKE_OK) { (void)hipStreamSynchronize(c->st); return rc; }
    hipLaunchKernelGGL(k_fd_final_scale, ge, b, 0, c->st, n, st, v[0], v[1]);
    FDCHK(hipGetLastError());
    FDCHK(hipMemcpyAsync(c->hstate, st, sizeof h, hipMemcpyDeviceToHost, c->st));
    FDCHK(hipStreamSynchronize(c->st));
    if (its) *its = c->hstate->its;
    return SPIKE_OK;
}
