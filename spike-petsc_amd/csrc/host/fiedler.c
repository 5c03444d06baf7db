/*
 * fiedler.c -- spectral (Fiedler-vector) symmetric ordering.
 *
 * Reference slot: MatGetOrdering_Fiedler, /root/reference/src/petsc_mat_fiedler.c:11-58, which calls HSL_MC73
 * (mc73_order job 3, coarsest_size 2, edge weights iff a(1) > 0: /root/reference/src/hslmc73f.F90:15-31) and inverts
 * the returned permutation (petsc_mat_fiedler.c:49).  HSL_MC73 is proprietary, not vendored, and absent here:
 * PARITY UNPINNED.  This file therefore states its OWN deterministic specification:
 *
 *  1. graph  : vertices = rows; edge {i,j}, i != j, when a_ij or a_ji is stored with |value| >= 1e-12 (the drop
 *              tolerance of the reference's own Laplacian builder, src/spectralPartition.c:63-139);
 *              weight = |a_ij| + |a_ji| in weighted mode, 1 otherwise; weighted mode iff a[0] > 0 (hslmc73f.F90:19).
 *  2. components in order of their smallest vertex; each ordered on its own, then concatenated.
 *  3. Fiedler vector by a multilevel scheme (as MC73 is multilevel): heavy-edge matching in index order (ties: smaller
 *     index) down to <= 64 vertices, cyclic-Jacobi dense eigen-solve there, then per level piecewise-constant
 *     prolongation + single-vector LOBPCG (Jacobi preconditioner, constant vector deflated), at most 300 iterations,
 *     stop at ||L x - rho x||_2 <= 1e-9 * max_i deg_i.  (Measured on the 321 821-vertex circuit-like matrix: cutting the
 *     iteration caps to 30-100 makes it 2-4x faster but lets a few vertices stray -- bandwidth 187-1194 instead of 49 at
 *     a 1-12 % larger profile -- so the caps stay at 300.)
 *  4. sign   : the entry of largest magnitude (lowest index among ties) is made positive.
 *  5. order  : stable sort by DESCENDING vector value (the reference's prototype reverses an ascending sort,
 *              src/spectralPartition.c:336-338), ties by index.  Output order[k] = old index of the vertex placed at
 *              position k (the "new -> old" convention of a PETSc IS, i.e. what petsc_mat_fiedler.c:49 builds).
 *
 *  6. arithmetic: IEEE fp64, one operation at a time (no fused multiply-add: this file is compiled with
 *     -ffp-contract=off), every sum of the LOBPCG refinement (dot products, the mean deflate() removes, the residual
 *     norm) in ONE fixed REDUCTION ORDER: chunks of 1024 consecutive indices; inside a chunk 256 slots,
 *     slot t = ((v[t] + v[t+256]) + v[t+512]) + v[t+768] (absent elements = +0.0), the binary tree s[t] += s[t+o] for
 *     o = 128 ... 1, and the chunk sums added in chunk order.  That order is what a 256-thread workgroup computes
 *     naturally, so the SAME refinement runs on the device (csrc/spike_fiedler.hip, levels of >= 12288 vertices when a
 *     device is present and use_device is set) with bit-identical results: refine_core() below is the only copy of the
 *     iteration, driven through a small table of vector operations that has a host and a device implementation.
 *
 * => the permutation is reproducible bit for bit, on the host alone or with the device doing the vector work
 *    (tests/test_host_gpu.py::test_fiedler_device_equals_host_bit_for_bit).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int64_t I;

typedef struct {
    I n;
    I *xadj, *adj;
    double *w;
    double *deg;
} graph_t;

static void graph_free(graph_t *g) { free(g->xadj); free(g->adj); free(g->w); free(g->deg); memset(g, 0, sizeof *g); }

static void lap_mult(const graph_t *g, const double *x, double *y)
{
    for (I i = 0; i < g->n; ++i) {
        double s = g->deg[i] * x[i];
        for (I k = g->xadj[i]; k < g->xadj[i + 1]; ++k) s -= g->w[k] * x[g->adj[k]];
        y[i] = s;
    }
}

/* THE reduction order of the spec (header, item 6): sum_i a[i]*b[i]; b == NULL means b = 1 */
static double dot(I n, const double *a, const double *b)
{
    double total = 0.0;
    for (I c0 = 0; c0 < n; c0 += 1024) {
        double sl[256];
        for (int t = 0; t < 256; ++t) {
            double v = 0.0;
            for (int q = 0; q < 4; ++q) {
                const I i = c0 + t + 256 * q;
                const double pr = (i < n) ? (b ? a[i] * b[i] : a[i] * 1.0) : 0.0;
                v = (q == 0) ? pr : v + pr;
            }
            sl[t] = v;
        }
        for (int o = 128; o > 0; o >>= 1)
            for (int t = 0; t < o; ++t) sl[t] += sl[t + o];
        total += sl[0];
    }
    return total;
}

/* cyclic Jacobi for a dense symmetric matrix (n <= ~64): eigenvalues in ev, eigenvectors in columns of V */
static void jacobi_eig(int n, double *A, double *V, double *ev)
{
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = (i == j);
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0;
        for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) off += A[p * n + q] * A[p * n + q];
        if (off < 1e-30) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; ++i) ev[i] = A[i * n + i];
}

/* heavy-edge matching coarsening; returns coarse graph and fine->coarse map (malloc'd) */
static I *coarsen(const graph_t *g, graph_t *c)
{
    const I n = g->n;
    I *map = (I *)malloc(sizeof(I) * (size_t)n);
    for (I i = 0; i < n; ++i) map[i] = -1;
    I nc = 0;
    for (I i = 0; i < n; ++i) {
        if (map[i] >= 0) continue;
        I best = -1;
        double bw = -1.0;
        for (I k = g->xadj[i]; k < g->xadj[i + 1]; ++k) {
            const I j = g->adj[k];
            if (map[j] >= 0 || j == i) continue;
            if (g->w[k] > bw || (g->w[k] == bw && j < best)) { bw = g->w[k]; best = j; }
        }
        map[i] = nc;
        if (best >= 0) map[best] = nc;
        ++nc;
    }
    /* build coarse adjacency with a marker array (deterministic: fine vertices in index order) */
    c->n = nc;
    c->xadj = (I *)calloc((size_t)nc + 1, sizeof(I));
    c->deg = (double *)calloc((size_t)nc, sizeof(double));
    I *first = (I *)malloc(sizeof(I) * (size_t)nc), *next = (I *)malloc(sizeof(I) * (size_t)n);
    for (I a = 0; a < nc; ++a) first[a] = -1;
    for (I i = n - 1; i >= 0; --i) { next[i] = first[map[i]]; first[map[i]] = i; }
    I *mark = (I *)malloc(sizeof(I) * (size_t)nc);
    for (I a = 0; a < nc; ++a) mark[a] = -1;
    size_t cap = (size_t)(g->xadj[n] > 0 ? g->xadj[n] : 1);
    c->adj = (I *)malloc(sizeof(I) * cap);
    c->w = (double *)malloc(sizeof(double) * cap);
    I pos = 0;
    for (I a = 0; a < nc; ++a) {
        const I start = pos;
        for (I i = first[a]; i >= 0; i = next[i])
            for (I k = g->xadj[i]; k < g->xadj[i + 1]; ++k) {
                const I b = map[g->adj[k]];
                if (b == a) continue;
                if (mark[b] >= start) c->w[mark[b]] += g->w[k];
                else { mark[b] = pos; c->adj[pos] = b; c->w[pos] = g->w[k]; ++pos; }
            }
        c->xadj[a + 1] = pos;
        double d = 0;
        for (I k = start; k < pos; ++k) d += c->w[k];
        c->deg[a] = d;
    }
    free(first); free(next); free(mark);
    return map;
}

static void eig3(int m, double G[3][3], double c[3], double *lam)
{
    double A[9], V[9], ev[3];
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) A[i * m + j] = G[i][j];
    jacobi_eig(m, A, V, ev);
    int b = 0;
    for (int i = 1; i < m; ++i) if (ev[i] < ev[b]) b = i;
    for (int i = 0; i < m; ++i) c[i] = V[i * m + b];
    *lam = ev[b];
}

/* ---- the vectors of one refinement and the operations the iteration is written in ------------------------------------------
 * ids: 0 x, 1 Lx, 2 w, 3 Lw, 4 p, 5 Lp, 6 the constant 1.  Two implementations: host arrays (below) and the device
 * (spike_fd_* of libspike_mi355.so); both execute the same IEEE operations in the same order. */
#define FD_DEVICE_MIN 12288   /* smaller levels are launch-latency bound on the device (~0.2 ms per iteration whatever n; the host needs ~17 ns per vertex and iteration); the result is the same either way */
static int g_last_its = 0;   /* SPIKE_FIEDLER_TRACE only */
typedef struct spike_fd_ctx spike_fd_ctx;
int spike_device_count(void);
int spike_fd_create(int64_t n, const int64_t *xadj, const int64_t *adj, const double *w, const double *deg, const double *x0, spike_fd_ctx **out);
int spike_fd_destroy(spike_fd_ctx *c);
int spike_fd_dots(spike_fd_ctx *c, int nd, const int *ia, const int *ib, double *sums);
int spike_fd_lap(spike_fd_ctx *c, int src, int dst);
int spike_fd_shift(spike_fd_ctx *c, int vec, double m);
int spike_fd_div(spike_fd_ctx *c, double s, int y, int y2);
int spike_fd_fill_alternating(spike_fd_ctx *c);
int spike_fd_download_x(spike_fd_ctx *c, double *x);
int spike_fd_resid_precond(spike_fd_ctx *c, int scale, double xn, double rho, double *sums);
int spike_fd_shift_dots(spike_fd_ctx *c, double m, int havep, double *sums);
int spike_fd_orth_p(spike_fd_ctx *c, double a, double b, int havep, double *sums);
int spike_fd_orth_w(spike_fd_ctx *c, double pn, double a2, int havep, double *ww);
int spike_fd_update_xx(spike_fd_ctx *c, double c0, double c1, double c2, int havep, double *xx);

typedef struct {
    const graph_t *g;
    I n;
    double *v[6];        /* host backend: x, Lx, w, Lw, p, Lp */
    spike_fd_ctx *dev;   /* device backend (NULL: host) */
    int err;
} vecs_t;

static void op_dots(vecs_t *V, int nd, const int *ia, const int *ib, double *sums)
{
    if (V->dev) { if (spike_fd_dots(V->dev, nd, ia, ib, sums)) V->err = 1; return; }
    for (int j = 0; j < nd; ++j) sums[j] = dot(V->n, V->v[ia[j]], ib[j] == 6 ? NULL : V->v[ib[j]]);
}
static double op_dot(vecs_t *V, int a, int b) { double s = 0.0; op_dots(V, 1, &a, &b, &s); return s; }
static void op_lap(vecs_t *V, int src, int dst)
{
    if (V->dev) { if (spike_fd_lap(V->dev, src, dst)) V->err = 1; return; }
    lap_mult(V->g, V->v[src], V->v[dst]);
}
static void op_shift(vecs_t *V, int a, double m)
{
    if (V->dev) { if (spike_fd_shift(V->dev, a, m)) V->err = 1; return; }
    for (I i = 0; i < V->n; ++i) V->v[a][i] -= m;
}
static void op_div(vecs_t *V, double s, int y, int y2)
{
    if (V->dev) { if (spike_fd_div(V->dev, s, y, y2)) V->err = 1; return; }
    for (I i = 0; i < V->n; ++i) V->v[y][i] /= s;
    if (y2 >= 0) for (I i = 0; i < V->n; ++i) V->v[y2][i] /= s;
}
static void op_update(vecs_t *V, double c0, double c1, double c2, int havep)
{
    double *x = V->v[0], *Lx = V->v[1], *w = V->v[2], *Lw = V->v[3], *p = V->v[4], *Lp = V->v[5];   /* host vectors only */
    for (I i = 0; i < V->n; ++i) {
        const double pn = c1 * w[i] + (havep ? c2 * p[i] : 0.0);
        const double Lpn = c1 * Lw[i] + (havep ? c2 * Lp[i] : 0.0);
        x[i] = c0 * x[i] + pn;
        Lx[i] = c0 * Lx[i] + Lpn;
        p[i] = pn; Lp[i] = Lpn;
    }
}
static void op_fill_alternating(vecs_t *V)
{
    if (V->dev) { if (spike_fd_fill_alternating(V->dev)) V->err = 1; return; }
    for (I i = 0; i < V->n; ++i) V->v[0][i] = (double)(i % 2 ? 1 : -1);
}
static void op_deflate(vecs_t *V, int a) { const double m = op_dot(V, a, 6) / (double)V->n; op_shift(V, a, m); }
static double op_normalize(vecs_t *V, int a) { const double s = sqrt(op_dot(V, a, a)); if (s > 0) op_div(V, s, a, -1); return s; }

/* ---- fused steps: the element-wise statements of one step, then that step's sums over the UPDATED values (one host round
 * trip each on the device; the host runs the same statements and takes the same sums) --------------------------------------- */
static void op_resid_precond(vecs_t *V, int scale, double xn, double rho, double *sums)   /* sums: |w|^2 before precond, sum of w after */
{
    if (V->dev) { if (spike_fd_resid_precond(V->dev, scale, xn, rho, sums)) V->err = 1; return; }
    double *x = V->v[0], *Lx = V->v[1], *w = V->v[2];
    if (scale) for (I i = 0; i < V->n; ++i) { x[i] /= xn; Lx[i] /= xn; }
    for (I i = 0; i < V->n; ++i) w[i] = Lx[i] - rho * x[i];
    sums[0] = dot(V->n, w, w);
    for (I i = 0; i < V->n; ++i) w[i] /= (V->g->deg[i] > 0 ? V->g->deg[i] : 1.0);
    sums[1] = dot(V->n, w, NULL);
}
static void op_shift_dots(vecs_t *V, double m, int havep, double *sums)   /* w -= m; sums: w.x, p.x */
{
    if (V->dev) { if (spike_fd_shift_dots(V->dev, m, havep, sums)) V->err = 1; return; }
    for (I i = 0; i < V->n; ++i) V->v[2][i] -= m;
    sums[0] = dot(V->n, V->v[2], V->v[0]);
    sums[1] = havep ? dot(V->n, V->v[4], V->v[0]) : 0.0;
}
static void op_orth_p(vecs_t *V, double a, double b, int havep, double *sums)   /* w -= a x; p -= b x; Lp -= b Lx; sums: p.p, w.p */
{
    if (V->dev) { if (spike_fd_orth_p(V->dev, a, b, havep, sums)) V->err = 1; return; }
    double *x = V->v[0], *Lx = V->v[1], *w = V->v[2], *p = V->v[4], *Lp = V->v[5];
    for (I i = 0; i < V->n; ++i) w[i] = w[i] - a * x[i];
    sums[0] = sums[1] = 0.0;
    if (!havep) return;
    for (I i = 0; i < V->n; ++i) { p[i] = p[i] - b * x[i]; Lp[i] = Lp[i] - b * Lx[i]; }
    sums[0] = dot(V->n, p, p);
    sums[1] = dot(V->n, w, p);
}
static double op_orth_w(vecs_t *V, double pn, double a2, int havep)   /* p /= pn; Lp /= pn; w -= a2 p; returns w.w */
{
    double ww = 0.0;
    if (V->dev) { if (spike_fd_orth_w(V->dev, pn, a2, havep, &ww)) V->err = 1; return ww; }
    double *w = V->v[2], *p = V->v[4], *Lp = V->v[5];
    if (havep) for (I i = 0; i < V->n; ++i) { p[i] = p[i] / pn; Lp[i] = Lp[i] / pn; w[i] = w[i] - a2 * p[i]; }
    return dot(V->n, w, w);
}
static double op_update_xx(vecs_t *V, double c0, double c1, double c2, int havep)   /* Rayleigh-Ritz update; returns x.x */
{
    double xx = 0.0;
    if (V->dev) { if (spike_fd_update_xx(V->dev, c0, c1, c2, havep, &xx)) V->err = 1; return xx; }
    op_update(V, c0, c1, c2, havep);
    return dot(V->n, V->v[0], V->v[0]);
}

/* single-vector LOBPCG for the smallest eigenpair of L restricted to the complement of the constant vector: THE iteration,
   the same statements whichever side holds the vectors.  Six reductions per iteration (each a host round trip when the
   vectors live on the device): [residual + preconditioning], [deflation + w.x, p.x], [orthogonalisation against x + p.p, w.p],
   [normalise p, orthogonalise w against p + w.w], [the six Rayleigh-Ritz products], [update + x.x].  The Rayleigh quotient of
   an iterate is its Ritz value (no seventh reduction); p is normalised by the scalar p.p and w's coefficient against it is
   (w.p)/|p|; x and Lx are scaled to unit length at the start of the next iteration. */
static void refine_core(vecs_t *V, double dmax, int maxit)
{
    op_deflate(V, 0);
    if (op_normalize(V, 0) == 0.0) { op_fill_alternating(V); op_deflate(V, 0); op_normalize(V, 0); }
    op_lap(V, 0, 1);
    int havep = 0, scale = 0;
    double rho = op_dot(V, 0, 1), xn = 1.0;
    g_last_its = 0;
    for (int it = 0; it < maxit && !V->err; ++it) {
        g_last_its = it + 1;
        double s2[2];
        op_resid_precond(V, scale, xn, rho, s2);
        scale = 0;
        if (sqrt(s2[0]) <= 1e-9 * dmax) break;
        op_shift_dots(V, s2[1] / (double)V->n, havep, s2);       /* constant vector deflated; w.x, p.x */
        const double a = s2[0], b = s2[1];
        op_orth_p(V, a, b, havep, s2);                            /* against x; p.p, w.p of the results */
        double pn = 1.0, a2 = 0.0;
        if (havep) {
            pn = sqrt(s2[0]);
            if (pn > 1e-300) a2 = s2[1] / pn;
            else havep = 0;
        }
        const double wn = sqrt(op_orth_w(V, pn, a2, havep));
        if (wn < 1e-300) break;
        op_div(V, wn, 2, -1);
        op_lap(V, 2, 3);
        const int m = havep ? 3 : 2;
        double G[3][3], c[3], lam, d6[6];
        if (m == 3) {
            const int ia[6] = {0, 0, 0, 2, 2, 4}, ib[6] = {1, 3, 5, 3, 5, 5};
            op_dots(V, 6, ia, ib, d6);
            G[0][0] = d6[0]; G[0][1] = G[1][0] = d6[1]; G[0][2] = G[2][0] = d6[2];
            G[1][1] = d6[3]; G[1][2] = G[2][1] = d6[4]; G[2][2] = d6[5];
        } else {
            const int ia[3] = {0, 0, 2}, ib[3] = {1, 3, 3};
            op_dots(V, 3, ia, ib, d6);
            G[0][0] = d6[0]; G[0][1] = G[1][0] = d6[1]; G[1][1] = d6[2];
        }
        eig3(m, G, c, &lam);
        if (c[0] < 0) for (int i = 0; i < m; ++i) c[i] = -c[i];
        xn = sqrt(op_update_xx(V, c[0], c[1], havep ? c[2] : 0.0, havep));
        havep = 1;
        scale = 1;          /* x, Lx are divided by xn at the start of the next iteration ... */
        rho = lam;          /* ... whose Rayleigh quotient is this Ritz value */
    }
    if (scale && !V->err) op_div(V, xn, 0, 1);   /* ... or here, after the last one */
}

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static void refine_impl(const graph_t *g, double *x, int maxit, int use_device);
static void refine(const graph_t *g, double *x, int maxit, int use_device)
{
    if (!getenv("SPIKE_FIEDLER_TRACE")) { refine_impl(g, x, maxit, use_device); return; }
    const double t0 = now_s();
    refine_impl(g, x, maxit, use_device);
    fprintf(stderr, "[fiedler] level n=%lld  %s  iterations=%d  %.3f ms\n", (long long)g->n,
            (use_device && g->n >= FD_DEVICE_MIN) ? "device" : "host", g_last_its, 1e3 * (now_s() - t0));
}

static void refine_impl(const graph_t *g, double *x, int maxit, int use_device)
{
    const I n = g->n;
    double dmax = 0;
    for (I i = 0; i < n; ++i) if (g->deg[i] > dmax) dmax = g->deg[i];
    vecs_t V;
    memset(&V, 0, sizeof V);
    V.g = g; V.n = n;
    if (use_device && n >= FD_DEVICE_MIN && spike_fd_create(n, g->xadj, g->adj, g->w, g->deg, x, &V.dev) == 0) {
        refine_core(&V, dmax, maxit);
        if (!V.err && spike_fd_download_x(V.dev, x)) V.err = 1;
        spike_fd_destroy(V.dev);
        if (!V.err) return;
        V.dev = NULL; V.err = 0;   /* device trouble: the host computes the same thing (x was not touched) */
    }
    double *buf = (double *)malloc(sizeof(double) * (size_t)n * 5);
    V.v[0] = x; V.v[1] = buf; V.v[2] = buf + n; V.v[3] = buf + 2 * n; V.v[4] = buf + 3 * n; V.v[5] = buf + 4 * n;
    refine_core(&V, dmax, maxit);
    free(buf);
}

static void fiedler_vector(const graph_t *g, double *x, int level, int use_device)
{
    const I n = g->n;
    if (n <= 64 || level >= 40) {
        if (n <= 64) {
            const int m = (int)n;
            double *A = (double *)calloc((size_t)m * m * 2 + m, sizeof(double)), *V = A + m * m, *ev = V + m * m;
            for (int i = 0; i < m; ++i) {
                A[i * m + i] = g->deg[i];
                for (I k = g->xadj[i]; k < g->xadj[i + 1]; ++k) A[i * m + g->adj[k]] -= g->w[k];
            }
            jacobi_eig(m, A, V, ev);
            int i0 = 0, i1 = -1;
            for (int i = 1; i < m; ++i) if (ev[i] < ev[i0]) i0 = i;
            for (int i = 0; i < m; ++i) if (i != i0 && (i1 < 0 || ev[i] < ev[i1])) i1 = i;
            for (int i = 0; i < m; ++i) x[i] = (i1 >= 0) ? V[i * m + i1] : 0.0;
            free(A);
        } else {
            for (I i = 0; i < n; ++i) x[i] = (double)i - 0.5 * (double)(n - 1);
        }
        refine(g, x, 300, use_device);
        return;
    }
    graph_t c;
    memset(&c, 0, sizeof c);
    I *map = coarsen(g, &c);
    if (c.n > (9 * n) / 10) { /* matching stalls (e.g. star graphs): stop coarsening here */
        for (I i = 0; i < n; ++i) x[i] = (double)i - 0.5 * (double)(n - 1);
        refine(g, x, 1000, use_device);
    } else {
        double *xc = (double *)malloc(sizeof(double) * (size_t)c.n);
        fiedler_vector(&c, xc, level + 1, use_device);
        for (I i = 0; i < n; ++i) x[i] = xc[map[i]];
        free(xc);
        refine(g, x, 300, use_device);
    }
    free(map);
    graph_free(&c);
}

typedef struct { double v; I idx; } key_t2;
static int cmp_desc(const void *a, const void *b)
{
    const key_t2 *x = (const key_t2 *)a, *y = (const key_t2 *)b;
    if (x->v > y->v) return -1;
    if (x->v < y->v) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

/*
 * n, ia, ja, a : 0-based CSR.  order[k] = old index at new position k.  vec (optional, length n) receives the
 * per-component Fiedler vectors.  Returns 0 or -1.
 */
int spike_fiedler_order_ex(int64_t n, const int64_t *ia, const int64_t *ja, const double *a, int64_t *order, double *vec,
                           int use_device);

/* host only */
int spike_fiedler_order(int64_t n, const int64_t *ia, const int64_t *ja, const double *a, int64_t *order, double *vec)
{
    return spike_fiedler_order_ex(n, ia, ja, a, order, vec, 0);
}

/* use_device != 0: the LOBPCG refinement of levels with >= 12288 vertices runs on the GPU (bit-identical result) */
int spike_fiedler_order_ex(int64_t n, const int64_t *ia, const int64_t *ja, const double *a, int64_t *order, double *vec,
                           int use_device)
{
    if (use_device && spike_device_count() <= 0) use_device = 0;
    if (n <= 0 || !ia || !ja || !a || !order) return -1;
    const int weighted = (ia[n] > 0 && a[0] > 0.0);
    const double tol = 1e-12;
    /* symmetrised adjacency: count, fill, then merge duplicates per row (sorted by column) */
    I *cnt = (I *)calloc((size_t)n + 1, sizeof(I));
    for (I i = 0; i < n; ++i)
        for (I k = ia[i]; k < ia[i + 1]; ++k) {
            const I j = ja[k];
            if (j < 0 || j >= n) { free(cnt); return -1; }
            if (j == i || fabs(a[k]) < tol) continue;
            ++cnt[i + 1]; ++cnt[j + 1];
        }
    for (I i = 0; i < n; ++i) cnt[i + 1] += cnt[i];
    const I tot = cnt[n];
    I *adj = (I *)malloc(sizeof(I) * (size_t)(tot > 0 ? tot : 1));
    double *w = (double *)malloc(sizeof(double) * (size_t)(tot > 0 ? tot : 1));
    I *fill = (I *)malloc(sizeof(I) * (size_t)n);
    memcpy(fill, cnt, sizeof(I) * (size_t)n);
    for (I i = 0; i < n; ++i)
        for (I k = ia[i]; k < ia[i + 1]; ++k) {
            const I j = ja[k];
            if (j == i || fabs(a[k]) < tol) continue;
            const double v = weighted ? fabs(a[k]) : 1.0;
            adj[fill[i]] = j; w[fill[i]++] = v;
            adj[fill[j]] = i; w[fill[j]++] = v;
        }
    graph_t g;
    g.n = n;
    g.xadj = (I *)calloc((size_t)n + 1, sizeof(I));
    g.adj = (I *)malloc(sizeof(I) * (size_t)(tot > 0 ? tot : 1));
    g.w = (double *)malloc(sizeof(double) * (size_t)(tot > 0 ? tot : 1));
    g.deg = (double *)calloc((size_t)n, sizeof(double));
    {
        /* bucket the (row, col) pairs by column, then by row: a two-pass counting sort => rows sorted by column */
        I *colcnt = (I *)calloc((size_t)n + 1, sizeof(I));
        for (I k = 0; k < tot; ++k) ++colcnt[adj[k] + 1];
        for (I j = 0; j < n; ++j) colcnt[j + 1] += colcnt[j];
        I *rowof = (I *)malloc(sizeof(I) * (size_t)(tot > 0 ? tot : 1));
        double *wof = (double *)malloc(sizeof(double) * (size_t)(tot > 0 ? tot : 1));
        I *colof = (I *)malloc(sizeof(I) * (size_t)(tot > 0 ? tot : 1));
        for (I i = 0; i < n; ++i)
            for (I k = cnt[i]; k < cnt[i + 1]; ++k) {
                const I p = colcnt[adj[k]]++;
                rowof[p] = i; colof[p] = adj[k]; wof[p] = w[k];
            }
        /* now entries are grouped by column ascending, rows ascending inside; scatter back by row */
        memcpy(fill, cnt, sizeof(I) * (size_t)n);
        for (I p = 0; p < tot; ++p) { const I i = rowof[p]; adj[fill[i]] = colof[p]; w[fill[i]++] = wof[p]; }
        free(colcnt); free(rowof); free(wof); free(colof);
        I pos = 0;
        for (I i = 0; i < n; ++i) {
            I k = cnt[i];
            while (k < cnt[i + 1]) {
                const I j = adj[k];
                double s = 0;
                int dup = 0;
                while (k < cnt[i + 1] && adj[k] == j) { s += w[k]; ++k; ++dup; }
                g.adj[pos] = j;
                g.w[pos] = weighted ? s : 1.0;
                g.deg[i] += g.w[pos];
                ++pos;
            }
            g.xadj[i + 1] = pos;
        }
    }
    free(adj); free(w); free(fill); free(cnt);

    /* components in order of smallest vertex */
    I *comp = (I *)malloc(sizeof(I) * (size_t)n), *queue = (I *)malloc(sizeof(I) * (size_t)n);
    I *loc = (I *)malloc(sizeof(I) * (size_t)n);
    for (I i = 0; i < n; ++i) comp[i] = -1;
    I outpos = 0, ncomp = 0;
    for (I s = 0; s < n; ++s) {
        if (comp[s] >= 0) continue;
        I head = 0, tail = 0;
        queue[tail++] = s; comp[s] = ncomp;
        while (head < tail) {
            const I v = queue[head++];
            for (I k = g.xadj[v]; k < g.xadj[v + 1]; ++k)
                if (comp[g.adj[k]] < 0) { comp[g.adj[k]] = ncomp; queue[tail++] = g.adj[k]; }
        }
        const I nc = tail;
        /* sort the component's vertices by index (insertion into a flag scan keeps it O(n) overall is not needed:
           qsort on I is fine and deterministic) */
        key_t2 *keys = (key_t2 *)malloc(sizeof(key_t2) * (size_t)nc);
        for (I t = 0; t < nc; ++t) { keys[t].v = -(double)queue[t]; keys[t].idx = queue[t]; }
        qsort(keys, (size_t)nc, sizeof(key_t2), cmp_desc); /* descending in -index == ascending in index */
        if (nc <= 2) {
            for (I t = 0; t < nc; ++t) { order[outpos++] = keys[t].idx; if (vec) vec[keys[t].idx] = (nc == 2) ? (t == 0 ? 0.7071067811865476 : -0.7071067811865476) : 0.0; }
            free(keys);
            ++ncomp;
            continue;
        }
        /* induced subgraph with local numbering in index order */
        for (I t = 0; t < nc; ++t) loc[keys[t].idx] = t;
        graph_t sg;
        sg.n = nc;
        sg.xadj = (I *)calloc((size_t)nc + 1, sizeof(I));
        I ne = 0;
        for (I t = 0; t < nc; ++t) ne += g.xadj[keys[t].idx + 1] - g.xadj[keys[t].idx];
        sg.adj = (I *)malloc(sizeof(I) * (size_t)(ne > 0 ? ne : 1));
        sg.w = (double *)malloc(sizeof(double) * (size_t)(ne > 0 ? ne : 1));
        sg.deg = (double *)malloc(sizeof(double) * (size_t)nc);
        I pos = 0;
        for (I t = 0; t < nc; ++t) {
            const I v = keys[t].idx;
            for (I k = g.xadj[v]; k < g.xadj[v + 1]; ++k) { sg.adj[pos] = loc[g.adj[k]]; sg.w[pos] = g.w[k]; ++pos; }
            sg.xadj[t + 1] = pos;
            sg.deg[t] = g.deg[v];
        }
        double *x = (double *)malloc(sizeof(double) * (size_t)nc);
        fiedler_vector(&sg, x, 0, use_device);
        /* sign: largest magnitude entry positive */
        I im = 0;
        for (I t = 1; t < nc; ++t) if (fabs(x[t]) > fabs(x[im])) im = t;
        if (x[im] < 0) for (I t = 0; t < nc; ++t) x[t] = -x[t];
        for (I t = 0; t < nc; ++t) { const I v = keys[t].idx; keys[t].v = x[t]; keys[t].idx = v; if (vec) vec[v] = x[t]; }
        qsort(keys, (size_t)nc, sizeof(key_t2), cmp_desc);
        for (I t = 0; t < nc; ++t) order[outpos++] = keys[t].idx;
        free(x); free(keys);
        graph_free(&sg);
        ++ncomp;
    }
    free(comp); free(queue); free(loc);
    graph_free(&g);
    return 0;
}

/*
 * Fiedler bisection + per-half second-stage ordering, the scheme the reference prototypes in
 * /root/reference/src/spectralPartition.c:326-417: order by the Fiedler vector (descending, :336-338), cut where the
 * vector changes sign (positive entries first, :331-333, 340-343), reorder the diagonal block of each half on its own
 * with a second ordering (there: -mat_ordering_type through MatGetOrdering on the two MatGetSubMatrix blocks, :369-381;
 * here: reverse Cuthill-McKee, the second stage of src/HOWTO:2 and src/testbed.c:236-284) and compose the two
 * permutations (:383-404).  halves_bw (optional, 4 entries) = bandwidth of the positive / negative block before and
 * after its reordering (what :377-382 prints).  order[k] = old index at new position k.
 */
int spike_rcm_order(int64_t n, const int64_t *ia, const int64_t *ja, int64_t *order);
int spike_profile_bandwidth(int64_t n, const int64_t *ia, const int64_t *ja, const int64_t *order, int64_t *profile,
                            int64_t *bandwidth);

int spike_fiedler_halves_order(int64_t n, const int64_t *ia, const int64_t *ja, const double *a, int64_t *order,
                               int64_t *pos_size, int64_t *halves_bw, int use_device)
{
    if (n <= 0 || !ia || !ja || !a || !order) return -1;
    double *vec = (double *)malloc(sizeof(double) * (size_t)n);
    I *of = (I *)malloc(sizeof(I) * (size_t)n), *newpos = (I *)malloc(sizeof(I) * (size_t)n);
    if (!vec || !of || !newpos || spike_fiedler_order_ex(n, ia, ja, a, of, vec, use_device)) { free(vec); free(of); free(newpos); return -1; }
    /* the reference counts strictly positive entries over the WHOLE vector (:331-333); with several components the
       per-component vectors are concatenated, so the cut is taken where the ordered sequence first becomes <= 0 */
    I np = 0;
    for (I k = 0; k < n; ++k) if (vec[of[k]] > 0.0) ++np;
    /* positions: positives first in Fiedler order, then the rest in Fiedler order (stable partition of `of`) */
    I *seq = (I *)malloc(sizeof(I) * (size_t)n);
    { I a0 = 0, b0 = np; for (I k = 0; k < n; ++k) { const I v = of[k]; if (vec[v] > 0.0) seq[a0++] = v; else seq[b0++] = v; } }
    for (I k = 0; k < n; ++k) newpos[seq[k]] = k;
    int rc = 0;
    for (int half = 0; half < 2 && !rc; ++half) {
        const I off = half == 0 ? 0 : np, m = half == 0 ? np : n - np;
        if (m <= 0) { if (halves_bw) { halves_bw[2 * half] = 0; halves_bw[2 * half + 1] = 0; } continue; }
        /* diagonal block of the permuted matrix: rows/columns = positions [off, off+m), local numbering */
        I *sia = (I *)calloc((size_t)m + 1, sizeof(I));
        for (I t = 0; t < m; ++t) {
            const I v = seq[off + t];
            for (I k = ia[v]; k < ia[v + 1]; ++k) { const I q = newpos[ja[k]]; if (q >= off && q < off + m) ++sia[t + 1]; }
        }
        for (I t = 0; t < m; ++t) sia[t + 1] += sia[t];
        I *sja = (I *)malloc(sizeof(I) * (size_t)(sia[m] > 0 ? sia[m] : 1));
        for (I t = 0; t < m; ++t) {
            const I v = seq[off + t];
            I w = sia[t];
            for (I k = ia[v]; k < ia[v + 1]; ++k) { const I q = newpos[ja[k]]; if (q >= off && q < off + m) sja[w++] = q - off; }
        }
        I *so = (I *)malloc(sizeof(I) * (size_t)m);
        if (spike_rcm_order(m, sia, sja, so)) rc = -1;
        else {
            if (halves_bw) {
                spike_profile_bandwidth(m, sia, sja, NULL, NULL, &halves_bw[2 * half]);
                spike_profile_bandwidth(m, sia, sja, so, NULL, &halves_bw[2 * half + 1]);
            }
            for (I t = 0; t < m; ++t) order[off + t] = seq[off + so[t]];   /* compose, :388-404 */
        }
        free(sia); free(sja); free(so);
    }
    if (pos_size) *pos_size = np;
    free(vec); free(of); free(newpos); free(seq);
    return rc;
}

/* profile and bandwidth of the symmetrised pattern under order[] (what petsc_mat_fiedler.c:51-52 prints) */
int spike_profile_bandwidth(int64_t n, const int64_t *ia, const int64_t *ja, const int64_t *order, int64_t *profile,
                            int64_t *bandwidth)
{
    if (n <= 0 || !ia || !ja) return -1;
    I *pos = (I *)malloc(sizeof(I) * (size_t)n), *first = (I *)malloc(sizeof(I) * (size_t)n);
    for (I k = 0; k < n; ++k) pos[order ? order[k] : k] = k;
    for (I k = 0; k < n; ++k) first[k] = k;
    I bw = 0;
    for (I i = 0; i < n; ++i)
        for (I k = ia[i]; k < ia[i + 1]; ++k) {
            const I pi = pos[i], pj = pos[ja[k]];
            const I lo = pi < pj ? pi : pj, hi = pi < pj ? pj : pi;
            if (lo < first[hi]) first[hi] = lo;
            if (hi - lo > bw) bw = hi - lo;
        }
    I prof = 0;
    for (I k = 0; k < n; ++k) prof += k - first[k];
    if (profile) *profile = prof;
    if (bandwidth) *bandwidth = bw;
    free(pos); free(first);
    return 0;
}
