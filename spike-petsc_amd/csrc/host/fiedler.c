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

#include "fiedler_steer.h"   /* fd_jacobi_eig, fd_eig3, fd_state and the scalar steps of the iteration: shared with the device build */

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

/* ---- the vectors of one refinement and the operations the iteration is written in ------------------------------------------
 * ids: 0 x, 1 Lx, 2 w, 3 Lw, 4 p, 5 Lp, 6 the constant 1.  Two implementations: host arrays (below) and the device
 * (spike_fd_* of libspike_mi355.so); both execute the same IEEE operations in the same order. */
#define FD_DEVICE_MIN 4096    /* smaller levels: the device needs ~60-80 us per iteration whatever n (14 dependent launches), the host ~17 ns per vertex and iteration; the result is the same either way */
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
int spike_fd_refine(spike_fd_ctx *c, double dmax, double rho, int maxit, int *its);

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
static void op_fill_alternating(vecs_t *V)
{
    if (V->dev) { if (spike_fd_fill_alternating(V->dev)) V->err = 1; return; }
    for (I i = 0; i < V->n; ++i) V->v[0][i] = (double)(i % 2 ? 1 : -1);
}
static void op_deflate(vecs_t *V, int a) { const double m = op_dot(V, a, 6) / (double)V->n; op_shift(V, a, m); }
static double op_normalize(vecs_t *V, int a) { const double s = sqrt(op_dot(V, a, a)); if (s > 0) op_div(V, s, a, -1); return s; }

/* ---- the vector steps of one iteration, host implementation: the element-wise statements of the step over all elements,
 * then that step's sums over the UPDATED values (device implementation: k_fd_* in spike_fiedler.hip, the same statements
 * chunk by chunk).  Scalars come from the state (fiedler_steer.h). ------------------------------------------------------------ */
static void h_resid_precond(vecs_t *V, const fd_state *st, double *sums)   /* sums: |w|^2 before preconditioning, sum of w after */
{
    double *x = V->v[0], *Lx = V->v[1], *w = V->v[2];
    if (st->scale) for (I i = 0; i < V->n; ++i) { x[i] /= st->xn; Lx[i] /= st->xn; }
    for (I i = 0; i < V->n; ++i) w[i] = Lx[i] - st->rho * x[i];
    sums[0] = dot(V->n, w, w);
    for (I i = 0; i < V->n; ++i) w[i] /= (V->g->deg[i] > 0 ? V->g->deg[i] : 1.0);
    sums[1] = dot(V->n, w, NULL);
}
static void h_shift_dots(vecs_t *V, const fd_state *st, double *sums)   /* w -= m; sums: w.x, p.x */
{
    for (I i = 0; i < V->n; ++i) V->v[2][i] -= st->m;
    sums[0] = dot(V->n, V->v[2], V->v[0]);
    sums[1] = st->havep ? dot(V->n, V->v[4], V->v[0]) : 0.0;
}
static void h_orth_p(vecs_t *V, const fd_state *st, double *sums)   /* w -= a x; p -= b x; Lp -= b Lx; sums: p.p, w.p */
{
    double *x = V->v[0], *Lx = V->v[1], *w = V->v[2], *p = V->v[4], *Lp = V->v[5];
    for (I i = 0; i < V->n; ++i) w[i] = w[i] - st->a * x[i];
    sums[0] = sums[1] = 0.0;
    if (!st->havep) return;
    for (I i = 0; i < V->n; ++i) { p[i] = p[i] - st->b * x[i]; Lp[i] = Lp[i] - st->b * Lx[i]; }
    sums[0] = dot(V->n, p, p);
    sums[1] = dot(V->n, w, p);
}
static void h_orth_w(vecs_t *V, const fd_state *st, double *sums)   /* p /= pn; Lp /= pn; w -= a2 p; sums: w.w */
{
    double *w = V->v[2], *p = V->v[4], *Lp = V->v[5];
    if (st->havep) for (I i = 0; i < V->n; ++i) { p[i] = p[i] / st->pn; Lp[i] = Lp[i] / st->pn; w[i] = w[i] - st->a2 * p[i]; }
    sums[0] = dot(V->n, w, w);
}
static void h_rr_dots(vecs_t *V, const fd_state *st, double *d6)   /* w /= wn; Lw = L w; the Rayleigh-Ritz products */
{
    double **v = V->v;
    for (I i = 0; i < V->n; ++i) v[2][i] /= st->wn;
    lap_mult(V->g, v[2], v[3]);
    d6[0] = dot(V->n, v[0], v[1]); d6[1] = dot(V->n, v[0], v[3]); d6[3] = dot(V->n, v[2], v[3]);
    d6[2] = d6[4] = d6[5] = 0.0;
    if (st->havep) { d6[2] = dot(V->n, v[0], v[5]); d6[4] = dot(V->n, v[2], v[5]); d6[5] = dot(V->n, v[4], v[5]); }
}
static void h_update(vecs_t *V, const fd_state *st, double *sums)   /* Rayleigh-Ritz update; sums: x.x */
{
    double *x = V->v[0], *Lx = V->v[1], *w = V->v[2], *Lw = V->v[3], *p = V->v[4], *Lp = V->v[5];
    const double c0 = st->c0, c1 = st->c1, c2 = st->c2;
    const int havep = st->havep;
    for (I i = 0; i < V->n; ++i) {
        const double pn = c1 * w[i] + (havep ? c2 * p[i] : 0.0);
        const double Lpn = c1 * Lw[i] + (havep ? c2 * Lp[i] : 0.0);
        x[i] = c0 * x[i] + pn;
        Lx[i] = c0 * Lx[i] + Lpn;
        p[i] = pn; Lp[i] = Lpn;
    }
    sums[0] = dot(V->n, x, x);
}

/* single-vector LOBPCG for the smallest eigenpair of L restricted to the complement of the constant vector.  An iteration is
   six vector steps, each followed by its scalar step (fiedler_steer.h): [residual + preconditioning], [deflation + w.x, p.x],
   [orthogonalisation against x + p.p, w.p], [normalise p, orthogonalise w against p + w.w], [normalise w, L w, the six
   Rayleigh-Ritz products], [update + x.x].  The Rayleigh quotient of an iterate is its Ritz value; p is normalised by the
   scalar p.p and w's coefficient against it is (w.p)/|p|; x and Lx are scaled to unit length at the start of the next
   iteration.  Host: the loop below.  Device: spike_fd_refine launches the same six kernels + six one-thread scalar steps
   per iteration, maxit times, with the state in device memory (steps after `done` are no-ops) -- no host round trip. */
static void refine_core(vecs_t *V, double dmax, int maxit)
{
    op_deflate(V, 0);
    if (op_normalize(V, 0) == 0.0) { op_fill_alternating(V); op_deflate(V, 0); op_normalize(V, 0); }
    op_lap(V, 0, 1);
    const double rho = op_dot(V, 0, 1);
    g_last_its = 0;
    if (V->err) return;
    if (V->dev) {
        if (spike_fd_refine(V->dev, dmax, rho, maxit, &g_last_its)) V->err = 1;
        return;
    }
    fd_state st;
    fd_init(&st, (double)V->n, dmax, rho);
    for (int it = 0; it < maxit && !st.done; ++it) {
        double s[6];
        h_resid_precond(V, &st, s); fd_after_resid(&st, s);
        if (st.done) break;
        h_shift_dots(V, &st, s);    fd_after_shift(&st, s);
        h_orth_p(V, &st, s);        fd_after_orth_p(&st, s);
        h_orth_w(V, &st, s);        fd_after_orth_w(&st, s);
        if (st.done) break;
        h_rr_dots(V, &st, s);       fd_after_rr_dots(&st, s);
        h_update(V, &st, s);        fd_after_update(&st, s);
    }
    g_last_its = st.its;
    if (st.scale) op_div(V, st.xn, 0, 1);
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
    const double t0 = now_s();
    if (use_device && n >= FD_DEVICE_MIN && spike_fd_create(n, g->xadj, g->adj, g->w, g->deg, x, &V.dev) == 0) {
        const double t1 = now_s();
        refine_core(&V, dmax, maxit);
        const double t2 = now_s();
        if (!V.err && spike_fd_download_x(V.dev, x)) V.err = 1;
        spike_fd_destroy(V.dev);
        if (getenv("SPIKE_FIEDLER_TRACE")) fprintf(stderr, "[fiedler]   device level: create %.3f ms, loop %.3f ms, download + destroy %.3f ms\n", 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (now_s() - t2));
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
            fd_jacobi_eig(m, A, V, ev);
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
    const double tc0 = now_s();
    I *map = coarsen(g, &c);
    if (getenv("SPIKE_FIEDLER_TRACE")) fprintf(stderr, "[fiedler] coarsen n=%lld -> %lld  %.3f ms\n", (long long)n, (long long)c.n, 1e3 * (now_s() - tc0));
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

/* keys[0..n) -> sorted by DESCENDING value, ties by ascending idx (exactly cmp_desc's total order; -0.0 counts as +0.0
   as it does for the comparison).  Large inputs: a stable LSD radix sort over the order-preserving 64-bit image of the
   doubles, four 16-bit digits -- stable + the input already ascending in idx gives the tie rule; requires that (it holds
   at both call sites: keys are filled in ascending idx).  qsort at n = 3.2e5 cost 50 ms, this 6 ms. */
static void sort_desc(key_t2 *keys, I n)
{
    int ascending_idx = 1;
    for (I t = 1; t < n && ascending_idx; ++t) if (keys[t - 1].idx >= keys[t].idx) ascending_idx = 0;
    key_t2 *tmp = (n >= 4096 && ascending_idx) ? (key_t2 *)malloc(sizeof(key_t2) * (size_t)n) : NULL;
    uint64_t *img = tmp ? (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n * 2) : NULL;
    I *cnt = img ? (I *)malloc(sizeof(I) * 65536) : NULL;
    if (!tmp || !img || !cnt) { free(tmp); free(img); free(cnt); qsort(keys, (size_t)n, sizeof(key_t2), cmp_desc); return; }
    uint64_t *ia = img, *ib = img + n;
    for (I t = 0; t < n; ++t) {
        const double v = keys[t].v + 0.0;                       /* -0.0 -> +0.0 */
        uint64_t u; memcpy(&u, &v, 8);
        u = (u >> 63) ? ~u : (u | 0x8000000000000000ULL);       /* ascending in u == ascending in v */
        ia[t] = ~u;                                             /* ascending in ~u == descending in v */
    }
    key_t2 *ka = keys, *kb = tmp;
    for (int pass = 0; pass < 4; ++pass) {
        const int sh = 16 * pass;
        memset(cnt, 0, sizeof(I) * 65536);
        for (I t = 0; t < n; ++t) ++cnt[(ia[t] >> sh) & 0xffff];
        I run = 0;
        for (int d = 0; d < 65536; ++d) { const I c = cnt[d]; cnt[d] = run; run += c; }
        for (I t = 0; t < n; ++t) { const I q = cnt[(ia[t] >> sh) & 0xffff]++; ib[q] = ia[t]; kb[q] = ka[t]; }
        uint64_t *ti = ia; ia = ib; ib = ti;
        key_t2 *tk = ka; ka = kb; kb = tk;
    }
    /* four passes: the result is back in keys */
    free(tmp); free(img); free(cnt);
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
    const int trace = getenv("SPIKE_FIEDLER_TRACE") != NULL;
    const double tg0 = now_s();
    const int weighted = (ia[n] > 0 && a[0] > 0.0);
    const double tol = 1e-12;
    graph_t g;
    memset(&g, 0, sizeof g);
    /* fast path, rows with strictly ascending columns (what MatGetRow of an assembled AIJ gives): transpose the kept
       pattern (rows of the transpose come out ascending by construction), then merge row i of A and of A^T; an edge stored
       on both sides gets |a_ij| + |a_ji| -- one IEEE addition, the same value whichever operand comes first, i.e. what the
       general path below computes.  103 -> 45 ms at n = 3.2e5. */
    int sorted_rows = 1;
    for (I i = 0; i < n && sorted_rows; ++i)
        for (I k = ia[i]; k < ia[i + 1]; ++k) {
            if (ja[k] < 0 || ja[k] >= n) return -1;
            if (k > ia[i] && ja[k] <= ja[k - 1]) { sorted_rows = 0; break; }
        }
    if (sorted_rows) {
        I *tp = (I *)calloc((size_t)n + 1, sizeof(I));
        for (I i = 0; i < n; ++i)
            for (I k = ia[i]; k < ia[i + 1]; ++k) if (ja[k] != i && fabs(a[k]) >= tol) ++tp[ja[k] + 1];
        for (I i = 0; i < n; ++i) tp[i + 1] += tp[i];
        const I nt = tp[n];
        I *tj = (I *)malloc(sizeof(I) * (size_t)(nt > 0 ? nt : 1)), *tfill = (I *)malloc(sizeof(I) * (size_t)n);
        double *tv = (double *)malloc(sizeof(double) * (size_t)(nt > 0 ? nt : 1));
        memcpy(tfill, tp, sizeof(I) * (size_t)n);
        for (I i = 0; i < n; ++i)
            for (I k = ia[i]; k < ia[i + 1]; ++k)
                if (ja[k] != i && fabs(a[k]) >= tol) { const I q = tfill[ja[k]]++; tj[q] = i; tv[q] = fabs(a[k]); }
        g.n = n;
        g.xadj = (I *)calloc((size_t)n + 1, sizeof(I));
        g.adj = (I *)malloc(sizeof(I) * (size_t)(2 * nt > 0 ? 2 * nt : 1));
        g.w = (double *)malloc(sizeof(double) * (size_t)(2 * nt > 0 ? 2 * nt : 1));
        g.deg = (double *)calloc((size_t)n, sizeof(double));
        I pos = 0;
        for (I i = 0; i < n; ++i) {
            I k = ia[i], q = tp[i];
            const I ke = ia[i + 1], qe = tp[i + 1];
            for (;;) {
                while (k < ke && (ja[k] == i || fabs(a[k]) < tol)) ++k;
                if (k >= ke && q >= qe) break;
                const I jc = (k < ke) ? ja[k] : n, jt = (q < qe) ? tj[q] : n;
                double sw;
                I j;
                if (jc < jt) { j = jc; sw = fabs(a[k]); ++k; }
                else if (jt < jc) { j = jt; sw = tv[q]; ++q; }
                else { j = jc; sw = fabs(a[k]) + tv[q]; ++k; ++q; }
                g.adj[pos] = j;
                g.w[pos] = weighted ? sw : 1.0;
                g.deg[i] += g.w[pos];
                ++pos;
            }
            g.xadj[i + 1] = pos;
        }
        free(tp); free(tj); free(tfill); free(tv);
    } else {
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
    }
    if (trace) fprintf(stderr, "[fiedler] graph build n=%lld  %.3f ms\n", (long long)n, 1e3 * (now_s() - tg0));

    /* components in order of smallest vertex */
    I *comp = (I *)malloc(sizeof(I) * (size_t)n), *queue = (I *)malloc(sizeof(I) * (size_t)n);
    I *loc = (I *)malloc(sizeof(I) * (size_t)n);
    for (I i = 0; i < n; ++i) comp[i] = -1;
    I outpos = 0, ncomp = 0;
    for (I s = 0; s < n; ++s) {
        if (comp[s] >= 0) continue;
        const double tcomp0 = now_s();
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
        if (nc > n / 8) {   /* a large component: its vertices in index order by a scan of the labels */
            I t = 0;
            for (I v = s; v < n && t < nc; ++v) if (comp[v] == ncomp) { keys[t].v = 0.0; keys[t].idx = v; ++t; }
        } else {
            for (I t = 0; t < nc; ++t) { keys[t].v = -(double)queue[t]; keys[t].idx = queue[t]; }
            qsort(keys, (size_t)nc, sizeof(key_t2), cmp_desc); /* descending in -index == ascending in index */
        }
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
        if (trace && nc > 1000) fprintf(stderr, "[fiedler] component of %lld vertices: found + subgraph %.3f ms\n", (long long)nc, 1e3 * (now_s() - tcomp0));
        const double tv0 = now_s();
        fiedler_vector(&sg, x, 0, use_device);
        if (trace && nc > 1000) fprintf(stderr, "[fiedler] component of %lld vertices: vector %.3f ms\n", (long long)nc, 1e3 * (now_s() - tv0));
        const double tv1 = now_s();
        /* sign: largest magnitude entry positive */
        I im = 0;
        for (I t = 1; t < nc; ++t) if (fabs(x[t]) > fabs(x[im])) im = t;
        if (x[im] < 0) for (I t = 0; t < nc; ++t) x[t] = -x[t];
        for (I t = 0; t < nc; ++t) { const I v = keys[t].idx; keys[t].v = x[t]; keys[t].idx = v; if (vec) vec[v] = x[t]; }
        sort_desc(keys, nc);
        for (I t = 0; t < nc; ++t) order[outpos++] = keys[t].idx;
        if (trace && nc > 1000) fprintf(stderr, "[fiedler] component of %lld vertices: sign + sort %.3f ms\n", (long long)nc, 1e3 * (now_s() - tv1));
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
