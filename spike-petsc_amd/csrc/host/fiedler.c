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
 * All arithmetic is sequential fp64 in a fixed order => the permutation is reproducible bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

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

static double dot(I n, const double *a, const double *b) { double s = 0; for (I i = 0; i < n; ++i) s += a[i] * b[i]; return s; }
static void deflate(I n, double *x) { double m = 0; for (I i = 0; i < n; ++i) m += x[i]; m /= (double)n; for (I i = 0; i < n; ++i) x[i] -= m; }
static double normalize(I n, double *x) { double s = sqrt(dot(n, x, x)); if (s > 0) for (I i = 0; i < n; ++i) x[i] /= s; return s; }

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

/* single-vector LOBPCG for the smallest eigenpair of L restricted to the complement of the constant vector */
static void refine(const graph_t *g, double *x, int maxit)
{
    const I n = g->n;
    double *Lx = (double *)malloc(sizeof(double) * (size_t)n * 6);
    double *w = Lx + n, *Lw = w + n, *p = Lw + n, *Lp = p + n, *t = Lp + n;
    double dmax = 0;
    for (I i = 0; i < n; ++i) if (g->deg[i] > dmax) dmax = g->deg[i];
    deflate(n, x);
    if (normalize(n, x) == 0.0) { for (I i = 0; i < n; ++i) x[i] = (double)(i % 2 ? 1 : -1); deflate(n, x); normalize(n, x); }
    lap_mult(g, x, Lx);
    int havep = 0;
    for (int it = 0; it < maxit; ++it) {
        const double rho = dot(n, x, Lx);
        double rn = 0;
        for (I i = 0; i < n; ++i) { const double r = Lx[i] - rho * x[i]; w[i] = r; rn += r * r; }
        if (sqrt(rn) <= 1e-9 * dmax) break;
        for (I i = 0; i < n; ++i) w[i] /= (g->deg[i] > 0 ? g->deg[i] : 1.0);
        deflate(n, w);
        /* orthogonalise w against x (and p), normalise */
        double a = dot(n, w, x);
        for (I i = 0; i < n; ++i) w[i] -= a * x[i];
        if (havep) {
            a = dot(n, p, x);
            for (I i = 0; i < n; ++i) { p[i] -= a * x[i]; Lp[i] -= a * Lx[i]; }
            const double pn = sqrt(dot(n, p, p));
            if (pn > 1e-300) { for (I i = 0; i < n; ++i) { p[i] /= pn; Lp[i] /= pn; } }
            else havep = 0;
        }
        if (havep) { a = dot(n, w, p); for (I i = 0; i < n; ++i) w[i] -= a * p[i]; }
        if (normalize(n, w) < 1e-300) break;
        lap_mult(g, w, Lw);
        const int m = havep ? 3 : 2;
        double G[3][3], c[3], lam;
        const double *S[3] = {x, w, p}, *LS[3] = {Lx, Lw, Lp};
        for (int i = 0; i < m; ++i) for (int j = i; j < m; ++j) G[i][j] = G[j][i] = dot(n, S[i], LS[j]);
        eig3(m, G, c, &lam);
        if (c[0] < 0) for (int i = 0; i < m; ++i) c[i] = -c[i];
        for (I i = 0; i < n; ++i) {
            const double pn = c[1] * w[i] + (havep ? c[2] * p[i] : 0.0);
            const double Lpn = c[1] * Lw[i] + (havep ? c[2] * Lp[i] : 0.0);
            t[i] = c[0] * x[i] + pn;
            Lx[i] = c[0] * Lx[i] + Lpn;
            p[i] = pn; Lp[i] = Lpn;
        }
        memcpy(x, t, sizeof(double) * (size_t)n);
        havep = 1;
        const double xn = sqrt(dot(n, x, x));
        for (I i = 0; i < n; ++i) { x[i] /= xn; Lx[i] /= xn; }
    }
    free(Lx);
}

static void fiedler_vector(const graph_t *g, double *x, int level)
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
        refine(g, x, 300);
        return;
    }
    graph_t c;
    memset(&c, 0, sizeof c);
    I *map = coarsen(g, &c);
    if (c.n > (9 * n) / 10) { /* matching stalls (e.g. star graphs): stop coarsening here */
        for (I i = 0; i < n; ++i) x[i] = (double)i - 0.5 * (double)(n - 1);
        refine(g, x, 1000);
    } else {
        double *xc = (double *)malloc(sizeof(double) * (size_t)c.n);
        fiedler_vector(&c, xc, level + 1);
        for (I i = 0; i < n; ++i) x[i] = xc[map[i]];
        free(xc);
        refine(g, x, 300);
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
int spike_fiedler_order(int64_t n, const int64_t *ia, const int64_t *ja, const double *a, int64_t *order, double *vec)
{
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
        fiedler_vector(&sg, x, 0);
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
