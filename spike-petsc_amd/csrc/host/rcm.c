/*
 * rcm.c -- reverse Cuthill-McKee on the symmetrised pattern.
 *
 * The reference takes "rcm" from PETSc's built-in orderings (second-stage reordering in /root/reference/src/testbed.c:236-284
 * via -mat_ordering_type2 rcm, the run line of /root/reference/src/HOWTO:2, and the sample options of
 * /root/reference/src/testbed2.c:4).  PETSc is absent, so the ordering is restated here with a fixed, deterministic rule:
 * per connected component (in order of smallest vertex) start from a pseudo-peripheral vertex (George-Liu: repeat BFS
 * from a minimum-degree vertex of the last level until the eccentricity stops growing), Cuthill-McKee BFS visiting
 * neighbours by increasing degree (ties by index), then reverse the whole order.  order[k] = old index at position k.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t I;

typedef struct { I deg, idx; } dk_t;
static int dk_cmp(const void *a, const void *b)
{
    const dk_t *x = (const dk_t *)a, *y = (const dk_t *)b;
    if (x->deg != y->deg) return x->deg < y->deg ? -1 : 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

/* BFS from s inside the component; fills level[] (>=0 visited in this pass via stamp), returns the eccentricity and the
 * min-degree vertex of the last level in *far */
static I bfs(I s, const I *xadj, const I *adj, I *level, I *queue, I *stampv, I stamp, I *far)
{
    I head = 0, tail = 0;
    queue[tail++] = s; level[s] = 0; stampv[s] = stamp;
    while (head < tail) {
        const I v = queue[head++];
        for (I k = xadj[v]; k < xadj[v + 1]; ++k) {
            const I u = adj[k];
            if (stampv[u] == stamp) continue;
            stampv[u] = stamp; level[u] = level[v] + 1; queue[tail++] = u;
        }
    }
    const I ecc = level[queue[tail - 1]];
    I best = queue[tail - 1];
    for (I t = tail - 1; t >= 0 && level[queue[t]] == ecc; --t) {
        const I v = queue[t];
        const I dv = xadj[v + 1] - xadj[v], db = xadj[best + 1] - xadj[best];
        if (dv < db || (dv == db && v < best)) best = v;
    }
    *far = best;
    return ecc;
}

int spike_rcm_order(int64_t n, const int64_t *ia, const int64_t *ja, int64_t *order)
{
    if (n <= 0 || !ia || !ja || !order) return -1;
    /* symmetrised adjacency without the diagonal, duplicates removed */
    I *cnt = (I *)calloc((size_t)n + 1, sizeof(I));
    for (I i = 0; i < n; ++i)
        for (I k = ia[i]; k < ia[i + 1]; ++k) {
            if (ja[k] < 0 || ja[k] >= n) { free(cnt); return -1; }
            if (ja[k] != i) { ++cnt[i + 1]; ++cnt[ja[k] + 1]; }
        }
    for (I i = 0; i < n; ++i) cnt[i + 1] += cnt[i];
    I *tmp = (I *)malloc(sizeof(I) * (size_t)(cnt[n] > 0 ? cnt[n] : 1)), *fill = (I *)malloc(sizeof(I) * (size_t)n);
    memcpy(fill, cnt, sizeof(I) * (size_t)n);
    for (I i = 0; i < n; ++i)
        for (I k = ia[i]; k < ia[i + 1]; ++k)
            if (ja[k] != i) { tmp[fill[i]++] = ja[k]; tmp[fill[ja[k]]++] = i; }
    I *xadj = (I *)calloc((size_t)n + 1, sizeof(I)), *adj = (I *)malloc(sizeof(I) * (size_t)(cnt[n] > 0 ? cnt[n] : 1));
    I *mark = (I *)malloc(sizeof(I) * (size_t)n);
    for (I i = 0; i < n; ++i) mark[i] = -1;
    I pos = 0;
    for (I i = 0; i < n; ++i) {
        for (I k = cnt[i]; k < cnt[i + 1]; ++k)
            if (mark[tmp[k]] != i) { mark[tmp[k]] = i; adj[pos++] = tmp[k]; }
        xadj[i + 1] = pos;
    }
    free(tmp); free(fill); free(cnt);
    I *level = (I *)malloc(sizeof(I) * (size_t)n), *queue = (I *)malloc(sizeof(I) * (size_t)n), *stampv = (I *)malloc(sizeof(I) * (size_t)n);
    char *done = (char *)calloc((size_t)n, 1);
    for (I i = 0; i < n; ++i) stampv[i] = -1;
    I stamp = 0, out = 0;
    dk_t *nb = (dk_t *)malloc(sizeof(dk_t) * (size_t)n);
    for (I s0 = 0; s0 < n; ++s0) {
        if (done[s0]) continue;
        /* pseudo-peripheral start: min-degree vertex of the component, then walk outwards */
        I far, s = s0;
        I ecc = bfs(s, xadj, adj, level, queue, stampv, stamp++, &far);
        for (int it = 0; it < 8; ++it) {
            I far2;
            const I e2 = bfs(far, xadj, adj, level, queue, stampv, stamp++, &far2);
            if (e2 <= ecc) { if (e2 == ecc && far < s) s = far; break; }
            s = far; ecc = e2; far = far2;
        }
        /* Cuthill-McKee from s */
        const I first = out;
        order[out++] = s; done[s] = 1;
        for (I head = first; head < out; ++head) {
            const I v = order[head];
            I m = 0;
            for (I k = xadj[v]; k < xadj[v + 1]; ++k) {
                const I u = adj[k];
                if (done[u]) continue;
                done[u] = 1;
                nb[m].deg = xadj[u + 1] - xadj[u]; nb[m].idx = u; ++m;
            }
            qsort(nb, (size_t)m, sizeof(dk_t), dk_cmp);
            for (I t = 0; t < m; ++t) order[out++] = nb[t].idx;
        }
    }
    for (I a = 0, b = n - 1; a < b; ++a, --b) { const I t = order[a]; order[a] = order[b]; order[b] = t; }
    free(nb); free(done); free(level); free(queue); free(stampv); free(mark); free(xadj); free(adj);
    return 0;
}
