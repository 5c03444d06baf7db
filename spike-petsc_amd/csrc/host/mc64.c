/*
 * mc64.c -- weighted bipartite matching "job 5" (maximum product of the matched entries plus
 * logarithmic row/column scalings), the algorithm of Duff & Koster (SIMAX 22(4), 2001) as the
 * reference runs it:
 *
 *   wrapper   MatGetOrdering_WBM          /root/reference/src/petsc_mat_wbm.c:13-61  (job = 5, icntl = {0,0,0,0,4})
 *   driver    HSLmc64AD (job 5 branch)    /root/reference/src/hslmc64.c:305-976  (costs :703-743, post-scaling :822-832)
 *   matching  HSLmc64WD                   /root/reference/src/hslmc64.c:1917-2380
 *   heap      mc64DD / mc64ED / mc64FD    /root/reference/src/hslmc64.c:3962, 4044, 4140   (iway = 2: min-heap)
 *   completion HSLmc64XD                  /root/reference/src/hslmc64.c:2555-2607
 *
 * This is a fresh, re-entrant restatement (no static locals, no gotos, no f2c scaffolding); what it keeps
 * from the reference are the DECISIONS that fix the result when the optimum is not unique:
 *   (i)   cost c = log(colmax) - log|a|, exact zero -> "infinite" cost; a column of zeros keeps RINF/n   (:706-736)
 *   (ii)  dual initialisation keeps the LAST minimum met in column-scan order (test is "a > u : skip")      (:1977-1982)
 *   (iii) cheap assignment in row order, skipping columns with more than n/10 entries when n > 50          (:1989-2009)
 *   (iv)  second pass: "di > vj skip / di < vj or di == RINF take / tie: prefer an unmatched row", then one
 *         augmentation step with the persistent column cursors pr[]                                       (:2016-2099)
 *   (v)   Dijkstra per unmatched column with the two-part queue sharing ONE array q: a binary min-heap in
 *         q[1..qlen] and the set of rows at distance dmin in q[low..up-1], filled downward from n+1;
 *         comparisons >=, <=, > exactly as in the reference                                               (:2114-2275)
 *   (vi)  augmentation through pr[]/out[], dual update only for the rows popped from the queue            (:2284-2312)
 *
 * Host C like the reference's (sequential, pointer chasing; SURVEY.md 8f-2 lists a device version as "next").
 * Indices inside are 1-based on purpose: the reference uses 0 as "none" in iperm/jperm/l.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t I;

typedef struct {
    I *q, *l;
    double *d;
} heap_t;

/* sift row i up (reference mc64DD, iway = 2) */
static void heap_up(heap_t *h, I i)
{
    I pos = h->l[i];
    if (pos > 1) {
        const double di = h->d[i];
        for (;;) {
            const I posk = pos / 2, qk = h->q[posk];
            if (di >= h->d[qk]) break;
            h->q[pos] = qk;
            h->l[qk] = pos;
            pos = posk;
            if (pos <= 1) break;
        }
    }
    h->q[pos] = i;
    h->l[i] = pos;
}

/* delete the root (reference mc64ED, iway = 2) */
static void heap_pop(heap_t *h, I *qlen)
{
    const I i = h->q[*qlen];
    const double di = h->d[i];
    --(*qlen);
    I pos = 1;
    for (;;) {
        I posk = pos * 2;
        if (posk > *qlen) break;
        double dk = h->d[h->q[posk]];
        if (posk < *qlen) {
            const double dr = h->d[h->q[posk + 1]];
            if (dk > dr) { ++posk; dk = dr; }
        }
        if (di <= dk) break;
        h->q[pos] = h->q[posk];
        h->l[h->q[pos]] = pos;
        pos = posk;
    }
    h->q[pos] = i;
    h->l[i] = pos;
}

/* delete the element at position pos0 (reference mc64FD, iway = 2) */
static void heap_delete(heap_t *h, I pos0, I *qlen)
{
    if (*qlen == pos0) { --(*qlen); return; }
    const I i = h->q[*qlen];
    const double di = h->d[i];
    --(*qlen);
    I pos = pos0;
    if (pos > 1) {
        for (;;) {
            const I posk = pos / 2, qk = h->q[posk];
            if (di >= h->d[qk]) break;
            h->q[pos] = qk;
            h->l[qk] = pos;
            pos = posk;
            if (pos <= 1) break;
        }
    }
    h->q[pos] = i;
    h->l[i] = pos;
    if (pos != pos0) return;
    for (;;) {
        I posk = pos * 2;
        if (posk > *qlen) break;
        double dk = h->d[h->q[posk]];
        if (posk < *qlen) {
            const double dr = h->d[h->q[posk + 1]];
            if (dk > dr) { ++posk; dk = dr; }
        }
        if (di <= dk) break;
        const I qk = h->q[posk];
        h->q[pos] = qk;
        h->l[qk] = pos;
        pos = posk;
    }
    h->q[pos] = i;
    h->l[i] = pos;
}

/* the matching proper on costs a[] (1-based CSC ip/irn); square n x n */
static void wd(I n, const I *ip, const I *irn, const double *a, I *iperm, I *num_out, I *jperm, I *out, I *pr, I *q,
               I *l, double *u, double *d)
{
    const double RINF = DBL_MAX;
    const I m = n;
    I num = 0;
    I isp = 0, jsp = 0;
    heap_t hp = {q, l, d};

    for (I k = 1; k <= n; ++k) { d[k] = 0.0; jperm[k] = 0; pr[k] = ip[k]; }
    for (I k = 1; k <= m; ++k) { u[k] = RINF; iperm[k] = 0; l[k] = 0; }
    /* (ii) dual initialisation */
    for (I j = 1; j <= n; ++j)
        for (I k = ip[j]; k <= ip[j + 1] - 1; ++k) {
            const I i = irn[k];
            if (a[k] > u[i]) continue;
            u[i] = a[k];
            iperm[i] = j;
            l[i] = k;
        }
    /* (iii) cheap assignment */
    for (I i = 1; i <= m; ++i) {
        const I j = iperm[i];
        if (j == 0) continue;
        iperm[i] = 0;
        if (jperm[j] != 0) continue;
        if (ip[j + 1] - ip[j] > n / 10 && n > 50) continue;
        ++num;
        iperm[i] = j;
        jperm[j] = l[i];
    }
    if (num != n) {
        /* (iv) second pass over the unassigned columns */
        for (I j = 1; j <= n; ++j) {
            if (jperm[j] != 0) continue;
            const I k1 = ip[j], k2 = ip[j + 1] - 1;
            if (k1 > k2) continue;
            I i0 = irn[k1], k0 = k1;
            double vj = a[k1] - u[i0];
            for (I k = k1 + 1; k <= k2; ++k) {
                const I i = irn[k];
                const double di = a[k] - u[i];
                if (di > vj) continue;
                if (!(di < vj || di == RINF)) {
                    if (iperm[i] != 0 || iperm[i0] == 0) continue;
                }
                vj = di; i0 = i; k0 = k;
            }
            d[j] = vj;
            I k = k0, i = i0;
            int assign = (iperm[i] == 0);
            if (!assign) {
                I jj = 0, kk = 0, ii = 0;
                int found = 0;
                for (k = k0; k <= k2 && !found; ++k) {
                    i = irn[k];
                    if (a[k] - u[i] > vj) continue;
                    jj = iperm[i];
                    if (jj <= 0) continue; /* outside the reference's defined behaviour */
                    const I kk1 = pr[jj], kk2 = ip[jj + 1] - 1;
                    if (kk1 > kk2) continue;
                    for (kk = kk1; kk <= kk2; ++kk) {
                        ii = irn[kk];
                        if (iperm[ii] > 0) continue;
                        if (a[kk] - u[ii] <= d[jj]) { found = 1; break; }
                    }
                    if (found) break;
                    pr[jj] = kk2 + 1;
                }
                if (found) {
                    jperm[jj] = kk;
                    iperm[ii] = jj;
                    pr[jj] = kk + 1;
                    assign = 1; /* with k, i as they stand at the break */
                }
            }
            if (assign) {
                ++num;
                jperm[j] = k;
                iperm[i] = j;
                pr[j] = k + 1;
            }
        }
    }
    if (num != n) {
        for (I i = 1; i <= m; ++i) { d[i] = RINF; l[i] = 0; }
        /* (v) main loop: shortest augmenting path from every unmatched column */
        for (I jord = 1; jord <= n; ++jord) {
            if (jperm[jord] != 0) continue;
            double dmin = RINF, csp = RINF;
            I qlen = 0, low = n + 1, up = n + 1;
            I j = jord;
            pr[j] = -1;
            for (I k = ip[j]; k <= ip[j + 1] - 1; ++k) {
                const I i = irn[k];
                const double dnew = a[k] - u[i];
                if (dnew >= csp) continue;
                if (iperm[i] == 0) { csp = dnew; isp = k; jsp = j; }
                else {
                    if (dnew < dmin) dmin = dnew;
                    d[i] = dnew;
                    ++qlen;
                    q[qlen] = k;
                }
            }
            I q0 = qlen;
            qlen = 0;
            for (I kk = 1; kk <= q0; ++kk) {
                const I k = q[kk];
                const I i = irn[k];
                if (csp <= d[i]) { d[i] = RINF; continue; }
                if (d[i] <= dmin) { --low; q[low] = i; l[i] = low; }
                else { ++qlen; l[i] = qlen; heap_up(&hp, i); }
                const I jj = iperm[i];
                out[jj] = k;
                pr[jj] = j;
            }
            for (I jdum = 1; jdum <= num; ++jdum) {
                if (low == up) {
                    if (qlen == 0) break;
                    I i = q[1];
                    if (d[i] >= csp) break;
                    dmin = d[i];
                    for (;;) {
                        heap_pop(&hp, &qlen);
                        --low; q[low] = i; l[i] = low;
                        if (qlen == 0) break;
                        i = q[1];
                        if (d[i] > dmin) break;
                    }
                }
                q0 = q[up - 1];
                const double dq0 = d[q0];
                if (dq0 >= csp) break;
                --up;
                j = iperm[q0];
                const double vj = dq0 - a[jperm[j]] + u[q0];
                for (I k = ip[j]; k <= ip[j + 1] - 1; ++k) {
                    const I i = irn[k];
                    if (l[i] >= up) continue;
                    const double dnew = vj + a[k] - u[i];
                    if (dnew >= csp) continue;
                    if (iperm[i] == 0) { csp = dnew; isp = k; jsp = j; }
                    else {
                        const double di = d[i];
                        if (di <= dnew) continue;
                        if (l[i] >= low) continue;
                        d[i] = dnew;
                        if (dnew <= dmin) {
                            const I lpos = l[i];
                            if (lpos != 0) heap_delete(&hp, lpos, &qlen);
                            --low; q[low] = i; l[i] = low;
                        } else {
                            if (l[i] == 0) { ++qlen; l[i] = qlen; }
                            heap_up(&hp, i);
                        }
                        const I jj = iperm[i];
                        out[jj] = k;
                        pr[jj] = j;
                    }
                }
            }
            if (csp != RINF) {
                /* (vi) augment and update the duals of the rows that left the queue */
                ++num;
                I i = irn[isp];
                iperm[i] = jsp;
                jperm[jsp] = isp;
                j = jsp;
                for (I jdum = 1; jdum <= num; ++jdum) {
                    const I jj = pr[j];
                    if (jj == -1) break;
                    const I k = out[j];
                    i = irn[k];
                    iperm[i] = jj;
                    jperm[jj] = k;
                    j = jj;
                }
                for (I kk = up; kk <= n; ++kk) {
                    const I r = q[kk];
                    u[r] = u[r] + d[r] - csp;
                }
            }
            for (I kk = low; kk <= n; ++kk) { const I r = q[kk]; d[r] = RINF; l[r] = 0; }
            for (I kk = 1; kk <= qlen; ++kk) { const I r = q[kk]; d[r] = RINF; l[r] = 0; }
        }
    }
    /* dual column variables; unmatched rows get u = 0 (:2333-2350) */
    for (I j = 1; j <= n; ++j) {
        const I k = jperm[j];
        d[j] = (k != 0) ? a[k] - u[irn[k]] : 0.0;
    }
    for (I i = 1; i <= m; ++i)
        if (iperm[i] == 0) u[i] = 0.0;
    if (!(num == n)) {
        /* completion with negative entries (reference HSLmc64XD); l and jperm are work arrays */
        I *rw = l, *cw = jperm;
        for (I j = 1; j <= n; ++j) cw[j] = 0;
        I k = 0;
        for (I i = 1; i <= m; ++i) {
            if (iperm[i] == 0) { ++k; rw[k] = i; }
            else cw[iperm[i]] = i;
        }
        k = 0;
        for (I j = 1; j <= n; ++j) {
            if (cw[j] != 0) continue;
            ++k;
            iperm[rw[k]] = -j;
        }
    }
    *num_out = num;
}

/*
 * spike_mc64_job5: square n x n, 0-based CSC on entry (the reference's wrapper hands the CSR arrays of A to a
 * CSC interface, i.e. it matches A^T -- do the same on the caller's side to reproduce it, petsc_mat_wbm.c:29,52).
 * Outputs: perm[i] = column matched to row i, 0-based; a row completed by the singular-case fill-in gets
 * -(j+1) (the reference's negative 1-based entry); u[i], v[j] natural-log scalings; *num = matching size.
 * Returns 0, or -1 on bad arguments / allocation failure.
 */
int spike_mc64_job5(int64_t n, const int64_t *colptr, const int64_t *rowind, const double *val, int64_t *perm,
                    double *u_out, double *v_out, int64_t *num_out)
{
    if (n <= 0 || !colptr || !rowind || !val || !perm) return -1;
    const I ne = colptr[n];
    I *ip = (I *)malloc(sizeof(I) * (size_t)(n + 2));
    I *irn = (I *)malloc(sizeof(I) * (size_t)(ne + 1));
    double *c = (double *)malloc(sizeof(double) * (size_t)(ne + 1));
    double *colmax = (double *)malloc(sizeof(double) * (size_t)(n + 1));
    I *iw = (I *)calloc((size_t)(6 * (n + 1)), sizeof(I));
    double *dw = (double *)calloc((size_t)(2 * (n + 1)), sizeof(double));
    if (!ip || !irn || !c || !colmax || !iw || !dw) { free(ip); free(irn); free(c); free(colmax); free(iw); free(dw); return -1; }
    for (I j = 0; j <= n; ++j) ip[j + 1] = colptr[j] + 1;
    for (I k = 0; k < ne; ++k) irn[k + 1] = rowind[k] + 1;
    /* (i) costs, hslmc64.c:407-408 and :703-743 */
    const double rinf_ad = DBL_MAX / (double)n;
    for (I j = 1; j <= n; ++j) {
        double fact = 0.0;
        for (I k = ip[j]; k <= ip[j + 1] - 1; ++k) {
            c[k] = fabs(val[k - 1]);
            if (c[k] > fact) fact = c[k];
        }
        colmax[j] = fact;
        fact = (fact != 0.0) ? log(fact) : rinf_ad;
        for (I k = ip[j]; k <= ip[j + 1] - 1; ++k) c[k] = (c[k] != 0.0) ? fact - log(c[k]) : rinf_ad;
    }
    I *iperm = iw, *jperm = iw + (n + 1), *out = iw + 2 * (n + 1), *pr = iw + 3 * (n + 1), *q = iw + 4 * (n + 1),
      *l = iw + 5 * (n + 1);
    double *u = dw, *d = dw + (n + 1);
    I num = 0;
    wd(n, ip, irn, c, iperm, &num, jperm, out, pr, q, l, u, d);
    /* hslmc64.c:822-832 */
    if (num == n)
        for (I j = 1; j <= n; ++j) d[j] = (colmax[j] != 0.0) ? d[j] - log(colmax[j]) : 0.0;
    for (I i = 1; i <= n; ++i) perm[i - 1] = iperm[i] > 0 ? iperm[i] - 1 : iperm[i];
    if (u_out) for (I i = 1; i <= n; ++i) u_out[i - 1] = u[i];
    if (v_out) for (I j = 1; j <= n; ++j) v_out[j - 1] = d[j];
    if (num_out) *num_out = num;
    free(ip); free(irn); free(c); free(colmax); free(iw); free(dw);
    return 0;
}
