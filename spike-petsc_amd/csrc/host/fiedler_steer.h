/*
 * fiedler_steer.h -- the SCALAR side of one LOBPCG iteration of the Fiedler refinement (spec: fiedler.c header, items 3
 * and 6): everything that is computed from a step's sums and steers the next step -- square roots, the quotients, the
 * 3 x 3 Rayleigh-Ritz eigenproblem, the sign rule, the two stopping tests.
 *
 * ONE copy, two compilations: gcc builds it into fiedler.c (host refinement, the state lives on the stack), hipcc builds
 * it into spike_fiedler.hip, where each step runs as a one-thread epilogue between the vector kernels and the state lives
 * in device memory -- the host then launches a level's 300 iterations without reading a single scalar back.  Only
 * operations IEEE 754 defines exactly are used (+ - * / sqrt fabs, comparisons), one at a time (no contraction: the
 * pragma below for clang/hipcc, -ffp-contract=off for gcc), so both compilations produce the same bits
 * (tests/test_host_gpu.py::test_fiedler_device_equals_host_bit_for_bit).
 *
 * Reference slot: MatGetOrdering_Fiedler, /root/reference/src/petsc_mat_fiedler.c:11-58 (HSL_MC73 absent: parity unpinned).
 */
#ifndef SPIKE_FIEDLER_STEER_H
#define SPIKE_FIEDLER_STEER_H

#ifndef FD_HD
#define FD_HD static inline
#endif
#ifdef __clang__
#pragma clang fp contract(off)
#endif

typedef struct fd_state {
    double n_d;          /* (double) number of vertices */
    double dmax;         /* largest weighted degree: the stopping test is ||L x - rho x|| <= 1e-9 dmax */
    double rho, xn;      /* Rayleigh quotient of the iterate (= last Ritz value); its norm before scaling */
    double m;            /* mean removed from the preconditioned residual */
    double a, b;         /* w.x, p.x */
    double pn, a2;       /* |p|, (w.p)/|p| */
    double wn;           /* |w| */
    double c0, c1, c2;   /* Ritz vector in the basis x, w, p */
    int havep, scale, done, its;
} fd_state;

/* cyclic Jacobi for a dense symmetric matrix (n <= ~64): eigenvalues in ev, eigenvectors in the columns of V */
FD_HD void fd_jacobi_eig(int n, double *A, double *V, double *ev)
{
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
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

/* smallest eigenpair of the leading m x m part (m = 2, 3) of G.  The two sizes are spelled out with a literal n so that
   the inlined Jacobi loops have constant bounds: the device build then keeps the 3 x 3 matrices in registers instead of
   scratch memory (20 -> 8 us per call on one lane); the operations and their order are those of fd_jacobi_eig either way. */
FD_HD void fd_eig3(int m, double G[3][3], double c[3], double *lam)
{
    double A[9], V[9], ev[3];
    int b = 0;
    if (m == 3) {
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i * 3 + j] = G[i][j];
        fd_jacobi_eig(3, A, V, ev);
        for (int i = 1; i < 3; ++i) if (ev[i] < ev[b]) b = i;
        for (int i = 0; i < 3; ++i) c[i] = V[i * 3 + b];
    } else {
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) A[i * 2 + j] = G[i][j];
        fd_jacobi_eig(2, A, V, ev);
        for (int i = 1; i < 2; ++i) if (ev[i] < ev[b]) b = i;
        for (int i = 0; i < 2; ++i) c[i] = V[i * 2 + b];
    }
    *lam = ev[b];
}

FD_HD void fd_init(fd_state *s, double n_d, double dmax, double rho)
{
    s->n_d = n_d; s->dmax = dmax; s->rho = rho; s->xn = 1.0;
    s->m = s->a = s->b = 0.0; s->pn = 1.0; s->a2 = 0.0; s->wn = 1.0; s->c0 = s->c1 = s->c2 = 0.0;
    s->havep = 0; s->scale = 0; s->done = 0; s->its = 0;
}

/* The iteration, step by step.  Vector step k (fiedler.c: op_*, spike_fiedler.hip: k_fd_*) reads its scalars from the
   state, produces sums; fd_after_k turns them into the scalars of step k+1.  After `done` every step is a no-op. */

/* after [x /= xn, Lx /= xn if scale]; w = Lx - rho x; sums[0] = w.w; w /= deg; sums[1] = sum w */
FD_HD void fd_after_resid(fd_state *s, const double *sums)
{
    if (s->done) return;
    s->its += 1;
    s->scale = 0;
    if (sqrt(sums[0]) <= 1e-9 * s->dmax) { s->done = 1; return; }
    s->m = sums[1] / s->n_d;
}
/* after w -= m; sums = w.x, p.x */
FD_HD void fd_after_shift(fd_state *s, const double *sums)
{
    if (s->done) return;
    s->a = sums[0];
    s->b = s->havep ? sums[1] : 0.0;
}
/* after w -= a x; [p -= b x; Lp -= b Lx]; sums = p.p, w.p */
FD_HD void fd_after_orth_p(fd_state *s, const double *sums)
{
    if (s->done) return;
    s->pn = 1.0; s->a2 = 0.0;
    if (s->havep) {
        s->pn = sqrt(sums[0]);
        if (s->pn > 1e-300) s->a2 = sums[1] / s->pn;
        else s->havep = 0;
    }
}
/* after [p /= pn; Lp /= pn; w -= a2 p]; sums[0] = w.w */
FD_HD void fd_after_orth_w(fd_state *s, const double *sums)
{
    if (s->done) return;
    s->wn = sqrt(sums[0]);
    if (s->wn < 1e-300) s->done = 1;
}
/* after w /= wn; Lw = L w; d6 = x.Lx, x.Lw, x.Lp, w.Lw, w.Lp, p.Lp (the p entries are not read when p does not exist) */
FD_HD void fd_after_rr_dots(fd_state *s, const double *d6)
{
    if (s->done) return;
    double G[3][3], c[3], lam;
    const int m = s->havep ? 3 : 2;
    G[0][0] = d6[0]; G[0][1] = G[1][0] = d6[1]; G[1][1] = d6[3];
    if (m == 3) { G[0][2] = G[2][0] = d6[2]; G[1][2] = G[2][1] = d6[4]; G[2][2] = d6[5]; }
    fd_eig3(m, G, c, &lam);
    if (c[0] < 0) for (int i = 0; i < m; ++i) c[i] = -c[i];
    s->c0 = c[0]; s->c1 = c[1]; s->c2 = s->havep ? c[2] : 0.0;
    s->rho = lam;            /* the Rayleigh quotient of the next iterate is this Ritz value */
}
/* after the Rayleigh-Ritz update (x, Lx, p, Lp); sums[0] = x.x */
FD_HD void fd_after_update(fd_state *s, const double *sums)
{
    if (s->done) return;
    s->xn = sqrt(sums[0]);
    s->havep = 1;
    s->scale = 1;            /* x, Lx are divided by xn at the start of the next iteration, or after the last one */
}

#endif
