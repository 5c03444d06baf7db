/*
 * idx32.c -- the ordering kernels for 32-bit index arrays.
 *
 * PETSc's default build has a 32-bit PetscInt; the reference hands PetscInt arrays straight to its kernels
 * (HSLmc64AD, /root/reference/src/petsc_mat_wbm.c:52; hslmc73_, src/petsc_mat_fiedler.c:45; the AWBM loops over
 * aij->i / aij->j, src/petsc_mat_awbm.c:70-95).  The kernels of this library work on int64 arrays; these entry points
 * widen the inputs, call them, and narrow the permutations, so that the PETSc glue (examples/petsc/kspreorder_spike.c)
 * can select on sizeof(PetscInt) without touching the heap past a 4-byte array.  Results are the 64-bit kernels' results.
 */
#include <stdint.h>
#include <stdlib.h>

#include "../../../include/spike_petsc_host.h"

typedef struct { int64_t *ia, *ja; } wide_t;

static int widen(int64_t n, const int32_t *ia, const int32_t *ja, wide_t *w)
{
    w->ia = NULL; w->ja = NULL;
    if (n <= 0 || !ia || !ja || ia[n] < 0) return -1;
    const int64_t nnz = ia[n];
    w->ia = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    w->ja = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz > 0 ? nnz : 1));
    if (!w->ia || !w->ja) { free(w->ia); free(w->ja); return -1; }
    for (int64_t i = 0; i <= n; ++i) w->ia[i] = ia[i];
    for (int64_t k = 0; k < nnz; ++k) w->ja[k] = ja[k];
    return 0;
}

static int narrow(int64_t n, const int64_t *src, int32_t *dst)
{
    for (int64_t i = 0; i < n; ++i) dst[i] = (int32_t)src[i];   /* |values| <= n < 2^31: they are row/column numbers */
    return 0;
}

int spike_mc64_job5_i32(int32_t n, const int32_t *colptr, const int32_t *rowind, const double *val, int32_t *perm, double *u,
                        double *v, int32_t *num)
{
    wide_t w;
    if (!perm || widen(n, colptr, rowind, &w)) return -1;
    int64_t *p64 = (int64_t *)malloc(sizeof(int64_t) * (size_t)n), num64 = 0;
    int rc = p64 ? spike_mc64_job5(n, w.ia, w.ja, val, p64, u, v, &num64) : -1;
    if (!rc) { narrow(n, p64, perm); if (num) *num = (int32_t)num64; }
    free(p64); free(w.ia); free(w.ja);
    return rc;
}

int spike_awbm_i32(int32_t n, const int32_t *ia, const int32_t *ja, const double *a, int32_t *perm, double *sr, double *sc)
{
    wide_t w;
    if (!perm || widen(n, ia, ja, &w)) return -1;
    int64_t *p64 = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    int rc = p64 ? spike_awbm(n, w.ia, w.ja, a, p64, sr, sc) : -1;
    if (!rc) narrow(n, p64, perm);
    free(p64); free(w.ia); free(w.ja);
    return rc;
}

int spike_fiedler_order_i32(int32_t n, const int32_t *ia, const int32_t *ja, const double *a, int32_t *order, double *vec,
                            int use_device)
{
    wide_t w;
    if (!order || widen(n, ia, ja, &w)) return -1;
    int64_t *o64 = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    int rc = o64 ? spike_fiedler_order_ex(n, w.ia, w.ja, a, o64, vec, use_device) : -1;
    if (!rc) narrow(n, o64, order);
    free(o64); free(w.ia); free(w.ja);
    return rc;
}

int spike_rcm_order_i32(int32_t n, const int32_t *ia, const int32_t *ja, int32_t *order)
{
    wide_t w;
    if (!order || widen(n, ia, ja, &w)) return -1;
    int64_t *o64 = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    int rc = o64 ? spike_rcm_order(n, w.ia, w.ja, o64) : -1;
    if (!rc) narrow(n, o64, order);
    free(o64); free(w.ia); free(w.ja);
    return rc;
}
