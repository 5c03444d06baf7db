/*
 * sp_host.c -- host mirror of the reference's plugin surface (see include/spike_petsc_host.h for the map of
 * names to /root/reference/src file:line).  Plain C, like the reference; the heavy lifting is in libspike_mi355.so.
 */
#include "../../../include/spike_petsc_host.h"
#include "../../../include/spike_mi355.h"

#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

/* engine entry points that are not part of the public PC surface */
int spike_set_operator_csr(spike_handle h, int64_t n, const int64_t *ia, const int64_t *ja, const double *a);
int spike_clear_operator(spike_handle h);
int spike_dev_malloc(void **p, size_t bytes);
int spike_dev_free(void *p);
int spike_dev_upload(void *dst, const void *src, size_t bytes);
int spike_dev_download(void *dst, const void *src, size_t bytes);
int spike_csr_band_k(int64_t n, const int64_t *ia, const int64_t *ja, const double *a, int kmax, double frac, int *k_out,
                     double *frac_out);
int spike_csr_to_band(int64_t n, const int64_t *ia, const int64_t *ja, const double *a, int K, double *band, int64_t ld);

/* ---------------------------------------------------------------------------------------------------- */
static _Thread_local char g_err[512];   /* per thread, like errno: two threads driving two solvers do not overwrite each other's text */
const char *SpikeHostLastError(void) { return g_err; }
static PetscErrorCode seterr(PetscErrorCode code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define CHK(e) do { PetscErrorCode ierr_ = (e); if (ierr_) return ierr_; } while (0)

/* ---- options --------------------------------------------------------------------------------------- */
/* one process-wide table, as PETSc's default options database; every access under one lock, look-ups return a copy */
#include <pthread.h>
#define MAXOPT 256
static struct { char name[128]; char val[128]; } g_opt[MAXOPT];
static int g_nopt = 0;
static pthread_mutex_t g_opt_lock = PTHREAD_MUTEX_INITIALIZER;

PetscErrorCode PetscOptionsSetValue(const char *name, const char *value)
{
    if (!name || name[0] != '-') return seterr(PETSC_ERR_ARG_WRONG, "option names start with '-'");
    PetscErrorCode rc = 0;
    pthread_mutex_lock(&g_opt_lock);
    int i = 0;
    for (; i < g_nopt; ++i)
        if (!strcmp(g_opt[i].name, name)) { snprintf(g_opt[i].val, sizeof g_opt[i].val, "%s", value ? value : ""); break; }
    if (i == g_nopt) {
        if (g_nopt >= MAXOPT) rc = PETSC_ERR_MEM;
        else {
            snprintf(g_opt[g_nopt].name, sizeof g_opt[g_nopt].name, "%s", name);
            snprintf(g_opt[g_nopt].val, sizeof g_opt[g_nopt].val, "%s", value ? value : "");
            ++g_nopt;
        }
    }
    pthread_mutex_unlock(&g_opt_lock);
    return rc ? seterr(rc, "options table full") : 0;
}
PetscErrorCode PetscOptionsClearValue(const char *name)
{
    pthread_mutex_lock(&g_opt_lock);
    for (int i = 0; i < g_nopt; ++i)
        if (!strcmp(g_opt[i].name, name)) { g_opt[i] = g_opt[g_nopt - 1]; --g_nopt; break; }
    pthread_mutex_unlock(&g_opt_lock);
    return 0;
}
PetscErrorCode PetscOptionsClear(void)
{
    pthread_mutex_lock(&g_opt_lock);
    g_nopt = 0;
    pthread_mutex_unlock(&g_opt_lock);
    return 0;
}

/* "-<prefix><key>": the value copied into a per-thread buffer (valid until the thread's next look-up) */
static const char *opt_find(const char *prefix, const char *key)
{
    static _Thread_local char val[128];
    char full[256];
    const char *r = NULL;
    snprintf(full, sizeof full, "-%s%s", prefix ? prefix : "", key);
    pthread_mutex_lock(&g_opt_lock);
    for (int i = 0; i < g_nopt; ++i)
        if (!strcmp(g_opt[i].name, full)) { memcpy(val, g_opt[i].val, sizeof val); r = val; break; }
    pthread_mutex_unlock(&g_opt_lock);
    return r;
}
static void opt_int(const char *prefix, const char *key, PetscInt *v) { const char *s = opt_find(prefix, key); if (s) *v = (PetscInt)atoll(s); }
static void opt_real(const char *prefix, const char *key, PetscReal *v) { const char *s = opt_find(prefix, key); if (s) *v = atof(s); }
static int opt_str(const char *prefix, const char *key, char *out, size_t len) { const char *s = opt_find(prefix, key); if (s) { snprintf(out, len, "%s", s); return 1; } return 0; }

/* ---- objects ----------------------------------------------------------------------------------------- */
struct _p_Mat { PetscInt n; PetscInt *ia, *ja; PetscScalar *a; int refct; };
struct _p_Vec { PetscInt n; PetscScalar *a; };
struct _p_IS { PetscInt n; PetscInt *idx; int refct; };

typedef struct {
    PetscErrorCode (*apply)(PC, Vec, Vec);
    PetscErrorCode (*applytranspose)(PC, Vec, Vec);
    PetscErrorCode (*setup)(PC);
    PetscErrorCode (*reset)(PC);
    PetscErrorCode (*destroy)(PC);
    PetscErrorCode (*setfromoptions)(PC);
    PetscErrorCode (*view)(PC, FILE *);
    PetscErrorCode (*applyrichardson)(PC);
    PetscErrorCode (*applysymmetricleft)(PC, Vec, Vec);
    PetscErrorCode (*applysymmetricright)(PC, Vec, Vec);
    PetscErrorCode (*getspike)(PC, void **);
} PCOps;
struct _p_PC { PCOps ops; void *data; Mat mat, pmat; int setupcalled; char prefix[128]; char type[32]; };

typedef struct {
    PetscErrorCode (*setup)(KSP);
    PetscErrorCode (*solve)(KSP);
    PetscErrorCode (*destroy)(KSP);
    PetscErrorCode (*view)(KSP, FILE *);
    PetscErrorCode (*setfromoptions)(KSP);
} KSPOps;
struct _p_KSP {
    KSPOps ops; void *data; PC pc; Mat A, M; Vec vec_sol, vec_rhs; KSPConvergedReason reason; PetscReal rtol, rnorm;
    PetscInt max_it, its, restart; int setupcalled; char prefix[128]; char type[32];
};

/* ---- Mat ------------------------------------------------------------------------------------------------ */
PetscErrorCode MatCreateSeqAIJWithArrays(PetscInt n, const PetscInt *ia, const PetscInt *ja, const PetscScalar *a, Mat *A)
{
    if (n <= 0 || !ia || !ja || !a || !A) return seterr(PETSC_ERR_ARG_WRONG, "MatCreateSeqAIJWithArrays: bad arguments");
    Mat M = (Mat)calloc(1, sizeof *M);
    const PetscInt nnz = ia[n];
    M->n = n; M->refct = 1;
    M->ia = (PetscInt *)malloc(sizeof(PetscInt) * (size_t)(n + 1));
    M->ja = (PetscInt *)malloc(sizeof(PetscInt) * (size_t)(nnz > 0 ? nnz : 1));
    M->a = (PetscScalar *)malloc(sizeof(PetscScalar) * (size_t)(nnz > 0 ? nnz : 1));
    memcpy(M->ia, ia, sizeof(PetscInt) * (size_t)(n + 1));
    memcpy(M->ja, ja, sizeof(PetscInt) * (size_t)nnz);
    memcpy(M->a, a, sizeof(PetscScalar) * (size_t)nnz);
    for (PetscInt k = 0; k < nnz; ++k)
        if (ja[k] < 0 || ja[k] >= n) { MatDestroy(&M); return seterr(PETSC_ERR_ARG_OUTOFRANGE, "column %lld out of range", (long long)ja[k]); }
    *A = M;
    return 0;
}
static Mat mat_ref(Mat A) { if (A) ++A->refct; return A; }
PetscErrorCode MatDestroy(Mat *A)
{
    if (!A || !*A) return 0;
    if (--(*A)->refct == 0) { free((*A)->ia); free((*A)->ja); free((*A)->a); free(*A); }
    *A = NULL;
    return 0;
}
PetscErrorCode MatGetSize(Mat A, PetscInt *m, PetscInt *n) { if (m) *m = A->n; if (n) *n = A->n; return 0; }
PetscErrorCode MatSeqAIJGetCSR(Mat A, PetscInt *n, const PetscInt **ia, const PetscInt **ja, const PetscScalar **a)
{
    if (n) *n = A->n;
    if (ia) *ia = A->ia;
    if (ja) *ja = A->ja;
    if (a) *a = A->a;
    return 0;
}
PetscErrorCode MatMult(Mat A, Vec x, Vec y)
{
    if (x->n != A->n || y->n != A->n) return seterr(PETSC_ERR_ARG_SIZ, "MatMult: size mismatch");
    for (PetscInt i = 0; i < A->n; ++i) {
        double s = 0;
        for (PetscInt k = A->ia[i]; k < A->ia[i + 1]; ++k) s += A->a[k] * x->a[A->ja[k]];
        y->a[i] = s;
    }
    return 0;
}
PetscErrorCode MatPermute(Mat A, IS rowp, IS colp, Mat *B)
{
    const PetscInt n = A->n;
    if (rowp->n != n || colp->n != n) return seterr(PETSC_ERR_ARG_SIZ, "MatPermute: permutation length");
    /* -mat_permute_device 0|1 (default 1): the gather + per-row sort on the GPU when one is present (libspike_mi355:
       spike_permute_csr; same arrays bit for bit); small matrices and device-less hosts take the loop below */
    {
        int usedev = 1;
        char v[16];
        if (opt_str(NULL, "mat_permute_device", v, sizeof v)) usedev = atoi(v) != 0;
        if (usedev && n >= 4096 && spike_device_count() > 0) {
            const PetscInt nz = A->ia[n];
            PetscInt *ib = (PetscInt *)malloc(sizeof(PetscInt) * (size_t)(n + 1)), *jb = (PetscInt *)malloc(sizeof(PetscInt) * (size_t)(nz > 0 ? nz : 1));
            PetscScalar *bb = (PetscScalar *)malloc(sizeof(PetscScalar) * (size_t)(nz > 0 ? nz : 1));
            const int rc = (ib && jb && bb) ? spike_permute_csr(n, A->ia, A->ja, A->a, rowp->idx, colp->idx, ib, jb, bb) : -7;
            PetscErrorCode e = rc == 0 ? MatCreateSeqAIJWithArrays(n, ib, jb, bb, B) : 0;
            free(ib); free(jb); free(bb);
            if (rc == 0) return e;
            if (rc == -1) return seterr(PETSC_ERR_ARG_WRONG, "MatPermute: row / column IS is not a permutation");
            /* any other failure (no memory, device trouble): the host loop computes the same thing */
        }
    }
    PetscInt *icol = (PetscInt *)malloc(sizeof(PetscInt) * (size_t)n);
    for (PetscInt j = 0; j < n; ++j) icol[j] = -1;
    for (PetscInt j = 0; j < n; ++j) {
        const PetscInt c = colp->idx[j];
        if (c < 0 || c >= n || icol[c] >= 0) { free(icol); return seterr(PETSC_ERR_ARG_WRONG, "MatPermute: column IS is not a permutation"); }
        icol[c] = j;
    }
    const PetscInt nnz = A->ia[n];
    PetscInt *ia = (PetscInt *)malloc(sizeof(PetscInt) * (size_t)(n + 1)), *ja = (PetscInt *)malloc(sizeof(PetscInt) * (size_t)(nnz > 0 ? nnz : 1));
    PetscScalar *a = (PetscScalar *)malloc(sizeof(PetscScalar) * (size_t)(nnz > 0 ? nnz : 1));
    PetscInt q = 0;
    for (PetscInt i = 0; i < n; ++i) {
        const PetscInt r = rowp->idx[i];
        if (r < 0 || r >= n) { free(icol); free(ia); free(ja); free(a); return seterr(PETSC_ERR_ARG_WRONG, "MatPermute: row IS out of range"); }
        ia[i] = q;
        const PetscInt start = q;
        for (PetscInt k = A->ia[r]; k < A->ia[r + 1]; ++k) { ja[q] = icol[A->ja[k]]; a[q] = A->a[k]; ++q; }
        /* keep rows sorted by column (insertion sort: rows are short) */
        for (PetscInt s = start + 1; s < q; ++s) {
            const PetscInt cj = ja[s]; const PetscScalar cv = a[s];
            PetscInt t = s - 1;
            while (t >= start && ja[t] > cj) { ja[t + 1] = ja[t]; a[t + 1] = a[t]; --t; }
            ja[t + 1] = cj; a[t + 1] = cv;
        }
    }
    ia[n] = q;
    PetscErrorCode e = MatCreateSeqAIJWithArrays(n, ia, ja, a, B);
    free(icol); free(ia); free(ja); free(a);
    return e;
}
PetscErrorCode MatComputeBandwidth(Mat A, PetscReal fraction, PetscInt *bw)
{
    (void)fraction;
    PetscInt b = 0;
    for (PetscInt i = 0; i < A->n; ++i)
        for (PetscInt k = A->ia[i]; k < A->ia[i + 1]; ++k) {
            const PetscInt d = A->ja[k] > i ? A->ja[k] - i : i - A->ja[k];
            if (d > b) b = d;
        }
    *bw = b;
    return 0;
}

/* src/matbanded.c:22-107.  The half-bandwidth rule itself is spike_csr_band_k (same pass order as the reference). */
PetscErrorCode MatCreateSubMatrixBanded(Mat A, PetscInt *kmax, PetscReal *frac, Mat *B)
{
    int k = 0;
    double f = 0;
    if (!A || !kmax || !frac || !B) return seterr(PETSC_ERR_ARG_WRONG, "MatCreateSubMatrixBanded: null argument");
    if (spike_csr_band_k(A->n, A->ia, A->ja, A->a, (int)*kmax, *frac, &k, &f)) return seterr(PETSC_ERR_LIB, "band rule failed");
    const PetscInt n = A->n;
    PetscInt nnz = 0;
    for (PetscInt r = 0; r < n; ++r)
        for (PetscInt p = A->ia[r]; p < A->ia[r + 1]; ++p) {
            const PetscInt d = A->ja[p] > r ? A->ja[p] - r : r - A->ja[p];
            if (d <= k) ++nnz;
        }
    PetscInt *ib = (PetscInt *)malloc(sizeof(PetscInt) * (size_t)(n + 1)), *jb = (PetscInt *)malloc(sizeof(PetscInt) * (size_t)(nnz > 0 ? nnz : 1));
    PetscScalar *b = (PetscScalar *)malloc(sizeof(PetscScalar) * (size_t)(nnz > 0 ? nnz : 1));
    PetscInt q = 0;
    for (PetscInt r = 0; r < n; ++r) {
        ib[r] = q;
        for (PetscInt p = A->ia[r]; p < A->ia[r + 1]; ++p) {
            const PetscInt d = A->ja[p] > r ? A->ja[p] - r : r - A->ja[p];
            if (d > k) continue; /* matbanded.c:91 */
            jb[q] = A->ja[p]; b[q] = A->a[p]; ++q;
        }
    }
    ib[n] = q;
    PetscErrorCode e = MatCreateSeqAIJWithArrays(n, ib, jb, b, B);
    free(ib); free(jb); free(b);
    if (e) return e;
    *kmax = k;  /* matbanded.c:104 */
    *frac = f;  /* matbanded.c:105 */
    return 0;
}

/* ---- Vec / IS ---------------------------------------------------------------------------------------------- */
PetscErrorCode VecCreateSeq(PetscInt n, Vec *v)
{
    if (n <= 0 || !v) return seterr(PETSC_ERR_ARG_WRONG, "VecCreateSeq");
    *v = (Vec)calloc(1, sizeof **v);
    (*v)->n = n;
    (*v)->a = (PetscScalar *)calloc((size_t)n, sizeof(PetscScalar));
    return 0;
}
PetscErrorCode VecDestroy(Vec *v) { if (v && *v) { free((*v)->a); free(*v); *v = NULL; } return 0; }
PetscErrorCode VecGetArray(Vec v, PetscScalar **a) { *a = v->a; return 0; }
PetscErrorCode VecGetSize(Vec v, PetscInt *n) { *n = v->n; return 0; }
PetscErrorCode VecSet(Vec v, PetscScalar s) { for (PetscInt i = 0; i < v->n; ++i) v->a[i] = s; return 0; }
PetscErrorCode VecCopy(Vec x, Vec y) { if (x->n != y->n) return seterr(PETSC_ERR_ARG_SIZ, "VecCopy"); memcpy(y->a, x->a, sizeof(PetscScalar) * (size_t)x->n); return 0; }
PetscErrorCode VecAXPY(Vec y, PetscScalar alpha, Vec x) { if (x->n != y->n) return seterr(PETSC_ERR_ARG_SIZ, "VecAXPY"); for (PetscInt i = 0; i < y->n; ++i) y->a[i] += alpha * x->a[i]; return 0; }
PetscErrorCode VecNorm2(Vec v, PetscReal *nrm) { double s = 0; for (PetscInt i = 0; i < v->n; ++i) s += v->a[i] * v->a[i]; *nrm = sqrt(s); return 0; }
PetscErrorCode VecPermute(Vec v, IS is, PetscBool inv)
{
    if (is->n != v->n) return seterr(PETSC_ERR_ARG_SIZ, "VecPermute: IS length");
    PetscScalar *t = (PetscScalar *)malloc(sizeof(PetscScalar) * (size_t)v->n);
    /* -vec_permute_device 1 (default 0): the gather on the GPU (spike_permute_vec).  The mirror's vectors live on the host, so
       this stages them over PCIe and is not faster here; it exists so that the device entry point is exercised by the same
       callers (src/kspreorder.c:122-127) -- a host with device-resident vectors passes device pointers to it directly */
    {
        char o[16];
        if (opt_str(NULL, "vec_permute_device", o, sizeof o) && atoi(o) != 0 && spike_device_count() > 0 &&
            spike_permute_vec(v->n, is->idx, inv ? 1 : 0, v->a, t, 0) == 0) {
            memcpy(v->a, t, sizeof(PetscScalar) * (size_t)v->n);
            free(t);
            return 0;
        }
    }
    if (!inv) for (PetscInt i = 0; i < v->n; ++i) t[i] = v->a[is->idx[i]];
    else for (PetscInt i = 0; i < v->n; ++i) t[is->idx[i]] = v->a[i];
    memcpy(v->a, t, sizeof(PetscScalar) * (size_t)v->n);
    free(t);
    return 0;
}
PetscErrorCode ISCreateGeneral(PetscInt n, const PetscInt *idx, IS *is)
{
    *is = (IS)calloc(1, sizeof **is);
    (*is)->n = n; (*is)->refct = 1;
    (*is)->idx = (PetscInt *)malloc(sizeof(PetscInt) * (size_t)(n > 0 ? n : 1));
    memcpy((*is)->idx, idx, sizeof(PetscInt) * (size_t)n); /* PETSC_COPY_VALUES, petsc_mat_wbm.c:58 */
    return 0;
}
PetscErrorCode ISCreateStride(PetscInt n, PetscInt first, PetscInt step, IS *is)
{
    *is = (IS)calloc(1, sizeof **is);
    (*is)->n = n; (*is)->refct = 1;
    (*is)->idx = (PetscInt *)malloc(sizeof(PetscInt) * (size_t)(n > 0 ? n : 1));
    for (PetscInt i = 0; i < n; ++i) (*is)->idx[i] = first + i * step;
    return 0;
}
PetscErrorCode ISDestroy(IS *is)
{
    if (!is || !*is) return 0;
    if (--(*is)->refct == 0) { free((*is)->idx); free(*is); }
    *is = NULL;
    return 0;
}
PetscErrorCode ISGetIndices(IS is, PetscInt *n, const PetscInt **idx) { if (n) *n = is->n; if (idx) *idx = is->idx; return 0; }

/* ---- orderings ------------------------------------------------------------------------------------------------- */
#define MAXREG 16
static struct { char name[32]; MatOrderingFn fn; } g_ord[MAXREG];
static int g_nord = 0;
PetscErrorCode MatOrderingRegister(const char *name, MatOrderingFn fn)
{
    for (int i = 0; i < g_nord; ++i) if (!strcmp(g_ord[i].name, name)) { g_ord[i].fn = fn; return 0; }
    if (g_nord >= MAXREG) return seterr(PETSC_ERR_MEM, "ordering table full");
    snprintf(g_ord[g_nord].name, sizeof g_ord[g_nord].name, "%s", name);
    g_ord[g_nord++].fn = fn;
    return 0;
}
PetscErrorCode MatGetOrdering(Mat A, MatOrderingType type, IS *row, IS *col)
{
    for (int i = 0; i < g_nord; ++i) if (!strcmp(g_ord[i].name, type)) return g_ord[i].fn(A, type, row, col);
    return seterr(PETSC_ERR_ARG_UNKNOWN_TYPE, "Unknown Mat ordering type %s", type);
}
PetscErrorCode MatGetOrdering_Natural(Mat A, MatOrderingType type, IS *row, IS *col)
{
    (void)type;
    CHK(ISCreateStride(A->n, 0, 1, row));
    return ISCreateStride(A->n, 0, 1, col);
}
/* src/petsc_mat_wbm.c:13-61.  The CSR arrays of A go to the matching's CSC interface (so it matches A^T, :29,52);
 * perm is made 0-based (:55); ROW IS = identity, COLUMN IS = perm (:57-58); the scalings are discarded (:56,59).
 * Option -mat_wbm_rows 1 (not in the reference) applies the same perm as a ROW permutation instead, which is what puts
 * the matched entries of an unsymmetric A on the diagonal. */
PetscErrorCode MatGetOrdering_WBM(Mat A, MatOrderingType type, IS *row, IS *col)
{
    (void)type;
    const PetscInt n = A->n;
    PetscInt *perm = (PetscInt *)malloc(sizeof(PetscInt) * (size_t)n);
    int64_t num = 0;
    if (spike_mc64_job5(n, A->ia, A->ja, A->a, perm, NULL, NULL, &num)) { free(perm); return seterr(PETSC_ERR_LIB, "MC64 job 5 failed"); }
    for (PetscInt i = 0; i < n; ++i)
        if (perm[i] < 0) perm[i] = -perm[i] - 1; /* structurally singular: the completion entries, used as plain indices */
    PetscInt rows = 0;
    opt_int("", "mat_wbm_rows", &rows);
    PetscErrorCode e;
    if (rows) { e = ISCreateGeneral(n, perm, row); if (!e) e = ISCreateStride(n, 0, 1, col); }
    else { e = ISCreateStride(n, 0, 1, row); if (!e) e = ISCreateGeneral(n, perm, col); }
    free(perm);
    return e;
}
/* the per-half scheme of src/spectralPartition.c:326-417 (Fiedler cut, each half reordered on its own, composed); prints
   the reference's two "Reduced ... bandwidth" lines (:377-382) */
PetscErrorCode MatGetOrdering_FiedlerHalves(Mat A, MatOrderingType type, IS *row, IS *col)
{
    (void)type;
    const PetscInt n = A->n;
    PetscInt *ord = (PetscInt *)malloc(sizeof(PetscInt) * (size_t)n);
    int usedev = 1;
    { char v[16]; if (opt_str(NULL, "mat_fiedler_device", v, sizeof v)) usedev = atoi(v) != 0; }
    int64_t np = 0, bw[4] = {0, 0, 0, 0};
    if (spike_fiedler_halves_order(n, A->ia, A->ja, A->a, ord, &np, bw, usedev)) { free(ord); return seterr(PETSC_ERR_LIB, "Fiedler per-half ordering failed"); }
    printf("Reduced positive bandwidth from %lld to %lld\n", (long long)bw[0], (long long)bw[1]);
    printf("Reduced negative bandwidth from %lld to %lld\n", (long long)bw[2], (long long)bw[3]);
    PetscErrorCode e = ISCreateGeneral(n, ord, row);
    free(ord);
    if (e) return e;
    ++(*row)->refct;
    *col = *row;
    return 0;
}
/* PETSc's built-in "rcm", used by the reference as second-stage ordering (src/HOWTO:2, src/testbed.c:236-284) */
PetscErrorCode MatGetOrdering_RCM(Mat A, MatOrderingType type, IS *row, IS *col)
{
    (void)type;
    const PetscInt n = A->n;
    PetscInt *ord = (PetscInt *)malloc(sizeof(PetscInt) * (size_t)n);
    if (spike_rcm_order(n, A->ia, A->ja, ord)) { free(ord); return seterr(PETSC_ERR_LIB, "RCM ordering failed"); }
    PetscErrorCode e = ISCreateGeneral(n, ord, row);
    free(ord);
    if (e) return e;
    ++(*row)->refct;
    *col = *row;
    return 0;
}
/* src/petsc_mat_awbm.c:42-225: row IS = p (p[match[c]] = c), column IS = identity (:200-205) */
PetscErrorCode MatGetOrdering_AWBM(Mat A, MatOrderingType type, IS *row, IS *col)
{
    (void)type;
    const PetscInt n = A->n;
    PetscInt *p = (PetscInt *)malloc(sizeof(PetscInt) * (size_t)n);
    /* -mat_awbm_device 0|1 (default 1): the two greedy phases on the GPU when one is present (libspike_mi355:
       spike_awbm_device, a fixed-point iteration that reproduces the sequential greedy: the SAME matching) */
    int usedev = 1, rc = -1;
    { char v[16]; if (opt_str(NULL, "mat_awbm_device", v, sizeof v)) usedev = atoi(v) != 0; }
    if (usedev && n >= 4096 && spike_device_count() > 0) rc = spike_awbm_device(n, A->ia, A->ja, A->a, p, NULL);
    if (rc) rc = spike_awbm(n, A->ia, A->ja, A->a, p, NULL, NULL);
    if (rc) { free(p); return seterr(PETSC_ERR_LIB, rc == -2 ? "Column unmatched" : "AWBM failed"); }
    PetscErrorCode e = ISCreateGeneral(n, p, row);
    free(p);
    if (e) return e;
    return ISCreateStride(n, 0, 1, col);
}
/* src/petsc_mat_fiedler.c:11-58: one symmetric permutation, returned for rows and columns (:54-56) */
PetscErrorCode MatGetOrdering_Fiedler(Mat A, MatOrderingType type, IS *row, IS *col)
{
    (void)type;
    const PetscInt n = A->n;
    PetscInt *ord = (PetscInt *)malloc(sizeof(PetscInt) * (size_t)n);
    /* -mat_fiedler_device 0|1 (default 1): the LOBPCG refinement on the GPU when one is present; same permutation either way */
    int usedev = 1;
    { char v[16]; if (opt_str(NULL, "mat_fiedler_device", v, sizeof v)) usedev = atoi(v) != 0; }
    if (spike_fiedler_order_ex(n, A->ia, A->ja, A->a, ord, NULL, usedev)) { free(ord); return seterr(PETSC_ERR_LIB, "Fiedler ordering failed"); }
    PetscErrorCode e = ISCreateGeneral(n, ord, row);
    free(ord);
    if (e) return e;
    ++(*row)->refct; /* PetscObjectReference, petsc_mat_fiedler.c:55 */
    *col = *row;
    return 0;
}

/* ---- PC framework -------------------------------------------------------------------------------------------------- */
static struct { char name[32]; PCCreateFn fn; } g_pc[MAXREG];
static int g_npc = 0;
PetscErrorCode PCRegister(const char *name, PCCreateFn fn)
{
    for (int i = 0; i < g_npc; ++i) if (!strcmp(g_pc[i].name, name)) { g_pc[i].fn = fn; return 0; }
    if (g_npc >= MAXREG) return seterr(PETSC_ERR_MEM, "PC table full");
    snprintf(g_pc[g_npc].name, sizeof g_pc[g_npc].name, "%s", name);
    g_pc[g_npc++].fn = fn;
    return 0;
}
PetscErrorCode PCCreate(PC *pc) { *pc = (PC)calloc(1, sizeof **pc); return 0; }
PetscErrorCode PCSetType(PC pc, PCType type)
{
    if (!strcmp(pc->type, type)) return 0;
    for (int i = 0; i < g_npc; ++i)
        if (!strcmp(g_pc[i].name, type)) {
            if (pc->ops.destroy) CHK(pc->ops.destroy(pc));
            memset(&pc->ops, 0, sizeof pc->ops);
            pc->data = NULL; pc->setupcalled = 0;
            snprintf(pc->type, sizeof pc->type, "%s", type);
            return g_pc[i].fn(pc);
        }
    return seterr(PETSC_ERR_ARG_UNKNOWN_TYPE, "Unable to find requested PC type %s", type);
}
PetscErrorCode PCSetOptionsPrefix(PC pc, const char *prefix) { snprintf(pc->prefix, sizeof pc->prefix, "%s", prefix ? prefix : ""); return 0; }
PetscErrorCode PCAppendOptionsPrefix(PC pc, const char *prefix) { strncat(pc->prefix, prefix, sizeof pc->prefix - strlen(pc->prefix) - 1); return 0; }
PetscErrorCode PCSetOperators(PC pc, Mat A, Mat P)
{
    Mat a = mat_ref(A), p = mat_ref(P);
    MatDestroy(&pc->mat); MatDestroy(&pc->pmat);
    pc->mat = a; pc->pmat = p;
    return 0;
}
PetscErrorCode PCSetFromOptions(PC pc)
{
    char t[32];
    if (opt_str(pc->prefix, "pc_type", t, sizeof t)) CHK(PCSetType(pc, t));
    else if (!pc->type[0]) CHK(PCSetType(pc, PCNONE));
    if (pc->ops.setfromoptions) CHK(pc->ops.setfromoptions(pc));
    return 0;
}
PetscErrorCode PCSetUp(PC pc)
{
    if (!pc->type[0]) CHK(PCSetType(pc, PCNONE));
    if (!pc->pmat) return seterr(PETSC_ERR_ARG_WRONGSTATE, "Matrix must be set first");
    if (pc->ops.setup) CHK(pc->ops.setup(pc));
    pc->setupcalled = 1;
    return 0;
}
PetscErrorCode PCApply(PC pc, Vec x, Vec y)
{
    if (x == y) return seterr(PETSC_ERR_ARG_WRONG, "x and y must be different vectors");
    if (!pc->setupcalled) CHK(PCSetUp(pc));
    if (!pc->ops.apply) return seterr(PETSC_ERR_SUP, "PC does not have apply");
    return pc->ops.apply(pc, x, y);
}
PetscErrorCode PCReset(PC pc)
{
    if (pc->ops.reset) CHK(pc->ops.reset(pc));
    MatDestroy(&pc->mat); MatDestroy(&pc->pmat);
    pc->setupcalled = 0;
    return 0;
}
PetscErrorCode PCDestroy(PC *pc)
{
    if (!pc || !*pc) return 0;
    if ((*pc)->ops.destroy) CHK((*pc)->ops.destroy(*pc));
    MatDestroy(&(*pc)->mat); MatDestroy(&(*pc)->pmat);
    free(*pc);
    *pc = NULL;
    return 0;
}
PetscErrorCode PCView(PC pc, FILE *viewer)
{
    fprintf(viewer, "PC Object: type: %s\n", pc->type[0] ? pc->type : "(unset)");
    if (pc->ops.view) CHK(pc->ops.view(pc, viewer));
    return 0;
}
PetscErrorCode PCGetDiagonalScale(PC pc, PetscBool *flag) { (void)pc; *flag = PETSC_FALSE; return 0; }
PetscErrorCode PCGetSpikeHandle(PC pc, void **handle)
{
    *handle = NULL;
    if (pc && pc->ops.getspike) return pc->ops.getspike(pc, handle);
    return 0;
}

/* ---- PCNONE -------------------------------------------------------------------------------------------------------- */
static PetscErrorCode PCApply_None(PC pc, Vec x, Vec y) { (void)pc; return VecCopy(x, y); }
PetscErrorCode PCCreate_None(PC pc) { pc->ops.apply = PCApply_None; return 0; }

/* ---- PCSPIKE: the MI355X engine as a PC ------------------------------------------------------------------------------ */
typedef struct { spike_handle h; PetscInt partitions; char variant[32]; PetscReal boost; PetscInt K; } PC_Spike;
static PetscErrorCode spk(PC pc, int rc)
{
    PC_Spike *s = (PC_Spike *)pc->data;
    if (rc < 0) return seterr(PETSC_ERR_LIB, "libspike_mi355: %s", spike_last_error(s->h));
    return 0;
}
static PetscErrorCode PCSetFromOptions_Spike(PC pc)
{
    PC_Spike *s = (PC_Spike *)pc->data;
    char num[64];
    opt_int(pc->prefix, "pc_spike_partitions", &s->partitions);
    opt_str(pc->prefix, "pc_spike_variant", s->variant, sizeof s->variant);
    opt_real(pc->prefix, "pc_spike_boost", &s->boost);
    snprintf(num, sizeof num, "%lld", (long long)s->partitions);
    CHK(spk(pc, spike_set_option(s->h, "partitions", num)));
    CHK(spk(pc, spike_set_option(s->h, "variant", s->variant)));
    snprintf(num, sizeof num, "%.17g", s->boost);
    return spk(pc, spike_set_option(s->h, "boost", num));
}
static PetscErrorCode PCSetUp_Spike(PC pc)
{
    PC_Spike *s = (PC_Spike *)pc->data;
    Mat P = pc->pmat;
    PetscInt bw = 0;
    CHK(MatComputeBandwidth(P, 0.0, &bw));
    if (bw > 256) return seterr(PETSC_ERR_SUP, "PCSPIKE: half-bandwidth %lld > 256; extract a band first (PCBANDED)", (long long)bw);
    double *band = (double *)malloc(sizeof(double) * (size_t)(2 * bw + 1) * (size_t)P->n);
    if (!band) return seterr(PETSC_ERR_MEM, "PCSPIKE: band allocation");
    spike_csr_to_band(P->n, P->ia, P->ja, P->a, (int)bw, band, P->n);
    const int rc = spike_setup_band(s->h, P->n, 0, P->n, (int)bw, band, P->n, 0);
    free(band);
    s->K = bw;
    return spk(pc, rc);
}
static PetscErrorCode PCApply_Spike(PC pc, Vec x, Vec y)
{
    PC_Spike *s = (PC_Spike *)pc->data;
    return spk(pc, spike_apply(s->h, x->a, y->a, 0));
}
static PetscErrorCode PCReset_Spike(PC pc) { PC_Spike *s = (PC_Spike *)pc->data; return spk(pc, spike_reset(s->h)); }
static PetscErrorCode PCDestroy_Spike(PC pc)
{
    PC_Spike *s = (PC_Spike *)pc->data;
    if (s) { spike_destroy(s->h); free(s); pc->data = NULL; }
    return 0;
}
static PetscErrorCode PCView_Spike(PC pc, FILE *viewer)
{
    PC_Spike *s = (PC_Spike *)pc->data;
    char buf[1024];
    if (!spike_view(s->h, buf, sizeof buf)) fputs(buf, viewer);
    return 0;
}
static PetscErrorCode PCGetSpike_Spike(PC pc, void **h) { *h = ((PC_Spike *)pc->data)->h; return 0; }
PetscErrorCode PCCreate_Spike(PC pc)
{
    PC_Spike *s = (PC_Spike *)calloc(1, sizeof *s);
    pc->data = s;
    s->partitions = 0; s->boost = 1e-10;
    snprintf(s->variant, sizeof s->variant, "coupled");
    if (spike_create(&s->h)) { free(s); pc->data = NULL; return seterr(PETSC_ERR_LIB, "spike_create failed: no HIP device"); }
    pc->ops.apply = PCApply_Spike; pc->ops.setup = PCSetUp_Spike; pc->ops.reset = PCReset_Spike;
    pc->ops.destroy = PCDestroy_Spike; pc->ops.setfromoptions = PCSetFromOptions_Spike; pc->ops.view = PCView_Spike;
    pc->ops.getspike = PCGetSpike_Spike;
    return 0;
}

/* ---- PCBANDED, src/matbanded.c:109-343 ------------------------------------------------------------------------------- */
typedef struct {
    PetscInt kmax, k;  /* matbanded.c:112 */
    PetscReal frac, f; /* :113 */
    Mat B;             /* :114 */
    PC pc;             /* :115 the embedded PC */
} PC_Banded;

static PetscErrorCode PCReset_Banded(PC pc) /* :120-129 */
{
    PC_Banded *b = (PC_Banded *)pc->data;
    CHK(MatDestroy(&b->B));
    return PCReset(b->pc);
}
static PetscErrorCode PCDestroy_Banded(PC pc) /* :133-145 */
{
    PC_Banded *b = (PC_Banded *)pc->data;
    if (!b) return 0;
    CHK(PCReset_Banded(pc));
    CHK(PCDestroy(&b->pc));
    free(b);
    pc->data = NULL;
    return 0;
}
static PetscErrorCode PCSetFromOptions_Banded(PC pc) /* :149-161 */
{
    PC_Banded *b = (PC_Banded *)pc->data;
    opt_int(pc->prefix, "pc_banded_kmax", &b->kmax);
    opt_real(pc->prefix, "pc_banded_frac", &b->frac);
    if (!opt_find(b->pc->prefix, "pc_type") && !b->pc->type[0]) CHK(PCSetType(b->pc, PCSPIKE));
    return PCSetFromOptions(b->pc);
}
static PetscErrorCode PCSetUp_Banded(PC pc) /* :165-180 */
{
    PC_Banded *b = (PC_Banded *)pc->data;
    if (pc->setupcalled == 0) {
        b->k = b->kmax;
        b->f = b->frac;
        CHK(MatDestroy(&b->B));
        CHK(MatCreateSubMatrixBanded(pc->pmat, &b->k, &b->f, &b->B));
        if (!b->pc->type[0]) CHK(PCSetType(b->pc, PCSPIKE));
        CHK(PCSetOperators(b->pc, pc->mat, b->B));
    }
    return PCSetUp(b->pc);
}
static PetscErrorCode PCApply_Banded(PC pc, Vec x, Vec y) /* :184-192 */
{
    PC_Banded *b = (PC_Banded *)pc->data;
    return PCApply(b->pc, x, y);
}
static PetscErrorCode PCView_Banded(PC pc, FILE *viewer) /* :196-211 */
{
    PC_Banded *b = (PC_Banded *)pc->data;
    fprintf(viewer, "  Banded: k = %d (%d max), frac = %g (%g max)\n", (int)b->k, (int)b->kmax, b->f, b->frac);
    return PCView(b->pc, viewer);
}
static PetscErrorCode PCGetSpike_Banded(PC pc, void **h) { return PCGetSpikeHandle(((PC_Banded *)pc->data)->pc, h); }
PetscErrorCode PCCreate_Banded(PC pc) /* :251-283 */
{
    PC_Banded *b = (PC_Banded *)calloc(1, sizeof *b);
    pc->data = b;
    b->kmax = 50;   /* :261 */
    b->frac = 0.95; /* :262 */
    pc->ops.apply = PCApply_Banded;
    pc->ops.applytranspose = NULL;
    pc->ops.setup = PCSetUp_Banded;
    pc->ops.reset = PCReset_Banded;
    pc->ops.destroy = PCDestroy_Banded;
    pc->ops.setfromoptions = PCSetFromOptions_Banded;
    pc->ops.view = PCView_Banded;
    pc->ops.applyrichardson = NULL;
    pc->ops.applysymmetricleft = NULL;
    pc->ops.applysymmetricright = NULL;
    pc->ops.getspike = PCGetSpike_Banded;
    CHK(PCCreate(&b->pc));
    CHK(PCSetOptionsPrefix(b->pc, pc->prefix));
    return PCAppendOptionsPrefix(b->pc, "banded_"); /* :281 */
}
PetscErrorCode PCBandedSetMaxHalfBandwidth(PC pc, PetscInt kmax)
{
    if (strcmp(pc->type, PCBANDED)) return 0; /* PetscTryMethod: silently ignored for other types */
    ((PC_Banded *)pc->data)->kmax = kmax;
    return 0;
}
PetscErrorCode PCBandedSetNormFraction(PC pc, PetscReal frac)
{
    if (strcmp(pc->type, PCBANDED)) return 0;
    ((PC_Banded *)pc->data)->frac = frac;
    return 0;
}
PetscErrorCode PCBandedGetInfo(PC pc, PetscInt *k, PetscReal *f, PetscInt *kmax, PetscReal *frac)
{
    if (strcmp(pc->type, PCBANDED)) return seterr(PETSC_ERR_ARG_WRONG, "not a banded PC");
    PC_Banded *b = (PC_Banded *)pc->data;
    if (k) *k = b->k;
    if (f) *f = b->f;
    if (kmax) *kmax = b->kmax;
    if (frac) *frac = b->frac;
    return 0;
}

/* ---- KSP framework ---------------------------------------------------------------------------------------------------- */
static struct { char name[32]; KSPCreateFn fn; } g_ksp[MAXREG];
static int g_nksp = 0;
PetscErrorCode KSPRegister(const char *name, KSPCreateFn fn)
{
    for (int i = 0; i < g_nksp; ++i) if (!strcmp(g_ksp[i].name, name)) { g_ksp[i].fn = fn; return 0; }
    if (g_nksp >= MAXREG) return seterr(PETSC_ERR_MEM, "KSP table full");
    snprintf(g_ksp[g_nksp].name, sizeof g_ksp[g_nksp].name, "%s", name);
    g_ksp[g_nksp++].fn = fn;
    return 0;
}
PetscErrorCode KSPCreate(KSP *ksp)
{
    *ksp = (KSP)calloc(1, sizeof **ksp);
    (*ksp)->rtol = 1e-5; (*ksp)->max_it = 10000; (*ksp)->restart = 30;
    return PCCreate(&(*ksp)->pc);
}
PetscErrorCode KSPSetType(KSP ksp, KSPType type)
{
    if (!strcmp(ksp->type, type)) return 0;
    for (int i = 0; i < g_nksp; ++i)
        if (!strcmp(g_ksp[i].name, type)) {
            if (ksp->ops.destroy) CHK(ksp->ops.destroy(ksp));
            memset(&ksp->ops, 0, sizeof ksp->ops);
            ksp->data = NULL; ksp->setupcalled = 0;
            snprintf(ksp->type, sizeof ksp->type, "%s", type);
            return g_ksp[i].fn(ksp);
        }
    return seterr(PETSC_ERR_ARG_UNKNOWN_TYPE, "Unable to find requested KSP type %s", type);
}
PetscErrorCode KSPSetOptionsPrefix(KSP ksp, const char *prefix)
{
    snprintf(ksp->prefix, sizeof ksp->prefix, "%s", prefix ? prefix : "");
    return PCSetOptionsPrefix(ksp->pc, ksp->prefix);
}
PetscErrorCode KSPAppendOptionsPrefix(KSP ksp, const char *prefix)
{
    strncat(ksp->prefix, prefix, sizeof ksp->prefix - strlen(ksp->prefix) - 1);
    return PCSetOptionsPrefix(ksp->pc, ksp->prefix);
}
PetscErrorCode KSPSetOperators(KSP ksp, Mat A, Mat M)
{
    Mat a = mat_ref(A), m = mat_ref(M);
    MatDestroy(&ksp->A); MatDestroy(&ksp->M);
    ksp->A = a; ksp->M = m;
    ksp->setupcalled = 0;
    return 0;
}
PetscErrorCode KSPGetPC(KSP ksp, PC *pc) { *pc = ksp->pc; return 0; }
PetscErrorCode KSPSetTolerances(KSP ksp, PetscReal rtol, PetscInt maxits) { ksp->rtol = rtol; ksp->max_it = maxits; return 0; }
PetscErrorCode KSPSetFromOptions(KSP ksp)
{
    char t[32];
    if (opt_str(ksp->prefix, "ksp_type", t, sizeof t)) CHK(KSPSetType(ksp, t));
    else if (!ksp->type[0]) CHK(KSPSetType(ksp, KSPGMRES));
    opt_real(ksp->prefix, "ksp_rtol", &ksp->rtol);
    opt_int(ksp->prefix, "ksp_max_it", &ksp->max_it);
    opt_int(ksp->prefix, "ksp_gmres_restart", &ksp->restart);
    if (ksp->ops.setfromoptions) CHK(ksp->ops.setfromoptions(ksp));
    return PCSetFromOptions(ksp->pc);
}
PetscErrorCode KSPSetUp(KSP ksp)
{
    if (!ksp->type[0]) CHK(KSPSetType(ksp, KSPGMRES));
    if (!ksp->A) return seterr(PETSC_ERR_ARG_WRONGSTATE, "KSPSetOperators must be called first");
    if (ksp->setupcalled) return 0;
    if (ksp->ops.setup) CHK(ksp->ops.setup(ksp));
    ksp->setupcalled = 1;
    return 0;
}
PetscErrorCode KSPSolve(KSP ksp, Vec b, Vec x)
{
    if (b == x) return seterr(PETSC_ERR_ARG_WRONG, "b and x must be different vectors");
    CHK(KSPSetUp(ksp));
    ksp->vec_rhs = b; ksp->vec_sol = x;
    ksp->reason = KSP_CONVERGED_ITERATING;
    if (!ksp->ops.solve) return seterr(PETSC_ERR_SUP, "KSP has no solve");
    return ksp->ops.solve(ksp);
}
PetscErrorCode KSPGetConvergedReason(KSP ksp, KSPConvergedReason *reason) { *reason = ksp->reason; return 0; }
PetscErrorCode KSPGetIterationNumber(KSP ksp, PetscInt *its) { *its = ksp->its; return 0; }
PetscErrorCode KSPGetResidualNorm(KSP ksp, PetscReal *rnorm) { *rnorm = ksp->rnorm; return 0; }
PetscErrorCode KSPView(KSP ksp, FILE *viewer)
{
    fprintf(viewer, "KSP Object: type: %s, rtol = %g, max_it = %lld\n", ksp->type, ksp->rtol, (long long)ksp->max_it);
    if (ksp->ops.view) CHK(ksp->ops.view(ksp, viewer));
    return PCView(ksp->pc, viewer);
}
PetscErrorCode KSPDestroy(KSP *ksp)
{
    if (!ksp || !*ksp) return 0;
    if ((*ksp)->ops.destroy) CHK((*ksp)->ops.destroy(*ksp));
    CHK(PCDestroy(&(*ksp)->pc));
    MatDestroy(&(*ksp)->A); MatDestroy(&(*ksp)->M);
    free(*ksp);
    *ksp = NULL;
    return 0;
}

/* ---- KSPGMRES: left-preconditioned GMRES(m) on the device (spike_gmres), options of src/makefile:18 ------------------- */
typedef struct { spike_handle own; } KSP_GMRES;
static PetscErrorCode KSPSetUp_GMRES(KSP ksp)
{
    CHK(PCSetOperators(ksp->pc, ksp->A, ksp->M));
    return PCSetUp(ksp->pc);
}
static PetscErrorCode KSPSolve_GMRES(KSP ksp)
{
    KSP_GMRES *g = (KSP_GMRES *)ksp->data;
    void *hv = NULL;
    CHK(PCGetSpikeHandle(ksp->pc, &hv));
    spike_handle h = (spike_handle)hv;
    int use_pc = 1;
    if (!h) {
        if (strcmp(ksp->pc->type, PCNONE)) return seterr(PETSC_ERR_SUP, "KSPGMRES runs on the device: PC type %s has no device apply", ksp->pc->type);
        if (!g->own && spike_create(&g->own)) return seterr(PETSC_ERR_LIB, "spike_create failed: no HIP device");
        h = g->own;
        use_pc = 0;
    }
    Mat A = ksp->A;
    const PetscInt n = A->n;
    if (ksp->vec_rhs->n != n || ksp->vec_sol->n != n) return seterr(PETSC_ERR_ARG_SIZ, "KSPSolve: vector size");
    if (spike_set_operator_csr(h, n, A->ia, A->ja, A->a) < 0) return seterr(PETSC_ERR_LIB, "libspike_mi355: %s", spike_last_error(h));
    void *db = NULL, *dx = NULL;
    if (spike_dev_malloc(&db, sizeof(double) * (size_t)n) || spike_dev_malloc(&dx, sizeof(double) * (size_t)n)) {
        if (db) spike_dev_free(db);
        spike_clear_operator(h);
        return seterr(PETSC_ERR_MEM, "device allocation");
    }
    spike_dev_upload(db, ksp->vec_rhs->a, sizeof(double) * (size_t)n);
    spike_dev_upload(dx, ksp->vec_sol->a, sizeof(double) * (size_t)n);
    {   /* PETSc's -ksp_gmres_cgs_refinement_type refine_never|refine_ifneeded|refine_always (default refine_never) */
        char rt[32];
        if (opt_str(ksp->prefix, "ksp_gmres_cgs_refinement_type", rt, sizeof rt) && spike_set_option(h, "gmres_cgs_refinement_type", rt)) {
            spike_dev_free(db); spike_dev_free(dx);   /* every exit releases the device vectors and the CSR operator */
            spike_clear_operator(h);
            return seterr(PETSC_ERR_ARG_OUTOFRANGE, "libspike_mi355: %s", spike_last_error(h));
        }
    }
    int its = 0;
    double rn = 0, ms = 0;
    const int rc = spike_gmres(h, (const double *)db, (double *)dx, (int)ksp->restart, ksp->rtol, (int)ksp->max_it, use_pc, &its, &rn, &ms);
    spike_dev_download(ksp->vec_sol->a, dx, sizeof(double) * (size_t)n);
    spike_dev_free(db); spike_dev_free(dx);
    spike_clear_operator(h);
    if (rc < 0) return seterr(PETSC_ERR_LIB, "libspike_mi355: %s", spike_last_error(h));
    ksp->its = its; ksp->rnorm = rn;
    ksp->reason = rc == 0 ? KSP_CONVERGED_RTOL : KSP_DIVERGED_ITS;
    return 0;
}
static PetscErrorCode KSPDestroy_GMRES(KSP ksp)
{
    KSP_GMRES *g = (KSP_GMRES *)ksp->data;
    if (g) { if (g->own) spike_destroy(g->own); free(g); ksp->data = NULL; }
    return 0;
}
PetscErrorCode KSPCreate_GMRES(KSP ksp)
{
    ksp->data = calloc(1, sizeof(KSP_GMRES));
    ksp->ops.setup = KSPSetUp_GMRES; ksp->ops.solve = KSPSolve_GMRES; ksp->ops.destroy = KSPDestroy_GMRES;
    return 0;
}

/* ---- KSPREORDER, src/kspreorder.c ---------------------------------------------------------------------------------------- */
typedef struct {
    KSP ksp;             /* kspreorder.c:4 the embedded KSP */
    char ordertype[256]; /* :5 */
    IS rorder, corder;   /* :6 */
} KSP_Reorder;

static PetscErrorCode KSPSetUp_Reorder(KSP ksp) /* :11-28 */
{
    KSP_Reorder *r = (KSP_Reorder *)ksp->data;
    Mat A = ksp->A, M = ksp->M, PA = NULL, PM = NULL;
    CHK(ISDestroy(&r->rorder)); CHK(ISDestroy(&r->corder));
    CHK(MatGetOrdering(M, r->ordertype, &r->rorder, &r->corder));
    CHK(MatPermute(M, r->rorder, r->corder, &PM));
    if (A != M) CHK(MatPermute(A, r->rorder, r->corder, &PA));
    else PA = PM;
    CHK(KSPSetOperators(r->ksp, PA, PM));
    CHK(KSPSetUp(r->ksp));
    CHK(MatDestroy(&PM));
    if (A != M) CHK(MatDestroy(&PA));
    return 0;
}
static PetscErrorCode KSPSolve_Reorder(KSP ksp) /* :113-128, the live #else branch */
{
    KSP_Reorder *r = (KSP_Reorder *)ksp->data;
    Vec x = ksp->vec_sol, b = ksp->vec_rhs;
    PetscBool diagonalscale;
    CHK(PCGetDiagonalScale(ksp->pc, &diagonalscale));
    if (diagonalscale) return seterr(PETSC_ERR_SUP, "Krylov method %s does not support diagonal scaling", ksp->type);
    CHK(VecPermute(x, r->corder, PETSC_FALSE));
    CHK(VecPermute(b, r->rorder, PETSC_FALSE));
    PetscErrorCode e = KSPSolve(r->ksp, b, x);
    if (!e) e = KSPGetConvergedReason(r->ksp, &ksp->reason);
    ksp->its = r->ksp->its; ksp->rnorm = r->ksp->rnorm;
    PetscErrorCode e2 = VecPermute(x, r->corder, PETSC_TRUE);
    PetscErrorCode e3 = VecPermute(b, r->rorder, PETSC_TRUE);
    return e ? e : (e2 ? e2 : e3);
}
static PetscErrorCode KSPSetFromOptions_Reorder(KSP ksp) /* :134-151 */
{
    KSP_Reorder *r = (KSP_Reorder *)ksp->data;
    char tname[256];
    snprintf(r->ordertype, sizeof r->ordertype, "%s", MATORDERINGNATURAL); /* reset on every call, :144 */
    if (opt_str(ksp->prefix, "mat_ordering_type", tname, sizeof tname)) snprintf(r->ordertype, sizeof r->ordertype, "%s", tname);
    return KSPSetFromOptions(r->ksp);
}
static PetscErrorCode KSPView_Reorder(KSP ksp, FILE *viewer) /* :155-170 */
{
    KSP_Reorder *r = (KSP_Reorder *)ksp->data;
    fprintf(viewer, "  reordering type = %s\n", r->ordertype);
    return KSPView(r->ksp, viewer);
}
static PetscErrorCode KSPDestroy_Reorder(KSP ksp) /* :174-185 */
{
    KSP_Reorder *r = (KSP_Reorder *)ksp->data;
    if (!r) return 0;
    CHK(ISDestroy(&r->rorder));
    CHK(ISDestroy(&r->corder));
    CHK(KSPDestroy(&r->ksp));
    free(r);
    ksp->data = NULL;
    return 0;
}
PetscErrorCode KSPCreate_Reorder(KSP ksp) /* :197-223 */
{
    KSP_Reorder *r = (KSP_Reorder *)calloc(1, sizeof *r);
    ksp->data = r;
    snprintf(r->ordertype, sizeof r->ordertype, "%s", MATORDERINGNATURAL);
    ksp->ops.setup = KSPSetUp_Reorder;
    ksp->ops.solve = KSPSolve_Reorder;
    ksp->ops.destroy = KSPDestroy_Reorder;
    ksp->ops.view = KSPView_Reorder;
    ksp->ops.setfromoptions = KSPSetFromOptions_Reorder;
    CHK(KSPCreate(&r->ksp));
    CHK(KSPSetOptionsPrefix(r->ksp, ksp->prefix));
    return KSPAppendOptionsPrefix(r->ksp, "reorder_"); /* :221 */
}
PetscErrorCode KSPReorderGetOrdering(KSP ksp, IS *row, IS *col)
{
    if (strcmp(ksp->type, KSPREORDER)) return seterr(PETSC_ERR_ARG_WRONG, "not a reorder KSP");
    KSP_Reorder *r = (KSP_Reorder *)ksp->data;
    if (row) *row = r->rorder;
    if (col) *col = r->corder;
    return 0;
}

/* ---- registration, src/testbed2.c:61-73 ---------------------------------------------------------------------------------- */
PetscErrorCode SpikePetscRegisterAll(void)
{
    CHK(MatOrderingRegister("natural", MatGetOrdering_Natural));
    CHK(MatOrderingRegister("rcm", MatGetOrdering_RCM));
    CHK(MatOrderingRegister("wbm", MatGetOrdering_WBM));         /* testbed2.c:66 */
    CHK(MatOrderingRegister("awbm", MatGetOrdering_AWBM));       /* :67 */
    CHK(MatOrderingRegister("fiedler", MatGetOrdering_Fiedler)); /* :68 */
    CHK(MatOrderingRegister("fiedler_halves", MatGetOrdering_FiedlerHalves)); /* spectralPartition.c:369-417 as an ordering */
    CHK(PCRegister(PCNONE, PCCreate_None));
    CHK(PCRegister(PCSPIKE, PCCreate_Spike));
    CHK(PCRegister(PCBANDED, PCCreate_Banded)); /* :70 */
    CHK(KSPRegister(KSPGMRES, KSPCreate_GMRES));
    CHK(KSPRegister(KSPREORDER, KSPCreate_Reorder)); /* :71 */
    return 0;
}
