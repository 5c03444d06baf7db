/*
 * matio.c -- the two file formats the reference's drivers use.
 *
 *   MatLoad              PETSc binary AIJ, what /root/reference/src/testbed2.c:93-96 reads
 *                        (PetscViewerBinaryOpen(FILE_MODE_READ) + MatLoad); big-endian:
 *                        int32 classid 1211216, int32 M, int32 N, int32 nz, int32 rowlen[M], int32 col[nz], float64 val[nz]
 *   MatViewBinary        the same format, written
 *   MatLoadMatrixMarket / MatViewMatrixMarket
 *                        coordinate real/integer/pattern, general/symmetric/skew-symmetric; the reference exports
 *                        MatrixMarket at /root/reference/src/wbm.c:520-522 and reads it at :476-477
 * None of the matrices named by the reference's docs ships with it; these readers make them drop-in when
 * present.  Square matrices only (the whole plugin surface is square, petsc_mat_wbm.c:30).
 */
#include "../../../include/spike_petsc_host.h"

#include <ctype.h>
#include <stdlib.h>
#include <string.h>

#define MAT_FILE_CLASSID 1211216

static int rd_be32(FILE *f, int32_t *v)
{
    unsigned char b[4];
    if (fread(b, 1, 4, f) != 4) return -1;
    *v = (int32_t)(((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | (uint32_t)b[3]);
    return 0;
}
static int rd_be64f(FILE *f, double *v)
{
    unsigned char b[8];
    if (fread(b, 1, 8, f) != 8) return -1;
    uint64_t u = 0;
    for (int i = 0; i < 8; ++i) u = (u << 8) | b[i];
    memcpy(v, &u, 8);
    return 0;
}
static void wr_be32(FILE *f, int32_t v)
{
    unsigned char b[4] = {(unsigned char)((uint32_t)v >> 24), (unsigned char)((uint32_t)v >> 16), (unsigned char)((uint32_t)v >> 8), (unsigned char)v};
    fwrite(b, 1, 4, f);
}
static void wr_be64f(FILE *f, double d)
{
    uint64_t u;
    memcpy(&u, &d, 8);
    unsigned char b[8];
    for (int i = 7; i >= 0; --i) { b[i] = (unsigned char)(u & 0xff); u >>= 8; }
    fwrite(b, 1, 8, f);
}

PetscErrorCode MatLoad(const char *path, Mat *A)
{
    FILE *f = fopen(path, "rb");
    if (!f) return PETSC_ERR_ARG_WRONG;
    int32_t cid = 0, M = 0, N = 0, nz = 0;
    if (rd_be32(f, &cid) || rd_be32(f, &M) || rd_be32(f, &N) || rd_be32(f, &nz) || cid != MAT_FILE_CLASSID || M <= 0 || M != N || nz < 0) {
        fclose(f);
        return PETSC_ERR_ARG_WRONG;
    }
    PetscInt *ia = (PetscInt *)malloc(sizeof(PetscInt) * (size_t)(M + 1)), *ja = (PetscInt *)malloc(sizeof(PetscInt) * (size_t)(nz > 0 ? nz : 1));
    PetscScalar *a = (PetscScalar *)malloc(sizeof(PetscScalar) * (size_t)(nz > 0 ? nz : 1));
    if (!ia || !ja || !a) {   /* the sizes come from the file header: a damaged header must not become a null dereference */
        free(ia); free(ja); free(a); fclose(f);
        return PETSC_ERR_MEM;
    }
    int bad = 0;
    ia[0] = 0;
    for (int32_t i = 0; i < M && !bad; ++i) { int32_t l = 0; bad = rd_be32(f, &l) || l < 0; ia[i + 1] = ia[i] + (bad ? 0 : l); }
    if (!bad && ia[M] != nz) bad = 1;
    for (int32_t k = 0; k < nz && !bad; ++k) { int32_t c; bad = rd_be32(f, &c) || c < 0 || c >= N; ja[k] = c; }
    for (int32_t k = 0; k < nz && !bad; ++k) bad = rd_be64f(f, &a[k]);
    fclose(f);
    PetscErrorCode e = bad ? PETSC_ERR_ARG_WRONG : MatCreateSeqAIJWithArrays(M, ia, ja, a, A);
    free(ia); free(ja); free(a);
    return e;
}

PetscErrorCode MatViewBinary(Mat A, const char *path)
{
    PetscInt n;
    const PetscInt *ia, *ja;
    const PetscScalar *a;
    MatSeqAIJGetCSR(A, &n, &ia, &ja, &a);
    if (ia[n] > 2147483647LL) return PETSC_ERR_ARG_OUTOFRANGE;
    FILE *f = fopen(path, "wb");
    if (!f) return PETSC_ERR_ARG_WRONG;
    wr_be32(f, MAT_FILE_CLASSID); wr_be32(f, (int32_t)n); wr_be32(f, (int32_t)n); wr_be32(f, (int32_t)ia[n]);
    for (PetscInt i = 0; i < n; ++i) wr_be32(f, (int32_t)(ia[i + 1] - ia[i]));
    for (PetscInt k = 0; k < ia[n]; ++k) wr_be32(f, (int32_t)ja[k]);
    for (PetscInt k = 0; k < ia[n]; ++k) wr_be64f(f, a[k]);
    fclose(f);
    return 0;
}

typedef struct { PetscInt r, c; double v; } trip_t;
static int trip_cmp(const void *x, const void *y)
{
    const trip_t *a = (const trip_t *)x, *b = (const trip_t *)y;
    if (a->r != b->r) return a->r < b->r ? -1 : 1;
    if (a->c != b->c) return a->c < b->c ? -1 : 1;
    return 0;
}

PetscErrorCode MatLoadMatrixMarket(const char *path, Mat *A)
{
    FILE *f = fopen(path, "r");
    if (!f) return PETSC_ERR_ARG_WRONG;
    char line[1024];
    if (!fgets(line, sizeof line, f)) { fclose(f); return PETSC_ERR_ARG_WRONG; }
    for (char *p = line; *p; ++p) *p = (char)tolower((unsigned char)*p);
    if (!strstr(line, "%%matrixmarket") || !strstr(line, "matrix") || !strstr(line, "coordinate") || strstr(line, "complex")) { fclose(f); return PETSC_ERR_SUP; }
    const int pattern = strstr(line, "pattern") != NULL;
    const int skew = strstr(line, "skew-symmetric") != NULL;
    const int symm = !skew && strstr(line, "symmetric") != NULL;
    do { if (!fgets(line, sizeof line, f)) { fclose(f); return PETSC_ERR_ARG_WRONG; } } while (line[0] == '%' || line[0] == '\n');
    long long M, N, nz;
    if (sscanf(line, "%lld %lld %lld", &M, &N, &nz) != 3 || M <= 0 || M != N || nz < 0) { fclose(f); return PETSC_ERR_ARG_WRONG; }
    if (M > 2000000000LL || nz > (1LL << 40)) { fclose(f); return PETSC_ERR_ARG_OUTOFRANGE; }
    trip_t *t = (trip_t *)malloc(sizeof(trip_t) * (size_t)(2 * nz + 1));
    if (!t) { fclose(f); return PETSC_ERR_MEM; }
    PetscInt cnt = 0;
    for (long long k = 0; k < nz; ++k) {
        long long i, j;
        double v = 1.0;
        if (!fgets(line, sizeof line, f)) { free(t); fclose(f); return PETSC_ERR_ARG_WRONG; }
        const int got = pattern ? sscanf(line, "%lld %lld", &i, &j) : sscanf(line, "%lld %lld %lf", &i, &j, &v);
        if (got != (pattern ? 2 : 3) || i < 1 || i > M || j < 1 || j > N) { free(t); fclose(f); return PETSC_ERR_ARG_OUTOFRANGE; }
        t[cnt].r = i - 1; t[cnt].c = j - 1; t[cnt].v = v; ++cnt;
        if ((symm || skew) && i != j) { t[cnt].r = j - 1; t[cnt].c = i - 1; t[cnt].v = skew ? -v : v; ++cnt; }
    }
    fclose(f);
    qsort(t, (size_t)cnt, sizeof(trip_t), trip_cmp);
    PetscInt *ia = (PetscInt *)calloc((size_t)(M + 1), sizeof(PetscInt)), *ja = (PetscInt *)malloc(sizeof(PetscInt) * (size_t)(cnt > 0 ? cnt : 1));
    PetscScalar *a = (PetscScalar *)malloc(sizeof(PetscScalar) * (size_t)(cnt > 0 ? cnt : 1));
    if (!ia || !ja || !a) { free(ia); free(ja); free(a); free(t); return PETSC_ERR_MEM; }
    PetscInt q = 0;
    for (PetscInt k = 0; k < cnt; ++k) {
        if (q > 0 && k > 0 && t[k].r == t[k - 1].r && t[k].c == t[k - 1].c) { a[q - 1] += t[k].v; continue; } /* duplicates add */
        ja[q] = t[k].c; a[q] = t[k].v; ++q;
        ++ia[t[k].r + 1];
    }
    for (PetscInt i = 0; i < M; ++i) ia[i + 1] += ia[i];
    PetscErrorCode e = MatCreateSeqAIJWithArrays(M, ia, ja, a, A);
    free(t); free(ia); free(ja); free(a);
    return e;
}

PetscErrorCode MatViewMatrixMarket(Mat A, const char *path)
{
    PetscInt n;
    const PetscInt *ia, *ja;
    const PetscScalar *a;
    MatSeqAIJGetCSR(A, &n, &ia, &ja, &a);
    FILE *f = fopen(path, "w");
    if (!f) return PETSC_ERR_ARG_WRONG;
    fprintf(f, "%%%%MatrixMarket matrix coordinate real general\n%lld %lld %lld\n", (long long)n, (long long)n, (long long)ia[n]);
    for (PetscInt i = 0; i < n; ++i)
        for (PetscInt k = ia[i]; k < ia[i + 1]; ++k) fprintf(f, "%lld %lld %.17g\n", (long long)i + 1, (long long)ja[k] + 1, a[k]);
    fclose(f);
    return 0;
}
