// spike_csr.hip -- host steps of the CSR entry of the engine (spike_setup_csr itself is in spike_engine.hip).
//
// Follows MatCreateSubMatrixBanded, /root/reference/src/matbanded.c:22-107, line for line in
// MEANING (not in code): like the reference this step runs on the host in one pass order, so
// the chosen half-bandwidth k and the achieved fraction are the same numbers the reference's
// sequential sums produce (weights :38-49, stopping rule :53-56 incl. the k=kmax fall-through
// without adding w[kmax], copy of |c-r|<=k :84-99, outputs :104-105).  The extracted band goes
// straight into the diagonal-major layout the device factorisation consumes.
#include "../../include/spike_mi355.h"
#include <cmath>
#include <cstdlib>
#include <vector>

extern "C" int spike_csr_band_k(int64_t n, const int64_t *ia, const int64_t *ja, const double *a, int kmax, double frac,
                                int *k_out, double *frac_out)
{
    if (n <= 0 || !ia || !ja || !a || kmax < 0 || !k_out || !frac_out) return SPIKE_ERR_ARG;
    std::vector<double> w((size_t)n, 0.0);  // the reference's weight Vec has the matrix' row count (:34)
    double normA = 0.0, normB = 0.0;
    for (int64_t r = 0; r < n; ++r)
        for (int64_t p = ia[r]; p < ia[r + 1]; ++p) {
            if (ja[p] < 0 || ja[p] >= n) return SPIKE_ERR_ARG;
            const int64_t d = r > ja[p] ? r - ja[p] : ja[p] - r;
            w[(size_t)d] += std::fabs(a[p]);
            normA += std::fabs(a[p]);
        }
    int k;
    for (k = 0; k < kmax; ++k) {
        if (k < n) normB += w[(size_t)k];
        if (normB >= frac * normA) break;
    }
    *k_out = k;
    *frac_out = normB / normA;
    return SPIKE_OK;
}

// Row-block-distributed form of the weight pass (the reference's extraction is written for PETSc's row-block MPI
// layout: MatGetOwnershipRange src/matbanded.c:36, the |r - c| <= k split into diagonal/off-diagonal blocks :74-75; its
// weight Vec, however, is indexed locally (:45), so its own multi-rank result is not defined -- SURVEY.md section 2).
// This rank's rows [row0, row0 + n_local), GLOBAL columns: w[d] for d in [0, kmax) and the local part of ||A||_1, summed
// in the reference's row order.  The caller combines the ranks' parts in rank order (spike_setup_csr_dist).
extern "C" int spike_csr_band_weights(int64_t n_global, int64_t row0, int64_t n_local, const int64_t *ia, const int64_t *ja,
                                      const double *a, int kmax, double *w /* kmax */, double *normA)
{
    if (n_global <= 0 || row0 < 0 || n_local < 0 || row0 + n_local > n_global || !ia || !ja || !a || kmax < 0 || !normA || (kmax > 0 && !w))
        return SPIKE_ERR_ARG;
    for (int k = 0; k < kmax; ++k) w[k] = 0.0;
    double na = 0.0;
    for (int64_t r = 0; r < n_local; ++r)
        for (int64_t p = ia[r]; p < ia[r + 1]; ++p) {
            if (ja[p] < 0 || ja[p] >= n_global) return SPIKE_ERR_ARG;
            const int64_t gr = row0 + r, d = gr > ja[p] ? gr - ja[p] : ja[p] - gr;
            if (d < kmax) w[d] += std::fabs(a[p]);
            na += std::fabs(a[p]);
        }
    *normA = na;
    return SPIKE_OK;
}

// the stopping rule of src/matbanded.c:53-56 on already summed weights (w[d] for d < min(kmax, n))
extern "C" int spike_band_rule(int64_t n, const double *w, double normA, int kmax, double frac, int *k_out, double *frac_out)
{
    if (!k_out || !frac_out || kmax < 0 || (kmax > 0 && !w)) return SPIKE_ERR_ARG;
    double normB = 0.0;
    int k;
    for (k = 0; k < kmax; ++k) {
        if (k < n) normB += w[k];
        if (normB >= frac * normA) break;
    }
    *k_out = k;
    *frac_out = normB / normA;
    return SPIKE_OK;
}

extern "C" int spike_csr_to_band(int64_t n, const int64_t *ia, const int64_t *ja, const double *a, int K, double *band,
                                 int64_t ld)
{
    if (n <= 0 || !ia || !ja || !a || K < 0 || !band || ld < n) return SPIKE_ERR_ARG;
    for (int d = 0; d <= 2 * K; ++d)
        for (int64_t i = 0; i < n; ++i) band[(size_t)d * ld + i] = 0.0;
    for (int64_t r = 0; r < n; ++r)
        for (int64_t p = ia[r]; p < ia[r + 1]; ++p) {
            const int64_t d = ja[p] - r + K;
            if (d >= 0 && d <= 2 * K) band[(size_t)d * ld + r] = a[p];   // a repeated pair keeps the last value (INSERT_VALUES, :98)
        }
    return SPIKE_OK;
}


// 32-bit index variants (PETSc's default PetscInt): the same rule on widened copies
static void widen(int64_t n, const int32_t *ia, const int32_t *ja, std::vector<int64_t> &ia64, std::vector<int64_t> &ja64)
{
    ia64.resize((size_t)n + 1);
    for (int64_t i = 0; i <= n; ++i) ia64[(size_t)i] = ia[i];
    const int64_t nnz = ia64[(size_t)n] > 0 ? ia64[(size_t)n] : 0;
    ja64.resize((size_t)(nnz > 0 ? nnz : 1));
    for (int64_t q = 0; q < nnz; ++q) ja64[(size_t)q] = ja[q];
}
extern "C" int spike_csr_band_k32(int64_t n, const int32_t *ia, const int32_t *ja, const double *a, int kmax, double frac,
                                  int *k_out, double *frac_out)
{
    if (n <= 0 || !ia || !ja) return SPIKE_ERR_ARG;
    std::vector<int64_t> ia64, ja64;
    widen(n, ia, ja, ia64, ja64);
    return spike_csr_band_k(n, ia64.data(), ja64.data(), a, kmax, frac, k_out, frac_out);
}
extern "C" int spike_csr_band_weights32(int64_t n_global, int64_t row0, int64_t n_local, const int32_t *ia, const int32_t *ja,
                                        const double *a, int kmax, double *w, double *normA)
{
    if (n_local < 0 || !ia || !ja) return SPIKE_ERR_ARG;
    std::vector<int64_t> ia64, ja64;
    widen(n_local, ia, ja, ia64, ja64);
    return spike_csr_band_weights(n_global, row0, n_local, ia64.data(), ja64.data(), a, kmax, w, normA);
}
