// spike_reorder.hip -- the reordering front-end's bulk data movement and greedy matching phases on the device.
//
// Reference slots (paths relative to /root/reference):
//   spike_permute_csr   MatPermute(M, rorder, corder, &PM)        src/kspreorder.c:20-22  (PETSc's own routine there)
//   spike_permute_vec   VecPermute(x, corder, PETSC_FALSE / TRUE) src/kspreorder.c:122-127
//   spike_awbm_device   MatGetOrdering_AWBM                       src/petsc_mat_awbm.c:42-225: phases 1 (tight-edge greedy,
//                       :98-112) and 3 (any-edge greedy, :143-153) as a parallel fixed-point iteration that reproduces the
//                       sequential greedy exactly; phases 2, 4, 5 (one-step augmentations, default fill) run on the host on
//                       what is left -- they rewrite matches as they go and are a few percent of the columns.
// All of it is integer / byte work: results are IDENTICAL to the host loops of csrc/host/sp_host.c (MatPermute, VecPermute)
// and csrc/host/awbm.c, which the tests check bit for bit.  gfx950 only; no CPU fallback inside (a caller without a device
// gets SPIKE_ERR_HIP and uses its own host loop, as the host mirror does).
#include "../../include/spike_mi355.h"

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <vector>

namespace {

#define RCHK(call) do { if ((call) != hipSuccess) { (void)hipGetLastError(); return SPIKE_ERR_HIP; } } while (0)

struct DevPool {   // device scratch of one call, released on every exit path
    std::vector<void *> p;
    template <class T> hipError_t alloc(T **q, size_t count)
    {
        *q = nullptr;
        hipError_t e = hipMalloc((void **)q, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) p.push_back((void *)*q);
        return e;
    }
    ~DevPool() { for (void *q : p) (void)hipFree(q); }
};

// ---- exclusive scan of n int64 values: per-block sums, a one-block scan of those, then the blocks ---------------------------
constexpr int SCAN_T = 256, SCAN_E = 8, SCAN_B = SCAN_T * SCAN_E;   // 2048 values per block

__global__ __launch_bounds__(SCAN_T) void k_scan_blocks(const int64_t *in, int64_t n, int64_t *out, int64_t *bsum)
{
    __shared__ int64_t sh[SCAN_T];
    const int64_t base = (int64_t)blockIdx.x * SCAN_B + (int64_t)threadIdx.x * SCAN_E;
    int64_t v[SCAN_E], t = 0;
#pragma unroll
    for (int e = 0; e < SCAN_E; ++e) { v[e] = base + e < n ? in[base + e] : 0; t += v[e]; }
    sh[threadIdx.x] = t;
    __syncthreads();
    for (int o = 1; o < SCAN_T; o <<= 1) {   // Hillis-Steele inclusive scan of the thread totals
        const int64_t a = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += a;
        __syncthreads();
    }
    int64_t run = sh[threadIdx.x] - t;       // exclusive prefix of this thread inside the block
#pragma unroll
    for (int e = 0; e < SCAN_E; ++e) { if (base + e < n) out[base + e] = run; run += v[e]; }
    if (threadIdx.x == SCAN_T - 1 && bsum) bsum[blockIdx.x] = sh[SCAN_T - 1];
}

__global__ __launch_bounds__(SCAN_T) void k_scan_add(int64_t *out, int64_t n, const int64_t *boff)
{
    const int64_t base = (int64_t)blockIdx.x * SCAN_B + (int64_t)threadIdx.x * SCAN_E;
    const int64_t o = boff[blockIdx.x];
#pragma unroll
    for (int e = 0; e < SCAN_E; ++e) if (base + e < n) out[base + e] += o;
}

// out[i] = sum_{j<i} in[j] for i < n (out may alias in); recursion depth <= 3 for n < 2^33
int exclusive_scan(const int64_t *in, int64_t n, int64_t *out, DevPool &pool, hipStream_t st)
{
    if (n <= 0) return SPIKE_OK;
    const int64_t nb = (n + SCAN_B - 1) / SCAN_B;
    int64_t *bsum = nullptr;
    RCHK(pool.alloc(&bsum, (size_t)nb));
    hipLaunchKernelGGL(k_scan_blocks, dim3((unsigned)nb), dim3(SCAN_T), 0, st, in, n, out, bsum);
    RCHK(hipGetLastError());
    if (nb > 1) {
        int rc = exclusive_scan(bsum, nb, bsum, pool, st);
        if (rc) return rc;
        hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nb), dim3(SCAN_T), 0, st, out, n, bsum);
        RCHK(hipGetLastError());
    }
    return SPIKE_OK;
}

// ---- MatPermute ------------------------------------------------------------------------------------------------------------------
// B = A(rowp, colp): row i of B is row rowp[i] of A, its column c becomes the position of c in colp (sp_host.c: MatPermute);
// rows of B sorted by column.
__global__ void k_inverse_perm(const int64_t *perm, int64_t n, int64_t *inv, int *bad)
{
    const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int64_t c = perm[j];
    if (c < 0 || c >= n) { atomicExch(bad, 1); return; }
    // a repeated target would be written twice: detected by counting (a permutation hits every slot exactly once)
    if (atomicAdd((unsigned long long *)&inv[c], (unsigned long long)(j + 1)) != 0) atomicExch(bad, 1);
}
__global__ void k_dec(int64_t *v, int64_t n)
{
    const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (j < n) v[j] -= 1;
}
__global__ void k_row_lengths(const int64_t *ia, const int64_t *rowp, int64_t n, int64_t *len, int *bad)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t r = rowp[i];
    if (r < 0 || r >= n) { atomicExch(bad, 1); len[i] = 0; return; }
    len[i] = ia[r + 1] - ia[r];
}
// one thread per row: copy with the column map, then insertion sort by column (rows are short; the host loop does the same)
__global__ void k_permute_rows(const int64_t *ia, const int64_t *ja, const double *a, const int64_t *rowp, const int64_t *icol,
                               int64_t n, const int64_t *ib, int64_t *jb, double *b)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t r = rowp[i], s0 = ia[r], len = ia[r + 1] - s0, d0 = ib[i];
    for (int64_t k = 0; k < len; ++k) {
        const int64_t cj = icol[ja[s0 + k]];
        const double cv = a[s0 + k];
        int64_t t = k - 1;
        while (t >= 0 && jb[d0 + t] > cj) { jb[d0 + t + 1] = jb[d0 + t]; b[d0 + t + 1] = b[d0 + t]; --t; }
        jb[d0 + t + 1] = cj;
        b[d0 + t + 1] = cv;
    }
}

__global__ void k_vec_permute(const double *x, const int64_t *idx, int64_t n, int inverse, double *y)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!inverse) y[i] = x[idx[i]];
    else y[idx[i]] = x[i];
}

// ---- AWBM: the greedy phases as a fixed-point iteration ---------------------------------------------------------------------
// Sequential rule (petsc_mat_awbm.c:98-112 / :143-153): columns in index order; column c takes the FIRST entry of its list whose
// row is eligible and not taken by an earlier column.  Fixed point: holder[r] = the smallest column proposing r, where column c
// proposes the first eligible row r with holder[r] >= c (i.e. not held by a smaller column).  Column 0's proposal is final after
// round 1, column c's once all smaller columns are final (induction), so iterating from holder = "nobody" reaches the sequential
// result; the number of rounds is the longest chain of displacements, not n.  A round rebuilds holder from scratch (atomicMin
// over the proposals): deterministic, no dependence on thread timing.
//   tight != 0: only entries with (w - u[row]) - v[c] <= eps are eligible (phase 1); else every entry (phase 3).
//   matchR[r] >= 0 marks rows taken by earlier phases; active[c] != 0 marks the columns still to be matched.
__global__ void k_awbm_propose(int64_t n, const int64_t *ia, const int64_t *ja, const double *w, const double *u, const double *v,
                               double eps, int tight, const int64_t *matchR, const unsigned char *active, const int64_t *holder,
                               int64_t *holder_new, int64_t *cand)
{
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= n) return;
    int64_t pick = -1;
    if (active[c]) {
        for (int64_t k = ia[c]; k < ia[c + 1]; ++k) {
            const int64_t r = ja[k];
            if (matchR[r] >= 0) continue;
            if (tight && !((w[k] - u[r]) - v[c] <= eps)) continue;
            if (holder[r] < c) continue;          // held by an earlier column
            pick = r;
            break;
        }
    }
    cand[c] = pick;
    if (pick >= 0) atomicMin((unsigned long long *)&holder_new[pick], (unsigned long long)c);
}
__global__ void k_fill64(int64_t *v, int64_t n, int64_t val)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) v[i] = val;
}
__global__ void k_differs(const int64_t *a, const int64_t *b, int64_t n, int *flag)
{
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n && a[i] != b[i]) atomicExch(flag, 1);
}
// after convergence: column c holds cand[c] iff holder[cand[c]] == c (always so at the fixed point)
__global__ void k_awbm_commit(int64_t n, const int64_t *cand, const int64_t *holder, int64_t *match, int64_t *matchR, unsigned char *active)
{
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= n || !active[c]) return;
    const int64_t r = cand[c];
    if (r >= 0 && holder[r] == c) { match[c] = r; matchR[r] = c; active[c] = 0; }
}
__global__ void k_awbm_active(int64_t n, const int64_t *match, unsigned char *active)
{
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c < n) active[c] = match[c] < 0;
}
// u[r] = min over the entries of row-index r of w (a min of doubles is exact, whatever the order); v[c] = min_k (w_k - u[ja_k])
__global__ void k_awbm_u(int64_t n, const int64_t *ia, const int64_t *ja, const double *w, unsigned long long *u_bits)
{
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= n) return;
    // weights are >= 0 (log(rowmax / |a|)) or DBL_MAX: for non-negative doubles the bit pattern orders like the value
    for (int64_t k = ia[c]; k < ia[c + 1]; ++k) atomicMin(&u_bits[ja[k]], (unsigned long long)__double_as_longlong(w[k]));
}
__global__ void k_awbm_v(int64_t n, const int64_t *ia, const int64_t *ja, const double *w, const double *u, double *v)
{
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= n) return;
    double m = DBL_MAX;
    for (int64_t k = ia[c]; k < ia[c + 1]; ++k) { const double t = w[k] - u[ja[k]]; if (t < m) m = t; }
    v[c] = m;
}

inline dim3 g1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

// ---- C-ABI ---------------------------------------------------------------------------------------------------------------------
extern "C" int spike_permute_vec(int64_t n, const int64_t *idx, int inverse, const double *x, double *y, int on_device)
{
    if (n <= 0 || !idx || !x || !y || x == y) return SPIKE_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return SPIKE_ERR_HIP; }
    if (on_device) {
        hipLaunchKernelGGL(k_vec_permute, g1(n), dim3(256), 0, nullptr, x, idx, n, inverse, y);
        RCHK(hipGetLastError());
        RCHK(hipStreamSynchronize(nullptr));
        return SPIKE_OK;
    }
    for (int64_t i = 0; i < n; ++i) if (idx[i] < 0 || idx[i] >= n) return SPIKE_ERR_ARG;
    DevPool pool;
    double *dx = nullptr, *dy = nullptr;
    int64_t *di = nullptr;
    RCHK(pool.alloc(&dx, (size_t)n)); RCHK(pool.alloc(&dy, (size_t)n)); RCHK(pool.alloc(&di, (size_t)n));
    RCHK(hipMemcpy(dx, x, sizeof(double) * n, hipMemcpyHostToDevice));
    RCHK(hipMemcpy(di, idx, sizeof(int64_t) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_vec_permute, g1(n), dim3(256), 0, nullptr, dx, di, n, inverse, dy);
    RCHK(hipGetLastError());
    RCHK(hipMemcpy(y, dy, sizeof(double) * n, hipMemcpyDeviceToHost));
    return SPIKE_OK;
}

extern "C" int spike_permute_csr(int64_t n, const int64_t *ia, const int64_t *ja, const double *a, const int64_t *rowp,
                                 const int64_t *colp, int64_t *ib, int64_t *jb, double *b)
{
    if (n <= 0 || !ia || !ja || !a || !rowp || !colp || !ib || !jb || !b) return SPIKE_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return SPIKE_ERR_HIP; }
    const int64_t nnz = ia[n];
    if (nnz < 0) return SPIKE_ERR_ARG;
    for (int64_t k = 0; k < nnz; ++k) if (ja[k] < 0 || ja[k] >= n) return SPIKE_ERR_ARG;
    DevPool pool;
    hipStream_t st = nullptr;
    int64_t *dia = nullptr, *dja = nullptr, *drow = nullptr, *dcol = nullptr, *dicol = nullptr, *dib = nullptr, *djb = nullptr;
    double *da = nullptr, *db = nullptr;
    int *dbad = nullptr;
    RCHK(pool.alloc(&dia, (size_t)n + 1)); RCHK(pool.alloc(&dja, (size_t)nnz)); RCHK(pool.alloc(&da, (size_t)nnz));
    RCHK(pool.alloc(&drow, (size_t)n)); RCHK(pool.alloc(&dcol, (size_t)n)); RCHK(pool.alloc(&dicol, (size_t)n));
    RCHK(pool.alloc(&dib, (size_t)n + 1)); RCHK(pool.alloc(&djb, (size_t)nnz)); RCHK(pool.alloc(&db, (size_t)nnz));
    RCHK(pool.alloc(&dbad, 1));
    RCHK(hipMemcpyAsync(dia, ia, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, st));
    RCHK(hipMemcpyAsync(dja, ja, sizeof(int64_t) * nnz, hipMemcpyHostToDevice, st));
    RCHK(hipMemcpyAsync(da, a, sizeof(double) * nnz, hipMemcpyHostToDevice, st));
    RCHK(hipMemcpyAsync(drow, rowp, sizeof(int64_t) * n, hipMemcpyHostToDevice, st));
    RCHK(hipMemcpyAsync(dcol, colp, sizeof(int64_t) * n, hipMemcpyHostToDevice, st));
    RCHK(hipMemsetAsync(dbad, 0, sizeof(int), st));
    RCHK(hipMemsetAsync(dicol, 0, sizeof(int64_t) * n, st));
    hipLaunchKernelGGL(k_inverse_perm, g1(n), dim3(256), 0, st, dcol, n, dicol, dbad);   // icol[c] = (position of c in colp) + 1
    hipLaunchKernelGGL(k_dec, g1(n), dim3(256), 0, st, dicol, n);
    hipLaunchKernelGGL(k_row_lengths, g1(n), dim3(256), 0, st, dia, drow, n, dib, dbad);
    RCHK(hipGetLastError());
    RCHK(hipMemsetAsync(dib + n, 0, sizeof(int64_t), st));
    int rc = exclusive_scan(dib, n + 1, dib, pool, st);
    if (rc) return rc;
    int bad = 0;
    RCHK(hipMemcpyAsync(&bad, dbad, sizeof(int), hipMemcpyDeviceToHost, st));
    RCHK(hipStreamSynchronize(st));
    if (bad) return SPIKE_ERR_ARG;   // not permutations
    hipLaunchKernelGGL(k_permute_rows, g1(n), dim3(256), 0, st, dia, dja, da, drow, dicol, n, dib, djb, db);
    RCHK(hipGetLastError());
    RCHK(hipMemcpyAsync(ib, dib, sizeof(int64_t) * (n + 1), hipMemcpyDeviceToHost, st));
    RCHK(hipMemcpyAsync(jb, djb, sizeof(int64_t) * nnz, hipMemcpyDeviceToHost, st));
    RCHK(hipMemcpyAsync(b, db, sizeof(double) * nnz, hipMemcpyDeviceToHost, st));
    RCHK(hipStreamSynchronize(st));
    return SPIKE_OK;
}

// Approximate weighted matching with the two greedy phases on the device.  perm[match[c]] = c as spike_awbm (libspike_petsc_host);
// rounds (optional) = fixed-point rounds of phase 1 and of phase 3.  The log weights are computed on the HOST (libm's log and
// the device's differ in the last bit, and the tight-edge test compares reduced costs with eps = sqrt(DBL_EPSILON): identical
// matchings need identical weights); u, v, the greedy phases and their bookkeeping are exact operations and run on the device.
extern "C" int spike_awbm_device(int64_t n, const int64_t *ia, const int64_t *ja, const double *a, int64_t *perm, int *rounds)
{
    if (n <= 0 || !ia || !ja || !a || !perm) return SPIKE_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); return SPIKE_ERR_HIP; }
    const int64_t nnz = ia[n];
    if (nnz < 0) return SPIKE_ERR_ARG;
    const double eps = std::sqrt(DBL_EPSILON);
    std::vector<double> w((size_t)(nnz > 0 ? nnz : 1));
    for (int64_t c = 0; c < n; ++c) {   // petsc_mat_awbm.c:65, 72-80: w = log(rowmax / |a|), zero entry -> the largest real
        double amax = 0.0;
        for (int64_t k = ia[c]; k < ia[c + 1]; ++k) { if (ja[k] < 0 || ja[k] >= n) return SPIKE_ERR_ARG; if (std::fabs(a[k]) > amax) amax = std::fabs(a[k]); }
        for (int64_t k = ia[c]; k < ia[c + 1]; ++k) { const double ar = std::fabs(a[k]); w[(size_t)k] = (ar == 0.0) ? DBL_MAX : std::log(amax / ar); }
    }
    DevPool pool;
    hipStream_t st = nullptr;
    int64_t *dia = nullptr, *dja = nullptr, *dmatch = nullptr, *dmatchR = nullptr, *dh0 = nullptr, *dh1 = nullptr, *dcand = nullptr;
    double *dw = nullptr, *du = nullptr, *dv = nullptr;
    unsigned char *dact = nullptr;
    int *dflag = nullptr;
    RCHK(pool.alloc(&dia, (size_t)n + 1)); RCHK(pool.alloc(&dja, (size_t)nnz)); RCHK(pool.alloc(&dw, (size_t)nnz));
    RCHK(pool.alloc(&du, (size_t)n)); RCHK(pool.alloc(&dv, (size_t)n)); RCHK(pool.alloc(&dmatch, (size_t)n)); RCHK(pool.alloc(&dmatchR, (size_t)n));
    RCHK(pool.alloc(&dh0, (size_t)n)); RCHK(pool.alloc(&dh1, (size_t)n)); RCHK(pool.alloc(&dcand, (size_t)n)); RCHK(pool.alloc(&dact, (size_t)n));
    RCHK(pool.alloc(&dflag, 1));
    RCHK(hipMemcpyAsync(dia, ia, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, st));
    RCHK(hipMemcpyAsync(dja, ja, sizeof(int64_t) * nnz, hipMemcpyHostToDevice, st));
    RCHK(hipMemcpyAsync(dw, w.data(), sizeof(double) * nnz, hipMemcpyHostToDevice, st));
    // u = DBL_MAX, then the row minima (:82-88); v (:90-96)
    hipLaunchKernelGGL(k_fill64, g1(n), dim3(256), 0, st, (int64_t *)du, n, (int64_t)0x7FEFFFFFFFFFFFFFLL);   // bits of DBL_MAX
    hipLaunchKernelGGL(k_awbm_u, g1(n), dim3(256), 0, st, n, dia, dja, dw, (unsigned long long *)du);
    hipLaunchKernelGGL(k_awbm_v, g1(n), dim3(256), 0, st, n, dia, dja, dw, du, dv);
    hipLaunchKernelGGL(k_fill64, g1(n), dim3(256), 0, st, dmatch, n, (int64_t)-1);
    hipLaunchKernelGGL(k_fill64, g1(n), dim3(256), 0, st, dmatchR, n, (int64_t)-1);
    RCHK(hipGetLastError());
    std::vector<int64_t> match((size_t)n), matchR((size_t)n);
    std::vector<double> u((size_t)n), v((size_t)n);
    int nrounds[2] = {0, 0};
    auto greedy = [&](int tight, int *count) -> int {
        hipLaunchKernelGGL(k_awbm_active, g1(n), dim3(256), 0, st, n, dmatch, dact);
        hipLaunchKernelGGL(k_fill64, g1(n), dim3(256), 0, st, dh0, n, (int64_t)n);     // "nobody": larger than every column
        int64_t *hold = dh0, *hnew = dh1;
        for (int64_t round = 0; round <= n; ++round) {
            hipLaunchKernelGGL(k_fill64, g1(n), dim3(256), 0, st, hnew, n, (int64_t)n);
            hipLaunchKernelGGL(k_awbm_propose, g1(n), dim3(256), 0, st, n, dia, dja, dw, du, dv, eps, tight, dmatchR, dact, hold, hnew, dcand);
            RCHK(hipMemsetAsync(dflag, 0, sizeof(int), st));
            hipLaunchKernelGGL(k_differs, g1(n), dim3(256), 0, st, hold, hnew, n, dflag);
            RCHK(hipGetLastError());
            int changed = 0;
            RCHK(hipMemcpyAsync(&changed, dflag, sizeof(int), hipMemcpyDeviceToHost, st));
            RCHK(hipStreamSynchronize(st));
            ++*count;
            int64_t *t = hold; hold = hnew; hnew = t;
            if (!changed) break;
        }
        // hold == the fixed point; cand was computed against the previous (equal) holder array
        hipLaunchKernelGGL(k_awbm_commit, g1(n), dim3(256), 0, st, n, dcand, hold, dmatch, dmatchR, dact);
        RCHK(hipGetLastError());
        return SPIKE_OK;
    };
    auto download = [&]() -> int {
        RCHK(hipMemcpyAsync(match.data(), dmatch, sizeof(int64_t) * n, hipMemcpyDeviceToHost, st));
        RCHK(hipMemcpyAsync(matchR.data(), dmatchR, sizeof(int64_t) * n, hipMemcpyDeviceToHost, st));
        RCHK(hipStreamSynchronize(st));
        return SPIKE_OK;
    };
    auto upload = [&]() -> int {
        RCHK(hipMemcpyAsync(dmatch, match.data(), sizeof(int64_t) * n, hipMemcpyHostToDevice, st));
        RCHK(hipMemcpyAsync(dmatchR, matchR.data(), sizeof(int64_t) * n, hipMemcpyHostToDevice, st));
        return SPIKE_OK;
    };
    int rc;
    if ((rc = greedy(1, &nrounds[0]))) return rc;                       // phase 1
    if ((rc = download())) return rc;
    RCHK(hipMemcpyAsync(u.data(), du, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    RCHK(hipMemcpyAsync(v.data(), dv, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    RCHK(hipStreamSynchronize(st));
    for (int64_t c = 0; c < n; ++c) {                                  // phase 2 (:115-140): one augmentation step through tight edges
        if (match[(size_t)c] >= 0) continue;
        for (int64_t k = ia[c]; k < ia[c + 1]; ++k) {
            if ((w[(size_t)k] - u[(size_t)ja[k]]) - v[(size_t)c] > eps) continue;
            const int64_t c1 = matchR[(size_t)ja[k]];
            if (c1 < 0) continue;
            for (int64_t k1 = ia[c1]; k1 < ia[c1 + 1]; ++k1)
                if (matchR[(size_t)ja[k1]] < 0 && (w[(size_t)k1] - u[(size_t)ja[k1]]) - v[(size_t)c1] <= eps) {
                    match[(size_t)c] = ja[k]; matchR[(size_t)ja[k]] = c;
                    match[(size_t)c1] = ja[k1]; matchR[(size_t)ja[k1]] = c1;
                    break;
                }
            if (match[(size_t)c] >= 0) break;
        }
    }
    if ((rc = upload())) return rc;
    if ((rc = greedy(0, &nrounds[1]))) return rc;                       // phase 3
    if ((rc = download())) return rc;
    for (int64_t c = 0; c < n; ++c) {                                  // phase 4 (:156-178): one augmentation step through any edge
        if (match[(size_t)c] >= 0) continue;
        for (int64_t k = ia[c]; k < ia[c + 1]; ++k) {
            const int64_t c1 = matchR[(size_t)ja[k]];
            if (c1 < 0) continue;
            for (int64_t k1 = ia[c1]; k1 < ia[c1 + 1]; ++k1)
                if (matchR[(size_t)ja[k1]] < 0) {
                    match[(size_t)c] = ja[k]; matchR[(size_t)ja[k]] = c;
                    match[(size_t)c1] = ja[k1]; matchR[(size_t)ja[k1]] = c1;
                    break;
                }
            if (match[(size_t)c] >= 0) break;
        }
    }
    for (int64_t c = 0, r = 0; c < n; ++c) {                            // phase 5 (:181-193): default fill, shared row cursor
        if (match[(size_t)c] >= 0) continue;
        for (; r < n; ++r)
            if (matchR[(size_t)r] < 0) { match[(size_t)c] = r; matchR[(size_t)r] = c; break; }
    }
    for (int64_t c = 0; c < n; ++c) if (match[(size_t)c] < 0 || match[(size_t)c] >= n) return SPIKE_ERR_STATE;
    for (int64_t c = 0; c < n; ++c) perm[match[(size_t)c]] = c;
    if (rounds) { rounds[0] = nrounds[0]; rounds[1] = nrounds[1]; }
    return SPIKE_OK;
}
