/*
 * spike_orderings.h -- the reordering kernels of libspike_petsc_host.so as plain C (no PETSc types), so that BOTH the
 * host mirror (include/spike_petsc_host.h) and real PETSc glue (examples/petsc/kspreorder_spike.c, which includes PETSc's
 * own headers and therefore cannot include the mirror's typedefs) can declare them.
 *
 * Reference slots: HSLmc64AD job 5 (/root/reference/src/hslmc64.c:305-976, called at src/petsc_mat_wbm.c:52),
 * the AWBM phases (src/petsc_mat_awbm.c:98-193), hslmc73_ (src/petsc_mat_fiedler.c:45; HSL_MC73 absent: own spec, see
 * csrc/host/fiedler.c), the "rcm" second stage (src/HOWTO:2).  All arrays 0-based.
 */
#ifndef SPIKE_ORDERINGS_H
#define SPIKE_ORDERINGS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int spike_mc64_job5(int64_t n, const int64_t *colptr, const int64_t *rowind, const double *val, int64_t *perm,
                    double *u, double *v, int64_t *num);
int spike_awbm(int64_t n, const int64_t *ia, const int64_t *ja, const double *a, int64_t *perm, double *sr, double *sc);
/* the same matching for a matrix distributed by rows (MatComputeMatching_MPIAIJ, src/wbm.c:201-440; csrc/host/awbm_dist.c):
 * every rank (1) computes its contribution to the per-column minima over all N columns, (2) the CALLER reduces those with MIN
 * over the ranks (MPI_Allreduce(..., MPI_MIN, comm) in a PETSc binding -- the reference's own "TODO ... MPI_MIN", :270), (3)
 * every rank matches its own rows inside its diagonal block: perm over the local indices, perm[match[c]] = c */
int spike_awbm_dist_rowmin(int64_t n_local, int64_t N, const int64_t *ia, const int64_t *ja, const double *a, double *umin);
int spike_awbm_dist_match(int64_t n_local, int64_t row0, int64_t N, const int64_t *ia, const int64_t *ja, const double *a,
                          const double *u, int64_t *perm, double *sr, double *sc);
int spike_fiedler_order(int64_t n, const int64_t *ia, const int64_t *ja, const double *a, int64_t *order, double *vec);
/* use_device != 0 and a HIP device present: the LOBPCG refinement of the large multilevel levels runs on the GPU
 * (libspike_mi355: spike_fd_*), with a bit-identical permutation */
int spike_fiedler_order_ex(int64_t n, const int64_t *ia, const int64_t *ja, const double *a, int64_t *order, double *vec,
                           int use_device);
int spike_rcm_order(int64_t n, const int64_t *ia, const int64_t *ja, int64_t *order);
/* Fiedler bisection + reverse Cuthill-McKee on each half's diagonal block, composed -- the per-half reordering prototyped
 * in src/spectralPartition.c:326-417.  pos_size: rows in the positive half; halves_bw[4] (optional): bandwidth of the
 * positive / negative block before and after its own reordering.  Ordering name in the registry: "fiedler_halves". */
int spike_fiedler_halves_order(int64_t n, const int64_t *ia, const int64_t *ja, const double *a, int64_t *order,
                               int64_t *pos_size, int64_t *halves_bw, int use_device);
/* the same kernels for 32-bit index arrays (PETSc's default PetscInt; csrc/host/idx32.c): inputs widened, permutations
 * narrowed, results those of the 64-bit entry points */
int spike_mc64_job5_i32(int32_t n, const int32_t *colptr, const int32_t *rowind, const double *val, int32_t *perm, double *u,
                        double *v, int32_t *num);
int spike_awbm_i32(int32_t n, const int32_t *ia, const int32_t *ja, const double *a, int32_t *perm, double *sr, double *sc);
int spike_fiedler_order_i32(int32_t n, const int32_t *ia, const int32_t *ja, const double *a, int32_t *order, double *vec,
                            int use_device);
int spike_rcm_order_i32(int32_t n, const int32_t *ia, const int32_t *ja, int32_t *order);
int spike_profile_bandwidth(int64_t n, const int64_t *ia, const int64_t *ja, const int64_t *order, int64_t *profile,
                            int64_t *bandwidth);


#ifdef __cplusplus
}
#endif
#endif
