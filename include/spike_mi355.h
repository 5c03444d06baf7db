/*
 * spike_mi355.h -- C-ABI of the MI355X-native SPIKE banded preconditioner engine.
 *
 * This library fills the one slot the reference leaves to "whatever PETSc provides":
 * the inner PC of PCBANDED.  Each entry point names the reference interface it
 * replaces (paths relative to /root/reference):
 *
 *   spike_create / spike_destroy   PCCreate(...,&b->pc) src/matbanded.c:278, PCDestroy(&b->pc) :142
 *   spike_set_option               PCSetFromOptions(b->pc) src/matbanded.c:159 (options prefix "banded_", :281)
 *   spike_setup_band               PCSetOperators(b->pc,pc->mat,b->B)+PCSetUp(b->pc) src/matbanded.c:176-178
 *   spike_setup_csr[_dist]         MatCreateSubMatrixBanded + PCSetUp(b->pc)  src/matbanded.c:22-107,174-178
 *                                  (_dist: row-block MPI layout, MatGetOwnershipRange :36)
 *   spike_apply                    PCApply(b->pc,x,y) src/matbanded.c:190   (the metric's "PCApply")
 *   spike_reset                    PCReset(b->pc) src/matbanded.c:127
 *   spike_view                     PCView(b->pc,viewer) src/matbanded.c:207
 *   spike_gmres                    KSPSolve(ksp,b,x) src/testbed2.c:128 with -ksp_type gmres (src/makefile:18)
 *   spike_comm_*                   the communicator of the PC, PetscObjectComm((PetscObject)pc) src/matbanded.c:278
 *
 * Plain C types only: pointers, sizes, ints.  All functions return 0 on success or a
 * negative spike_status; spike_last_error() gives the message (the reference's
 * PetscErrorCode/SETERRQ convention, src/matbanded.c:95).
 *
 * Band layout ("diagonal-major"):  band[d*ld + i] = A[i, i + d - K],  d in [0,2K], i in [0,n).
 * Entries whose column falls outside [0,N) are ignored.  fp64 throughout.
 *
 * Threading: one host thread drives one handle; one process per GPU.  Multi-GPU jobs
 * give every rank a handle that owns a contiguous block of rows, and the ranks are
 * joined with spike_comm_init (RCCL).
 */
#ifndef SPIKE_MI355_H
#define SPIKE_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spike_handle_s *spike_handle;

typedef enum {
    SPIKE_OK = 0,
    SPIKE_ERR_ARG = -1,       /* bad argument / unsupported size */
    SPIKE_ERR_HIP = -2,       /* HIP runtime error (message in spike_last_error) */
    SPIKE_ERR_STATE = -3,     /* call order (apply before setup ...) */
    SPIKE_ERR_PARTITION = -4, /* P does not fit N (needs >= P 64-row blocks and rows/partition >= K) */
    SPIKE_ERR_COMM = -5,      /* RCCL error */
    SPIKE_ERR_SINGULAR = -6,  /* an interface system is singular */
    SPIKE_ERR_NOMEM = -7
} spike_status;

#define SPIKE_VARIANT_DECOUPLED 0 /* block-Jacobi over the P partitions: one pass over the factors */
#define SPIKE_VARIANT_COUPLED 1   /* truncated SPIKE: pass, interface solves, corrected pass        */

#define SPIKE_UNIQUE_ID_BYTES 128

typedef struct {
    int64_t n_local;       /* rows owned by this handle */
    int64_t n_global;      /* rows of the whole system  */
    int64_t row0;          /* first global row owned    */
    int32_t K;             /* half-bandwidth given      */
    int32_t Kp;            /* half-bandwidth the sweep kernels stream (K padded) */
    int32_t P_local;       /* partitions on this handle */
    int32_t P_global;
    int32_t variant;
    int32_t rows_per_block;  /* R */
    int32_t waves_per_chain; /* NW */
    int32_t nranks, rank;
    int64_t nboost;          /* pivots boosted during factorisation */
    int64_t factor_bytes;    /* bytes of packed factors streamed by ONE pass (L tiles + U tiles + 1/diag) */
    int64_t iface_bytes;     /* bytes of interface matrices read per coupled apply */
    double setup_ms;         /* wall time of the last setup (device work, synchronised) */
    int32_t k_extracted;     /* spike_setup_csr: chosen half-bandwidth (matbanded.c:104) */
    double frac_extracted;   /* spike_setup_csr: achieved norm fraction (matbanded.c:105) */
    int32_t passes;          /* passes over the packed factors per apply: 1, or 2 (coupled variant re-solving) */
    int32_t spike_rows;      /* rows kept of every spike (0: none, the coupled variant re-solves) */
    int64_t spike_bytes;     /* bytes of stored spikes read per coupled apply */
    int32_t chains_local;    /* chains the kernels sweep: P_local, or a multiple of it when "subsplit" cut the partitions */
    int32_t twisted;         /* 1: the chains are paired into diagonal blocks factored from both ends (exact seam system per pair,
                                truncated interfaces with stored spikes only between pairs); P_local then counts the blocks' owners,
                                i.e. the caller's partitions, or chains_local / 2 when the library chose them */
    int32_t spike_rows_fp64; /* of spike_rows, the rows next to the interface kept in fp64 (the others in fp32) */
    int32_t seams_local;     /* seam systems solved between the two sweep launches (= chains_local / 2 when twisted) */
} spike_info;

/* ---- lifecycle ------------------------------------------------------------------ */
int spike_create(spike_handle *h);
int spike_destroy(spike_handle h);
int spike_reset(spike_handle h); /* drop the factors, keep options and communicator */
const char *spike_last_error(spike_handle h);

/* keys (round 3 additions at the end): "partitions" (int >=1, or 0 = auto), "variant" ("decoupled"|"coupled"|0|1),
 *       "boost" (double, relative to max|diag|, default 1e-10), "keep_band" (0|1, default 1),
 *       "spike_storage" ("auto"|"off": keep the spikes' decayed part and apply the coupled variant in ONE pass
 *        when they are short, else/off: second pass over the factors), "spike_tol" (relative drop level, 1e-16),
 *       "subsplit" ("auto"|"off": a caller-chosen partition count is honoured, but each partition may be swept as
 *        several chains when setup MEASURES that the spikes die inside a chain, which leaves the preconditioner
 *        unchanged to rounding; off: exactly one chain per partition),
 *       "profile" (0|1: record HIP events around the sweep launches),
 *       "overlap_exchange" ("on"|"off", default on: several ranks run the tip exchange and the rank-boundary interfaces on
 *        a second stream beside the local coupling work; same bits either way),
 *       "small_coupling_kmax" (0..8, default 3: half-bandwidths up to this take the one-launch narrow-band coupling step
 *        on one rank (K = 2, 3: behind the one-launch scan only); same preconditioner),
 *       "gmres_cgs_refinement_type" ("refine_never"|"refine_ifneeded"|"refine_always": the Gram-Schmidt refinement of
 *        spike_gmres, names and default (never) of PETSc's -ksp_gmres_cgs_refinement_type),
 *       "twist" ("auto"|"off": two-ended factorisation of chain pairs where stored spikes apply; same preconditioner),
 *       "spike_tol" default 1e-13 since round 3, "spike_fp32" ("auto"|"off": far part of the stored spikes in fp32),
 *       "iface_form" ("matrix"|"staged": one-stage or three-stage interface solves; same result to rounding),
 *       "correct_threads" (64|128|256: workgroup size of the spike correction, measurement option),
 *       "narrow_scan_kmax" (1..3, default 3: half-bandwidths up to this are solved by the wavefront scan -- no tiles, the
 *        algorithmic (2K+3)*8 bytes per row -- above it by the tile sweeps; same preconditioner),
 *       "sweep_autotune" ("on"|"off", default on: K > 64 with chains of >= 8192 rows: the first setup of a shape on a handle times
 *        the candidate sweep shapes (how the diagonals of a chain are dealt to waves, bundles in flight) over the real factors,
 *        ~12 ms once, and keeps the fastest -- which one that is depends on the device; off: the base shape.  Same
 *        preconditioner; the cross-wave summation order, hence the last bits, may differ between two choices),
 *       "workspace_cache" ("on"|"off", default on: the device blocks of a handle are recycled by size between its setups --
 *        a refactorisation on the same handle makes no hipMalloc / hipFree calls; idle blocks a whole setup did not use are
 *        released at its end, spike_reset / spike_destroy release everything; off: every setup allocates and frees),
 *       "narrow_scan_rows" (1|4, default 4: K = 1 through the one-row-per-lane scan of round 2 or the four-rows-per-lane
 *        kernels; K = 2, 3 always four)      */
int spike_set_option(spike_handle h, const char *key, const char *value);
/* HIP stream (hipStream_t) all device work of this handle is issued on; NULL = default stream */
int spike_set_stream(spike_handle h, void *hip_stream);

/* ---- multi-GPU (one process per GPU; optional) ------------------------------------ */
int spike_comm_unique_id(char id[SPIKE_UNIQUE_ID_BYTES]); /* rank 0 creates, host code broadcasts */
/* Joins nranks handles.  Must precede setup.  row0/n_global describe this rank's row block. */
int spike_comm_init(spike_handle h, int nranks, int rank, const char id[SPIKE_UNIQUE_ID_BYTES]);
/* Test transport: the nranks handles are driven by nranks host THREADS of one process that share one
 * GPU (RCCL refuses two ranks on one device).  Same algorithm, buffers and call order as the RCCL
 * transport; bytes move by hipMemcpy behind a host barrier.  `group` names the communicator.       */
int spike_comm_init_local(spike_handle h, int nranks, int rank, int group);

/* ---- setup ------------------------------------------------------------------------- */
/* Factor the local rows [row0,row0+n_local) of an n_global system.  The local band holds the
 * local rows only (ld >= n_local) but with GLOBAL column meaning, so entries that couple to a
 * neighbouring rank sit in their natural band slots.  Single GPU: row0=0, n_local=n_global.
 * on_device != 0: band is a device pointer.                                               */
int spike_setup_band(spike_handle h, int64_t n_global, int64_t row0, int64_t n_local, int K,
                     const double *band, int64_t ld, int on_device);

/* CSR entry (single rank): band extraction with the reference's rule, then spike_setup_band.
 * kmax/frac as PCBANDED's -pc_banded_kmax/-pc_banded_frac (defaults 50 / 0.95).
 * ia/ja are 0-based int64 host arrays.  An assembled AIJ matrix holds a column at most once per row (MatGetRow at
 * matbanded.c:40); if a (row, column) pair is repeated anyway, the band keeps the LAST stored value -- the INSERT_VALUES
 * semantics of the reference's MatSetValues (matbanded.c:98) -- deterministically (one thread per row, storage order);
 * the half-bandwidth rule's weights count every stored entry.                                                  */
int spike_setup_csr(spike_handle h, int64_t n, const int64_t *ia, const int64_t *ja, const double *a,
                    int kmax, double frac, int *k_out, double *frac_out);

/* CSR entry for a ROW-BLOCK-DISTRIBUTED matrix, one rank per GPU (the layout MatCreateSubMatrixBanded is written for:
 * MatGetOwnershipRange src/matbanded.c:36, diagonal/off-diagonal split :74-75): this rank passes its rows
 * [row0, row0+n_local) with GLOBAL 0-based column indices (ia has n_local+1 entries starting at 0).  Collective over the
 * ranks joined by spike_comm_init: the per-diagonal weights of the reference's rule are summed per rank in row order and
 * combined in rank order, so every rank chooses the same k.  One rank: identical to spike_setup_csr.            */
int spike_setup_csr_dist(spike_handle h, int64_t n_global, int64_t row0, int64_t n_local, const int64_t *ia,
                         const int64_t *ja, const double *a, int kmax, double frac, int *k_out, double *frac_out);

/* The same two entry points for 32-bit index arrays -- PETSc's DEFAULT build has a 32-bit PetscInt (the glue in
 * examples/petsc/ selects on sizeof(PetscInt)).  row0 / n_global stay 64-bit: several ranks may hold > 2^31 rows. */
int spike_setup_csr32(spike_handle h, int64_t n, const int32_t *ia, const int32_t *ja, const double *a,
                      int kmax, double frac, int *k_out, double *frac_out);
int spike_setup_csr_dist32(spike_handle h, int64_t n_global, int64_t row0, int64_t n_local, const int32_t *ia,
                           const int32_t *ja, const double *a, int kmax, double frac, int *k_out, double *frac_out);

/* ---- apply --------------------------------------------------------------------------- */
/* y = M^{-1} x on the local rows.  x != y.  on_device != 0: device pointers, asynchronous on
 * the handle's stream; otherwise host pointers, synchronous.                               */
int spike_apply(spike_handle h, const double *x, double *y, int on_device);

/* ---- Krylov caller ---------------------------------------------------------------------- */
/* Left-preconditioned GMRES(restart) on the band given at setup (keep_band=1) with this handle
 * as preconditioner (use_pc=0: unpreconditioned).  b,x device pointers of n_local doubles; x
 * holds the initial guess on entry.  Convergence: preconditioned residual <= rtol * initial.
 * restart <= 64.  Returns 0 converged, 1 hit maxit, <0 error.                              */
int spike_gmres(spike_handle h, const double *b, double *x, int restart, double rtol, int maxit, int use_pc,
                int *iters, double *rnorm, double *solve_ms);

/* Optional CSR operator for spike_gmres: the reference preconditions the FULL matrix A with its band
 * (KSPSetOperators(ksp,A,B), src/testbed2.c:126).  Host 0-based int64 CSR, copied to the device; replaces the
 * kept band as the operator until spike_clear_operator.  May be set on a handle without factors
 * (then spike_gmres needs use_pc = 0).  Single rank.                                              */
int spike_set_operator_csr(spike_handle h, int64_t n, const int64_t *ia, const int64_t *ja, const double *a);
int spike_clear_operator(spike_handle h);
/* A BANDED operator different from the matrix given at setup (same n_local, K, row block; device, diagonal-major;
 * copied at once, NULL clears): the preconditioner-from-a-nearby-matrix case.  Out-of-range corner slots must be 0. */
int spike_set_operator_band(spike_handle h, const double *band_dev, int64_t ld);
/* y = Op x with the operator spike_gmres uses (CSR operator > banded operator > band kept at setup); device pointers */
int spike_operator_matvec(spike_handle h, const double *x, double *y);

/* raw device memory for C hosts that have no HIP headers (the host mirror uses these) */
int spike_dev_malloc(void **p, size_t bytes);
int spike_dev_free(void *p);
int spike_dev_upload(void *dev_dst, const void *host_src, size_t bytes);
int spike_dev_download(void *host_dst, const void *dev_src, size_t bytes);

/* host steps of spike_setup_csr, usable on their own: the reference's half-bandwidth rule
 * (src/matbanded.c:38-56,104-105) and the CSR -> diagonal-major band conversion                     */
int spike_csr_band_k(int64_t n, const int64_t *ia, const int64_t *ja, const double *a, int kmax, double frac,
                     int *k_out, double *frac_out);
int spike_csr_to_band(int64_t n, const int64_t *ia, const int64_t *ja, const double *a, int K, double *band,
                      int64_t ld);
/* pieces of the distributed rule: one rank's weights w[0..kmax) and its part of ||A||_1 (src/matbanded.c:38-49), and
 * the stopping rule on summed weights (:53-56, 104-105) */
int spike_csr_band_weights(int64_t n_global, int64_t row0, int64_t n_local, const int64_t *ia, const int64_t *ja,
                           const double *a, int kmax, double *w, double *normA);
int spike_band_rule(int64_t n, const double *w, double normA, int kmax, double frac, int *k_out, double *frac_out);
/* 32-bit index variants of the rule (MatCreateSubMatrixBanded in the PETSc glue of a default PETSc build) */
int spike_csr_band_k32(int64_t n, const int32_t *ia, const int32_t *ja, const double *a, int kmax, double frac,
                       int *k_out, double *frac_out);
int spike_csr_band_weights32(int64_t n_global, int64_t row0, int64_t n_local, const int32_t *ia, const int32_t *ja,
                             const double *a, int kmax, double *w, double *normA);

/* ---- helpers (device) ----------------------------------------------------------------------- */
/* y = A x with the band kept at setup (device pointers, local rows; halo via RCCL when nranks>1) */
int spike_band_matvec(spike_handle h, const double *x, double *y);
/* the synthetic system of SURVEY.md 8d, generated in place on the device (bit-identical to the
 * CPU oracle's generator): rows [row0,row0+nrows) of an n_global band, ld >= nrows            */
int spike_gen_band(void *hip_stream, int64_t n_global, int K, uint64_t seed, double delta, int64_t row0,
                   int64_t nrows, double *band_dev, int64_t ld);

/* ---- introspection --------------------------------------------------------------------------- */
/* the CHAIN count setup picks for partitions = 0 (host logic, needs no device); <0: K unsupported (> 512).  Where the twisted
 * factorisation applies the chains are paired: spike_info.P_local is then half of it, spike_info.chains_local all of it */
int spike_auto_partitions(int K, int64_t n_local);
int spike_get_info(spike_handle h, spike_info *info);
int spike_view(spike_handle h, char *buf, size_t buflen);
/* test hooks: copy the K x K spike tips of the local interfaces to host (row-major) */
int spike_get_tips(spike_handle h, double *Vb, double *Wt);
/* timing hook: average device time (ms) of the sweep kernels of the last spike_apply, from
 * HIP events recorded on the handle's stream (valid after spike_set_option(h,"profile","1")) */
int spike_last_sweep_ms(spike_handle h, double *ms_total, int *nlaunches);

/* measurement hook: GB/s of a pure read stream over the handle's packed factors (same 16-byte-per-lane
 * non-temporal access shape as the sweeps), i.e. the read ceiling this device delivers                     */
int spike_measure_read_bw(spike_handle h, int reps, double *gbps);

/* ---- reordering front-end on the device (csrc/spike_reorder.hip) ---------------------------------------------
 * Integer / byte work with results identical to the host loops of libspike_petsc_host (MatPermute, VecPermute, spike_awbm).
 * No device: SPIKE_ERR_HIP (the caller keeps its host loop).
 *   spike_permute_csr  = MatPermute(M, rorder, corder, &PM), src/kspreorder.c:20-22: row i of B is row rowp[i] of A, column c
 *                        becomes the position of c in colp; rows of B sorted by column.  Host arrays in and out (ib: n+1,
 *                        jb / b: ia[n] entries); SPIKE_ERR_ARG if rowp / colp are not permutations.
 *   spike_permute_vec  = VecPermute(x, is, inverse), src/kspreorder.c:122-127, out of place: y[i] = x[idx[i]]
 *                        (inverse: y[idx[i]] = x[i]); on_device != 0: x, y, idx are device pointers.
 *   spike_awbm_device  = MatGetOrdering_AWBM, src/petsc_mat_awbm.c:42-225, with its two greedy phases (:98-112, :143-153) as
 *                        a parallel fixed-point iteration on the device that reproduces the sequential greedy exactly; the
 *                        one-step augmentations and the default fill (:115-140, :156-193) on the host.  perm[match[c]] = c;
 *                        rounds[2] (optional): fixed-point rounds of the two phases.                                         */
int spike_permute_csr(int64_t n, const int64_t *ia, const int64_t *ja, const double *a, const int64_t *rowp, const int64_t *colp,
                      int64_t *ib, int64_t *jb, double *b);
int spike_permute_vec(int64_t n, const int64_t *idx, int inverse, const double *x, double *y, int on_device);
int spike_awbm_device(int64_t n, const int64_t *ia, const int64_t *ja, const double *a, int64_t *perm, int *rounds);

/* ---- device half of the Fiedler ordering (reference slot MatGetOrdering_Fiedler, src/petsc_mat_fiedler.c:11-58) -------------
 * The ordering itself is host code (libspike_petsc_host: spike_fiedler_order_ex); its floating-point part, the LOBPCG
 * refinement of a multilevel level, can run here.  These calls hold one level's vectors on the device (ids 0 x, 1 Lx, 2 w,
 * 3 Lw, 4 p, 5 Lp, 6 the constant 1) and execute the statements of fiedler.c:refine_core() in the same IEEE operations and
 * the same reduction order as the host implementation: the resulting permutation is bit-identical.                     */
typedef struct spike_fd_ctx spike_fd_ctx;
int spike_device_count(void); /* HIP devices visible (0: none; never an error) */
int spike_fd_create(int64_t n, const int64_t *xadj, const int64_t *adj, const double *w, const double *deg,
                    const double *x0, spike_fd_ctx **out);
int spike_fd_destroy(spike_fd_ctx *c);
int spike_fd_dots(spike_fd_ctx *c, int nd, const int *ia, const int *ib, double *sums); /* nd <= 6 dot products */
int spike_fd_lap(spike_fd_ctx *c, int src, int dst);                                    /* v[dst] = L v[src]   */
int spike_fd_shift(spike_fd_ctx *c, int vec, double m);
int spike_fd_div(spike_fd_ctx *c, double s, int y, int y2);
/* a level's whole LOBPCG loop: maxit x (six vector kernels, each followed by a one-thread scalar step that is the SAME C
 * code the host loop runs, csrc/host/fiedler_steer.h), the iteration state in device memory, launched without a host
 * round trip; steps after the stopping test fired are no-ops.  rho = x.Lx of the start vector; its = iterations run. */
int spike_fd_refine(spike_fd_ctx *c, double dmax, double rho, int maxit, int *its);
int spike_fd_fill_alternating(spike_fd_ctx *c);
int spike_fd_download_x(spike_fd_ctx *c, double *x);

#ifdef __cplusplus
}
#endif
#endif /* SPIKE_MI355_H */
