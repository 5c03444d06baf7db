/*
 * pcbanded_spike.c -- PETSc glue: PCBANDED with libspike_mi355.so as its inner preconditioner.
 *
 * NOT BUILT IN THIS REPOSITORY (PETSc is installed on neither box; tests/test_petsc_glue_syntax.py runs a
 * `gcc -fsyntax-only` TYPO CHECK of it against the declarations-only header examples/petsc/syntax_check/, for a 32-bit and
 * a 64-bit PetscInt -- a typo check, not a build and not an oracle); it is the file a maintainer of
 * spikegpu/spike-petsc drops next to src/matbanded.c.  It replaces the embedded `PC pc` of PC_Banded
 * (/root/reference/src/matbanded.c:111-116) by a spike_handle and maps the ops table 1:1 onto
 * include/spike_mi355.h.  Written against the PETSc >= 3.19 API (PetscCall, PETSC_SUCCESS); line numbers
 * in comments refer to the reference's src/matbanded.c.
 *
 *   link: -I<repo>/include -L<repo>/spike-petsc_amd -lspike_mi355
 *   run : -pc_type banded -pc_banded_kmax 128 -banded_spike_partitions 0 -banded_spike_variant coupled
 *         mpiexec -n 8 (one rank per GPU, MATMPIAIJ): the row blocks go through spike_setup_csr_dist, the rank-boundary
 *         interface systems are assembled by the library's RCCL all-gather.  The KSP-level hook is kspreorder_spike.c.
 */
#include <petsc/private/pcimpl.h>
#include <petscdevice_hip.h>
#include <spike_mi355.h>

/* PetscInt is 32-bit in PETSc's DEFAULT build and 64-bit with --with-64-bit-indices: the library has an entry point for
   each, selected here at compile time.  (Never cast a PetscInt array to int64_t*: with 32-bit indices the library would
   read past ia/ja.) */
#if defined(PETSC_USE_64BIT_INDICES)
typedef int64_t SpikeIdx;
#define spike_setup_csr_distX    spike_setup_csr_dist
#define spike_csr_band_weightsX  spike_csr_band_weights
#else
typedef int32_t SpikeIdx;
#define spike_setup_csr_distX    spike_setup_csr_dist32
#define spike_csr_band_weightsX  spike_csr_band_weights32
#endif
typedef char SpikeIdxMatchesPetscInt[(sizeof(SpikeIdx) == sizeof(PetscInt)) ? 1 : -1];

/* MatCreateSubMatrixBanded (/root/reference/src/matbanded.h:5, src/matbanded.c:22-107; called directly by
   src/testbed.c:286-296): the half-bandwidth rule on the library's host code (per-rank weights in row order, combined in
   RANK order so that every rank chooses the same k -- the reference's own weight Vec is indexed locally, :45, and is only
   right on one rank), then the copy of the |c - r| <= k entries into a new AIJ matrix with PETSc calls (:65-99).
   In: *kmax, *frac as the reference; out: *kmax = k (:104), *frac = achieved fraction (:105), *B. */
PetscErrorCode MatCreateSubMatrixBanded(Mat A, PetscInt *kmax, PetscReal *frac, Mat *B)
{
  MPI_Comm        comm = PetscObjectComm((PetscObject)A);
  Mat             Aloc = A;
  PetscBool       ismpi, done;
  PetscInt        N, n, rstart, rend, r, c, ncols, nb, *dnnz, *onnz, *bcols;
  const PetscInt *ia, *ja, *cols;
  PetscScalar    *a, *bvals;
  const PetscScalar *vals;
  PetscMPIInt     size, q;
  double         *part, *all, *w, normA = 0.0, f = 0.0;
  int             k = 0, km = (int)*kmax, d;

  PetscFunctionBegin;
  PetscCallMPI(MPI_Comm_size(comm, &size));
  PetscCall(MatGetSize(A, &N, NULL));
  PetscCall(MatGetOwnershipRange(A, &rstart, &rend)); /* :36 */
  PetscCall(PetscObjectTypeCompare((PetscObject)A, MATMPIAIJ, &ismpi));
  if (ismpi) PetscCall(MatMPIAIJGetLocalMat(A, MAT_INITIAL_MATRIX, &Aloc));
  PetscCall(MatGetRowIJ(Aloc, 0, PETSC_FALSE, PETSC_FALSE, &n, &ia, &ja, &done));
  PetscCheck(done && n == rend - rstart, comm, PETSC_ERR_SUP, "MatGetRowIJ failed");
  PetscCall(MatSeqAIJGetArray(Aloc, &a));
  /* passes 1 and 2 (:38-56): [w[0..kmax) | normA] per rank, gathered, added in rank order */
  PetscCall(PetscCalloc3(km + 1, &part, (size_t)(km + 1) * size, &all, km > 0 ? km : 1, &w));
  PetscCheck(spike_csr_band_weightsX((int64_t)N, (int64_t)rstart, (int64_t)n, (const SpikeIdx *)ia, (const SpikeIdx *)ja, a, km, part, &part[km]) == 0,
             comm, PETSC_ERR_ARG_OUTOFRANGE, "MatCreateSubMatrixBanded: column index out of range");
  PetscCallMPI(MPI_Allgather(part, km + 1, MPI_DOUBLE, all, km + 1, MPI_DOUBLE, comm));
  for (q = 0; q < size; ++q) {
    for (d = 0; d < km; ++d) w[d] += all[(size_t)q * (km + 1) + d];
    normA += all[(size_t)q * (km + 1) + km];
  }
  PetscCheck(spike_band_rule((int64_t)N, w, normA, km, (double)*frac, &k, &f) == 0, comm, PETSC_ERR_LIB, "spike_band_rule failed");
  PetscCall(PetscFree3(part, all, w));
  PetscCall(MatSeqAIJRestoreArray(Aloc, &a));
  PetscCall(MatRestoreRowIJ(Aloc, 0, PETSC_FALSE, PETSC_FALSE, &n, &ia, &ja, &done));
  if (ismpi) PetscCall(MatDestroy(&Aloc));
  /* pass 3 (:65-99): count, preallocate, copy */
  PetscCall(PetscCalloc2(n, &dnnz, n, &onnz));
  for (r = rstart; r < rend; ++r) {
    PetscCall(MatGetRow(A, r, &ncols, &cols, NULL));
    for (c = 0; c < ncols; ++c)
      if (PetscAbsInt(cols[c] - r) <= (PetscInt)k) { if (cols[c] >= rstart && cols[c] < rend) ++dnnz[r - rstart]; else ++onnz[r - rstart]; } /* :73-75 */
    PetscCall(MatRestoreRow(A, r, &ncols, &cols, NULL));
  }
  PetscCall(MatCreate(comm, B));
  PetscCall(MatSetSizes(*B, n, n, N, N));
  PetscCall(MatSetType(*B, MATAIJ));
  PetscCall(MatXAIJSetPreallocation(*B, 1, dnnz, onnz, NULL, NULL)); /* :81 */
  PetscCall(PetscMalloc2(2 * (size_t)k + 1, &bcols, 2 * (size_t)k + 1, &bvals));
  for (r = rstart; r < rend; ++r) {
    PetscCall(MatGetRow(A, r, &ncols, &cols, &vals));
    for (nb = 0, c = 0; c < ncols; ++c)
      if (PetscAbsInt(cols[c] - r) <= (PetscInt)k) {
        PetscCheck(nb < 2 * (PetscInt)k + 1, comm, PETSC_ERR_PLIB, "row %" PetscInt_FMT " holds more than 2k+1 band entries", r); /* :95 */
        bcols[nb] = cols[c]; bvals[nb] = vals[c]; ++nb;
      }
    PetscCall(MatSetValues(*B, 1, &r, nb, bcols, bvals, INSERT_VALUES)); /* :98 */
    PetscCall(MatRestoreRow(A, r, &ncols, &cols, &vals));
  }
  PetscCall(PetscFree2(bcols, bvals));
  PetscCall(PetscFree2(dnnz, onnz));
  PetscCall(MatAssemblyBegin(*B, MAT_FINAL_ASSEMBLY));
  PetscCall(MatAssemblyEnd(*B, MAT_FINAL_ASSEMBLY));
  *kmax = (PetscInt)k; /* :104 */
  *frac = (PetscReal)f; /* :105 */
  PetscFunctionReturn(PETSC_SUCCESS);
}

typedef struct {
  PetscInt     kmax, k;   /* :112 */
  PetscReal    frac, f;   /* :113 */
  spike_handle spike;     /* replaces Mat B (:114) and PC pc (:115): the band lives in the library */
  PetscBool    comm_done;                 /* spike_comm_init done for this PC's communicator */
  PetscBool    refactor_on_same_pattern;  /* -pc_banded_refactor: values changed, pattern did not */
} PC_Banded;

#define SPIKE_CHK(pc, b, call) \
  do { if ((call) != 0) SETERRQ(PetscObjectComm((PetscObject)(pc)), PETSC_ERR_LIB, "libspike_mi355: %s", spike_last_error((b)->spike)); } while (0)

static PetscErrorCode PCReset_Banded(PC pc) /* :120-129 */
{
  PC_Banded *b = (PC_Banded *)pc->data;
  PetscFunctionBegin;
  if (b->spike) SPIKE_CHK(pc, b, spike_reset(b->spike));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCDestroy_Banded(PC pc) /* :133-145 */
{
  PC_Banded *b = (PC_Banded *)pc->data;
  PetscFunctionBegin;
  PetscCall(PCReset_Banded(pc));
  PetscCall(PetscObjectComposeFunction((PetscObject)pc, "PCBandedSetMaxHalfBandwidth_C", NULL));
  PetscCall(PetscObjectComposeFunction((PetscObject)pc, "PCBandedSetNormFraction_C", NULL));
  if (b->spike) (void)spike_destroy(b->spike);
  PetscCall(PetscFree(pc->data));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCSetFromOptions_Banded(PC pc, PetscOptionItems *PetscOptionsObject) /* :149-161 */
{
  PC_Banded *b = (PC_Banded *)pc->data;
  PetscInt   P = 0;
  char       variant[32] = "coupled";
  PetscReal  boost = 1e-10;
  PetscBool  set;
  char       num[64];
  PetscFunctionBegin;
  PetscOptionsHeadBegin(PetscOptionsObject, "Banded options");
  PetscCall(PetscOptionsInt("-pc_banded_kmax", "Maximum half-bandwidth", "PCBandedSetMaxHalfBandwidth", b->kmax, &b->kmax, NULL));
  PetscCall(PetscOptionsReal("-pc_banded_frac", "Fraction of the 1-norm to keep", "PCBandedSetNormFraction", b->frac, &b->frac, NULL));
  PetscCall(PetscOptionsBool("-pc_banded_refactor", "Re-extract and re-factor when only the values changed", NULL, b->refactor_on_same_pattern, &b->refactor_on_same_pattern, NULL));
  /* the inner PC's options (prefix banded_, :281) become the engine's options */
  PetscCall(PetscOptionsInt("-banded_spike_partitions", "SPIKE partitions (0 = automatic)", NULL, P, &P, &set));
  if (set) { PetscCall(PetscSNPrintf(num, sizeof num, "%" PetscInt_FMT, P)); SPIKE_CHK(pc, b, spike_set_option(b->spike, "partitions", num)); }
  PetscCall(PetscOptionsString("-banded_spike_variant", "coupled | decoupled", NULL, variant, variant, sizeof variant, &set));
  if (set) SPIKE_CHK(pc, b, spike_set_option(b->spike, "variant", variant));
  PetscCall(PetscOptionsReal("-banded_spike_boost", "pivot boost, relative to max|diag|", NULL, boost, &boost, &set));
  if (set) { PetscCall(PetscSNPrintf(num, sizeof num, "%g", (double)boost)); SPIKE_CHK(pc, b, spike_set_option(b->spike, "boost", num)); }
  PetscOptionsHeadEnd();
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* One rank per GPU: join the ranks of the PC's communicator once (the unique id travels by MPI, the data path is RCCL). */
static PetscErrorCode PCBandedJoinRanks(PC pc)
{
  PC_Banded  *b = (PC_Banded *)pc->data;
  MPI_Comm    comm = PetscObjectComm((PetscObject)pc);
  PetscMPIInt size, rank;
  char        id[SPIKE_UNIQUE_ID_BYTES];

  PetscFunctionBegin;
  if (b->comm_done) PetscFunctionReturn(PETSC_SUCCESS);
  PetscCallMPI(MPI_Comm_size(comm, &size));
  PetscCallMPI(MPI_Comm_rank(comm, &rank));
  if (size > 1) {
    if (rank == 0) PetscCheck(spike_comm_unique_id(id) == 0, comm, PETSC_ERR_LIB, "spike_comm_unique_id: cannot load librccl");
    PetscCallMPI(MPI_Bcast(id, SPIKE_UNIQUE_ID_BYTES, MPI_BYTE, 0, comm));
    SPIKE_CHK(pc, b, spike_comm_init(b->spike, (int)size, (int)rank, id)); /* must precede setup */
  }
  b->comm_done = PETSC_TRUE;
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCSetUp_Banded(PC pc) /* :165-180 */
{
  PC_Banded      *b = (PC_Banded *)pc->data;
  Mat             Aloc = pc->pmat;   /* the rows this rank owns, as one SeqAIJ with GLOBAL column indices */
  const PetscInt *ia, *ja;
  PetscScalar    *a;
  PetscInt        n, N, rstart, rend;
  PetscBool       done, ismpi;
  int             k;
  double          f;

  PetscFunctionBegin;
  /* The reference extracts the band on the first call only (pc->setupcalled == 0, :171) and re-runs the inner
     PCSetUp every time (:178).  Here extraction + factorisation are one library call, made when the operator is new:
     first call, or PETSc flagged a changed matrix (pc->flag != SAME_PRECONDITIONER path reaches setup again). */
  if (pc->setupcalled && pc->flag == SAME_NONZERO_PATTERN && !b->refactor_on_same_pattern) PetscFunctionReturn(PETSC_SUCCESS);
  PetscCall(PCBandedJoinRanks(pc));
  PetscCall(MatGetSize(pc->pmat, &N, NULL));
  PetscCall(MatGetOwnershipRange(pc->pmat, &rstart, &rend)); /* :36 */
  PetscCall(PetscObjectTypeCompare((PetscObject)pc->pmat, MATMPIAIJ, &ismpi));
  if (ismpi) PetscCall(MatMPIAIJGetLocalMat(pc->pmat, MAT_INITIAL_MATRIX, &Aloc)); /* diagonal + off-diagonal block merged, global columns (:74-75) */
  /* MatCreateSubMatrixBanded (:22-107) + PCSetUp(b->pc) (:178) in one call: the library applies the reference's
     half-bandwidth rule to the CSR arrays (per-rank sums combined in rank order) and factors the extracted band on the
     GPU.  The entry point matches sizeof(PetscInt) (SpikeIdx above). */
  PetscCall(MatGetRowIJ(Aloc, 0, PETSC_FALSE, PETSC_FALSE, &n, &ia, &ja, &done));
  PetscCheck(done && n == rend - rstart, PetscObjectComm((PetscObject)pc), PETSC_ERR_SUP, "MatGetRowIJ failed");
  PetscCall(MatSeqAIJGetArray(Aloc, &a));
  b->k = b->kmax; b->f = b->frac; /* :172-173 */
  SPIKE_CHK(pc, b, spike_setup_csr_distX(b->spike, (int64_t)N, (int64_t)rstart, (int64_t)n, (const SpikeIdx *)ia, (const SpikeIdx *)ja, a,
                                         (int)b->kmax, (double)b->frac, &k, &f));
  b->k = k; b->f = f;
  PetscCall(PetscInfo(pc, "PCBANDED: half-bandwidth: %d norm fraction: %g\n", k, f)); /* :175 */
  PetscCall(MatSeqAIJRestoreArray(Aloc, &a));
  PetscCall(MatRestoreRowIJ(Aloc, 0, PETSC_FALSE, PETSC_FALSE, &n, &ia, &ja, &done));
  if (ismpi) PetscCall(MatDestroy(&Aloc));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCApply_Banded(PC pc, Vec x, Vec y) /* :184-192 -- the metric's "PCApply" */
{
  PC_Banded         *b = (PC_Banded *)pc->data;
  const PetscScalar *xa;
  PetscScalar       *ya;
  PetscFunctionBegin;
  PetscCall(VecHIPGetArrayRead(x, &xa));
  PetscCall(VecHIPGetArrayWrite(y, &ya));
  SPIKE_CHK(pc, b, spike_apply(b->spike, xa, ya, /*on_device=*/1));
  PetscCall(VecHIPRestoreArrayWrite(y, &ya));
  PetscCall(VecHIPRestoreArrayRead(x, &xa));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCView_Banded(PC pc, PetscViewer viewer) /* :196-211 */
{
  PC_Banded *b = (PC_Banded *)pc->data;
  PetscBool  isascii;
  char       buf[1024];
  PetscFunctionBegin;
  PetscCall(PetscObjectTypeCompare((PetscObject)viewer, PETSCVIEWERASCII, &isascii));
  if (isascii) {
    PetscCall(PetscViewerASCIIPrintf(viewer, "  Banded: half-bandwidth: %" PetscInt_FMT " norm fraction: %g\n", b->k, (double)b->f));
    if (spike_view(b->spike, buf, sizeof buf) == 0) PetscCall(PetscViewerASCIIPrintf(viewer, "%s", buf));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode PCBandedSetMaxHalfBandwidth_Banded(PC pc, PetscInt kmax) { ((PC_Banded *)pc->data)->kmax = kmax; return PETSC_SUCCESS; } /* :287-294 */
static PetscErrorCode PCBandedSetNormFraction_Banded(PC pc, PetscReal frac) { ((PC_Banded *)pc->data)->frac = frac; return PETSC_SUCCESS; }     /* :317-324 */

PETSC_EXTERN PetscErrorCode PCCreate_Banded(PC pc) /* :251-283 */
{
  PC_Banded *b;
  PetscFunctionBegin;
  PetscCall(PetscNew(&b));
  pc->data = (void *)b;
  b->kmax = 50;   /* :261 */
  b->refactor_on_same_pattern = PETSC_TRUE;
  b->frac = 0.95; /* :262 */
  PetscCheck(spike_create(&b->spike) == 0, PetscObjectComm((PetscObject)pc), PETSC_ERR_LIB, "spike_create: no HIP device");
  pc->ops->apply               = PCApply_Banded;
  pc->ops->applytranspose      = NULL; /* as the reference, :265-268 */
  pc->ops->setup               = PCSetUp_Banded;
  pc->ops->reset               = PCReset_Banded;
  pc->ops->destroy             = PCDestroy_Banded;
  pc->ops->setfromoptions      = PCSetFromOptions_Banded;
  pc->ops->view                = PCView_Banded;
  pc->ops->applyrichardson     = NULL;
  pc->ops->applysymmetricleft  = NULL;
  pc->ops->applysymmetricright = NULL;
  PetscCall(PetscObjectComposeFunction((PetscObject)pc, "PCBandedSetMaxHalfBandwidth_C", PCBandedSetMaxHalfBandwidth_Banded));
  PetscCall(PetscObjectComposeFunction((PetscObject)pc, "PCBandedSetNormFraction_C", PCBandedSetNormFraction_Banded));
  PetscFunctionReturn(PETSC_SUCCESS);
}
