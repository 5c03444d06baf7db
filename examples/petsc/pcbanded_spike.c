/*
 * pcbanded_spike.c -- PETSc glue: PCBANDED with libspike_mi355.so as its inner preconditioner.
 *
 * NOT BUILT IN THIS REPOSITORY (PETSc is installed on neither box); it is the file a maintainer of
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
     GPU.  Needs --with-64-bit-indices (PetscInt = int64_t); otherwise widen ia/ja first. */
  PetscCall(MatGetRowIJ(Aloc, 0, PETSC_FALSE, PETSC_FALSE, &n, &ia, &ja, &done));
  PetscCheck(done && n == rend - rstart, PetscObjectComm((PetscObject)pc), PETSC_ERR_SUP, "MatGetRowIJ failed");
  PetscCall(MatSeqAIJGetArray(Aloc, &a));
  b->k = b->kmax; b->f = b->frac; /* :172-173 */
  SPIKE_CHK(pc, b, spike_setup_csr_dist(b->spike, (int64_t)N, (int64_t)rstart, (int64_t)n, (const int64_t *)ia, (const int64_t *)ja, a,
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
