/*
 * kspreorder_spike.c -- PETSc glue: KSPREORDER (/root/reference/src/kspreorder.c) for a current PETSc, with the
 * reordering kernels of this repository behind MatGetOrdering.
 *
 * NOT BUILT IN THIS REPOSITORY (PETSc is installed on neither box; tests/test_petsc_glue_syntax.py runs a
 * `gcc -fsyntax-only` TYPO CHECK of it against the declarations-only header examples/petsc/syntax_check/, for a 32-bit and
 * a 64-bit PetscInt -- a typo check, not a build and not an oracle); it is the file a maintainer of
 * spikegpu/spike-petsc drops next to src/kspreorder.c, beside pcbanded_spike.c.  The reference's file mixes PETSc 3.4
 * and 3.5 calls and compiles against no release; this one is written against PETSc >= 3.19
 * (PetscCall, 3-argument KSPSetOperators, PetscOptionsHeadBegin).  Structure, option names, prefixes and the in-place
 * VecPermute bracket are the reference's; line numbers in comments refer to src/kspreorder.c.
 *
 *   link: -I<repo>/include -L<repo>/spike-petsc_amd -lspike_petsc_host -lspike_mi355
 *   run : -ksp_type reorder -mat_ordering_type wbm -reorder_ksp_type gmres -reorder_pc_type banded
 */
#include <petsc/private/kspimpl.h>
#include <petscmat.h>
#include <spike_orderings.h> /* spike_mc64_job5[_i32], spike_awbm[_i32], spike_fiedler_order_ex/_i32: plain C, no PETSc types */

/* PetscInt is 32-bit in PETSc's DEFAULT build, 64-bit with --with-64-bit-indices: one entry point of the library for each.
   (Never cast a PetscInt array to int64_t*: with 32-bit indices the kernels would read past ia/ja and write 8-byte values
   into the 4-byte perm/order arrays.) */
#if defined(PETSC_USE_64BIT_INDICES)
typedef int64_t SpikeIdx;
#define spike_mc64_job5X(n, ia, ja, a, perm, u, v, num) spike_mc64_job5(n, ia, ja, a, perm, u, v, num)
#define spike_awbmX(n, ia, ja, a, perm)                 spike_awbm(n, ia, ja, a, perm, NULL, NULL)
#define spike_fiedler_orderX(n, ia, ja, a, order)       spike_fiedler_order_ex(n, ia, ja, a, order, NULL, 1)
#else
typedef int32_t SpikeIdx;
#define spike_mc64_job5X(n, ia, ja, a, perm, u, v, num) spike_mc64_job5_i32(n, ia, ja, a, perm, u, v, num)
#define spike_awbmX(n, ia, ja, a, perm)                 spike_awbm_i32(n, ia, ja, a, perm, NULL, NULL)
#define spike_fiedler_orderX(n, ia, ja, a, order)       spike_fiedler_order_i32(n, ia, ja, a, order, NULL, 1)
#endif
typedef char SpikeIdxMatchesPetscInt[(sizeof(SpikeIdx) == sizeof(PetscInt)) ? 1 : -1];

typedef struct {
  KSP  ksp;            /* :4 the embedded KSP */
  char ordertype[256]; /* :5 */
  IS   rorder, corder; /* :6 */
} KSP_Reorder;

/* ---- orderings (src/petsc_mat_wbm.c:13-61, src/petsc_mat_fiedler.c:11-58, src/petsc_mat_awbm.c) on the library's kernels ---- */
static PetscErrorCode SpikeGetCSR(Mat mat, PetscInt *n, const PetscInt **ia, const PetscInt **ja, PetscScalar **a)
{
  PetscBool done;
  PetscFunctionBegin;
  /* 0-based, unsymmetrised: index and value arrays stay aligned (the reference asks for the symmetrised structure
     with the raw value array, src/petsc_mat_wbm.c:29,33 -- see SURVEY.md appendix A) */
  PetscCall(MatGetRowIJ(mat, 0, PETSC_FALSE, PETSC_FALSE, n, ia, ja, &done));
  PetscCheck(done, PetscObjectComm((PetscObject)mat), PETSC_ERR_SUP, "Cannot get rows for matrix type %s", ((PetscObject)mat)->type_name);
  PetscCall(MatSeqAIJGetArray(mat, a));
  PetscFunctionReturn(PETSC_SUCCESS);
}
static PetscErrorCode SpikeRestoreCSR(Mat mat, PetscInt *n, const PetscInt **ia, const PetscInt **ja, PetscScalar **a)
{
  PetscBool done;
  PetscFunctionBegin;
  PetscCall(MatSeqAIJRestoreArray(mat, a));
  PetscCall(MatRestoreRowIJ(mat, 0, PETSC_FALSE, PETSC_FALSE, n, ia, ja, &done));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PETSC_EXTERN PetscErrorCode MatGetOrdering_WBM(Mat mat, MatOrderingType type, IS *row, IS *col) /* src/petsc_mat_wbm.c:13 */
{
  PetscInt        n;
  SpikeIdx        num = 0;
  const PetscInt *ia, *ja;
  PetscScalar    *a;
  PetscInt       *perm;
  PetscReal      *u, *v;
  PetscFunctionBegin;
  PetscCall(SpikeGetCSR(mat, &n, &ia, &ja, &a));
  PetscCall(PetscMalloc3(n, &perm, n, &u, n, &v));
  /* the CSR arrays go where MC64 expects CSC, as in the reference (:52): MC64 works on the transpose */
  PetscCheck(spike_mc64_job5X((SpikeIdx)n, (const SpikeIdx *)ia, (const SpikeIdx *)ja, a, (SpikeIdx *)perm, u, v, &num) == 0, PETSC_COMM_SELF, PETSC_ERR_LIB, "spike_mc64_job5 failed");
  PetscCall(ISCreateStride(PETSC_COMM_SELF, n, 0, 1, row));                     /* :57 */
  PetscCall(ISCreateGeneral(PETSC_COMM_SELF, n, perm, PETSC_COPY_VALUES, col)); /* :58 */
  PetscCall(PetscFree3(perm, u, v));                                            /* scalings dropped as :56,59 */
  PetscCall(SpikeRestoreCSR(mat, &n, &ia, &ja, &a));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PETSC_EXTERN PetscErrorCode MatGetOrdering_Fiedler(Mat mat, MatOrderingType type, IS *row, IS *col) /* src/petsc_mat_fiedler.c:11 */
{
  PetscInt        n;
  const PetscInt *ia, *ja;
  PetscScalar    *a;
  PetscInt       *order;
  PetscFunctionBegin;
  PetscCall(SpikeGetCSR(mat, &n, &ia, &ja, &a));
  PetscCall(PetscMalloc1(n, &order));
  PetscCheck(spike_fiedler_orderX((SpikeIdx)n, (const SpikeIdx *)ia, (const SpikeIdx *)ja, a, (SpikeIdx *)order) == 0, PETSC_COMM_SELF, PETSC_ERR_LIB, "spike_fiedler_order failed");
  PetscCall(ISCreateGeneral(PETSC_COMM_SELF, n, order, PETSC_OWN_POINTER, row)); /* :54 */
  PetscCall(PetscObjectReference((PetscObject)*row));                            /* the same IS for rows and columns, :55-56 */
  *col = *row;
  PetscCall(SpikeRestoreCSR(mat, &n, &ia, &ja, &a));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PETSC_EXTERN PetscErrorCode MatGetOrdering_AWBM(Mat mat, MatOrderingType type, IS *row, IS *col) /* src/petsc_mat_awbm.c:42 */
{
  PetscInt        n;
  const PetscInt *ia, *ja;
  PetscScalar    *a;
  PetscInt       *perm;
  PetscFunctionBegin;
  PetscCall(SpikeGetCSR(mat, &n, &ia, &ja, &a));
  PetscCall(PetscMalloc1(n, &perm));
  PetscCheck(spike_awbmX((SpikeIdx)n, (const SpikeIdx *)ia, (const SpikeIdx *)ja, a, (SpikeIdx *)perm) == 0, PETSC_COMM_SELF, PETSC_ERR_LIB, "spike_awbm failed");
  PetscCall(ISCreateStride(PETSC_COMM_SELF, n, 0, 1, row));
  PetscCall(ISCreateGeneral(PETSC_COMM_SELF, n, perm, PETSC_OWN_POINTER, col)); /* src/petsc_mat_awbm.c:202 */
  PetscCall(SpikeRestoreCSR(mat, &n, &ia, &ja, &a));
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* ---- KSPREORDER ----------------------------------------------------------------------------------------------------------- */
static PetscErrorCode KSPSetUp_Reorder(KSP ksp) /* :11-28 */
{
  KSP_Reorder *r = (KSP_Reorder *)ksp->data;
  Mat          A, M, PA, PM;
  PetscFunctionBegin;
  PetscCall(KSPGetOperators(ksp, &A, &M));
  PetscCall(ISDestroy(&r->rorder)); /* a second KSPSetUp must not leak the first ordering (the reference does) */
  PetscCall(ISDestroy(&r->corder));
  PetscCall(MatGetOrdering(M, r->ordertype, &r->rorder, &r->corder)); /* :19 */
  PetscCall(MatPermute(M, r->rorder, r->corder, &PM));                /* :20 */
  if (A != M) PetscCall(MatPermute(A, r->rorder, r->corder, &PA));    /* :21 */
  else PA = PM;
  PetscCall(KSPSetOperators(r->ksp, PA, PM)); /* :23 */
  PetscCall(KSPSetUp(r->ksp));                /* :24 -> PCSetUp_Banded -> spike_setup_csr_dist */
  PetscCall(MatDestroy(&PM));
  if (A != M) PetscCall(MatDestroy(&PA));
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode KSPSolve_Reorder(KSP ksp) /* :113-128, the live branch */
{
  KSP_Reorder *r = (KSP_Reorder *)ksp->data;
  Vec          x = ksp->vec_sol, b = ksp->vec_rhs;
  PetscBool    diagonalscale;
  PetscFunctionBegin;
  PetscCall(PCGetDiagonalScale(ksp->pc, &diagonalscale));
  PetscCheck(!diagonalscale, PetscObjectComm((PetscObject)ksp), PETSC_ERR_SUP, "Krylov method %s does not support diagonal scaling", ((PetscObject)ksp)->type_name); /* :120-121 */
  PetscCall(VecPermute(x, r->corder, PETSC_FALSE)); /* :122 in place on the caller's vectors */
  PetscCall(VecPermute(b, r->rorder, PETSC_FALSE)); /* :123 */
  PetscCall(KSPSolve(r->ksp, b, x));                /* :124 */
  PetscCall(KSPGetConvergedReason(r->ksp, &ksp->reason));
  PetscCall(KSPGetIterationNumber(r->ksp, &ksp->its));
  PetscCall(VecPermute(x, r->corder, PETSC_TRUE)); /* :126 */
  PetscCall(VecPermute(b, r->rorder, PETSC_TRUE)); /* :127 the right-hand side is restored */
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode KSPSetFromOptions_Reorder(KSP ksp, PetscOptionItems *PetscOptionsObject) /* :134-151 */
{
  KSP_Reorder      *r = (KSP_Reorder *)ksp->data;
  PetscFunctionList ordlist;
  char              tname[256];
  PetscBool         flg;
  PetscFunctionBegin;
  PetscOptionsHeadBegin(PetscOptionsObject, "KSP Reorder Options");
  PetscCall(PetscStrncpy(r->ordertype, MATORDERINGNATURAL, sizeof r->ordertype)); /* reset on every call, :144 */
  PetscCall(MatGetOrderingList(&ordlist));
  PetscCall(PetscOptionsFList("-mat_ordering_type", "Reordering for matrix", "main", ordlist, r->ordertype, tname, sizeof tname, &flg)); /* :146 */
  if (flg) PetscCall(PetscStrncpy(r->ordertype, tname, sizeof r->ordertype));
  PetscOptionsHeadEnd();
  PetscCall(KSPSetFromOptions(r->ksp)); /* :149 */
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode KSPView_Reorder(KSP ksp, PetscViewer viewer) /* :155-170 */
{
  KSP_Reorder *r = (KSP_Reorder *)ksp->data;
  PetscBool    isascii;
  PetscFunctionBegin;
  PetscCall(PetscObjectTypeCompare((PetscObject)viewer, PETSCVIEWERASCII, &isascii));
  if (isascii) {
    PetscCall(PetscViewerASCIIPrintf(viewer, "  reordering type = %s\n", r->ordertype));
    PetscCall(PetscViewerASCIIPushTab(viewer));
    PetscCall(KSPView(r->ksp, viewer));
    PetscCall(PetscViewerASCIIPopTab(viewer));
  }
  PetscFunctionReturn(PETSC_SUCCESS);
}

static PetscErrorCode KSPDestroy_Reorder(KSP ksp) /* :174-185 */
{
  KSP_Reorder *r = (KSP_Reorder *)ksp->data;
  PetscFunctionBegin;
  PetscCall(ISDestroy(&r->rorder));
  PetscCall(ISDestroy(&r->corder));
  PetscCall(KSPDestroy(&r->ksp));
  PetscCall(KSPDestroyDefault(ksp));
  PetscFunctionReturn(PETSC_SUCCESS);
}

PETSC_EXTERN PetscErrorCode KSPCreate_Reorder(KSP ksp) /* :197-223 */
{
  KSP_Reorder *r;
  const char  *prefix;
  PetscFunctionBegin;
  PetscCall(PetscNew(&r));
  ksp->data = (void *)r;
  PetscCall(KSPSetSupportedNorm(ksp, KSP_NORM_PRECONDITIONED, PC_LEFT, 2));   /* :207 */
  PetscCall(KSPSetSupportedNorm(ksp, KSP_NORM_UNPRECONDITIONED, PC_LEFT, 1)); /* :208 */
  ksp->ops->setup          = KSPSetUp_Reorder;
  ksp->ops->solve          = KSPSolve_Reorder;
  ksp->ops->destroy        = KSPDestroy_Reorder;
  ksp->ops->buildsolution  = KSPBuildSolutionDefault;
  ksp->ops->buildresidual  = KSPBuildResidualDefault;
  ksp->ops->view           = KSPView_Reorder;
  ksp->ops->setfromoptions = KSPSetFromOptions_Reorder;
  PetscCall(KSPCreate(PetscObjectComm((PetscObject)ksp), &r->ksp));
  PetscCall(PetscObjectGetOptionsPrefix((PetscObject)ksp, &prefix));
  PetscCall(KSPSetOptionsPrefix(r->ksp, prefix));
  PetscCall(KSPAppendOptionsPrefix(r->ksp, "reorder_")); /* :221 */
  PetscFunctionReturn(PETSC_SUCCESS);
}

/* = LoadModules of /root/reference/src/testbed2.c:61-73 */
PETSC_EXTERN PetscErrorCode PCCreate_Banded(PC);
PetscErrorCode SpikePetscRegisterAll(void)
{
  PetscFunctionBegin;
  PetscCall(MatOrderingRegister("wbm", MatGetOrdering_WBM));         /* :66 */
  PetscCall(MatOrderingRegister("awbm", MatGetOrdering_AWBM));       /* :67 */
  PetscCall(MatOrderingRegister("fiedler", MatGetOrdering_Fiedler)); /* :68 */
  PetscCall(PCRegister("banded", PCCreate_Banded));                  /* :70 */
  PetscCall(KSPRegister("reorder", KSPCreate_Reorder));              /* :71 */
  PetscFunctionReturn(PETSC_SUCCESS);
}
