/*
 * spike_petsc_host.h -- host-side mirror of the reference's PETSc plugin surface (libspike_petsc_host.so).
 *
 * PETSc is absent on both boxes, so the plugin objects of /root/reference/src are rebuilt here on a minimal object
 * model that keeps the reference's NAMES, argument meaning, option keys and error behaviour, so that callers (and the
 * parity tests) read like /root/reference/src/testbed2.c:
 *
 *   MatCreateSubMatrixBanded              src/matbanded.c:22-107      (decl. src/matbanded.h:5)
 *   PCCreate_Banded + ops                 src/matbanded.c:111-283     options -pc_banded_kmax / -pc_banded_frac (:156-157)
 *   PCBandedSetMaxHalfBandwidth / NormFraction  src/matbanded.c:215-343 (the reference's wrappers are no-ops by a
 *                                          typo in the composed-function name, :311,341; these ones work)
 *   KSPCreate_Reorder + ops               src/kspreorder.c:1-223      option -mat_ordering_type (:146), prefix reorder_ (:221)
 *   MatGetOrdering_WBM                    src/petsc_mat_wbm.c:13-61   (HSLmc64AD job 5 -> spike_mc64_job5)
 *   MatGetOrdering_AWBM                   src/petsc_mat_awbm.c:42-225 (approximate matching, 5 greedy phases -> spike_awbm)
 *   MatGetOrdering_Fiedler                src/petsc_mat_fiedler.c:11-58 (HSL_MC73 absent -> spike_fiedler_order, own spec)
 *   SpikePetscRegisterAll                 LoadModules, src/testbed2.c:61-73
 *
 * The inner PC of PCBANDED (prefix "banded_", matbanded.c:281) is the MI355X engine: PC type "spike"
 * (include/spike_mi355.h).  KSP type "gmres" runs entirely on the device through spike_gmres.
 * Error convention: every function returns PetscErrorCode (0 = success), as in the reference; the message of the last
 * error is available from SpikeHostLastError().
 */
#ifndef SPIKE_PETSC_HOST_H
#define SPIKE_PETSC_HOST_H
#include <stdint.h>
#include <stdio.h>
#include "spike_orderings.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef int PetscErrorCode;
typedef int64_t PetscInt;
typedef double PetscReal;
typedef double PetscScalar;
typedef int PetscBool;
#define PETSC_TRUE 1
#define PETSC_FALSE 0
/* PETSc's numeric error classes */
#define PETSC_ERR_MEM 55
#define PETSC_ERR_SUP 56
#define PETSC_ERR_ARG_SIZ 60
#define PETSC_ERR_ARG_WRONG 62
#define PETSC_ERR_ARG_OUTOFRANGE 63
#define PETSC_ERR_ARG_WRONGSTATE 73
#define PETSC_ERR_LIB 76
#define PETSC_ERR_ARG_UNKNOWN_TYPE 86

typedef struct _p_Mat *Mat;
typedef struct _p_Vec *Vec;
typedef struct _p_IS *IS;
typedef struct _p_PC *PC;
typedef struct _p_KSP *KSP;
typedef const char *MatOrderingType;
typedef const char *PCType;
typedef const char *KSPType;

#define MATORDERINGNATURAL "natural"
#define PCBANDED "banded"
#define PCSPIKE "spike"
#define PCNONE "none"
#define KSPGMRES "gmres"
#define KSPREORDER "reorder"

typedef enum { KSP_CONVERGED_ITERATING = 0, KSP_CONVERGED_RTOL = 2, KSP_DIVERGED_ITS = -3, KSP_DIVERGED_BREAKDOWN = -5 } KSPConvergedReason;

const char *SpikeHostLastError(void);
PetscErrorCode SpikePetscRegisterAll(void); /* testbed2.c:61-73 */

/* options database (PetscOptionsSetValue-like; names carry the leading '-') */
PetscErrorCode PetscOptionsSetValue(const char *name, const char *value);
PetscErrorCode PetscOptionsClearValue(const char *name);
PetscErrorCode PetscOptionsClear(void);

/* Mat (SeqAIJ: CSR, 0-based, copied on creation) */
PetscErrorCode MatCreateSeqAIJWithArrays(PetscInt n, const PetscInt *ia, const PetscInt *ja, const PetscScalar *a, Mat *A);
PetscErrorCode MatDestroy(Mat *A);
PetscErrorCode MatGetSize(Mat A, PetscInt *m, PetscInt *n);
PetscErrorCode MatSeqAIJGetCSR(Mat A, PetscInt *n, const PetscInt **ia, const PetscInt **ja, const PetscScalar **a);
PetscErrorCode MatMult(Mat A, Vec x, Vec y);
PetscErrorCode MatPermute(Mat A, IS rowp, IS colp, Mat *B); /* B[i][j] = A[rowp[i]][colp[j]] */
PetscErrorCode MatComputeBandwidth(Mat A, PetscReal fraction, PetscInt *bw);
PetscErrorCode MatCreateSubMatrixBanded(Mat A, PetscInt *kmax, PetscReal *frac, Mat *B);
/* file formats of the reference's drivers: PETSc binary AIJ (testbed2.c:93-96) and MatrixMarket (wbm.c:476-477,520-522) */
PetscErrorCode MatLoad(const char *petsc_binary_path, Mat *A);
PetscErrorCode MatViewBinary(Mat A, const char *path);
PetscErrorCode MatLoadMatrixMarket(const char *path, Mat *A);
PetscErrorCode MatViewMatrixMarket(Mat A, const char *path);

/* Vec (sequential, host array) */
PetscErrorCode VecCreateSeq(PetscInt n, Vec *v);
PetscErrorCode VecDestroy(Vec *v);
PetscErrorCode VecGetArray(Vec v, PetscScalar **a);
PetscErrorCode VecGetSize(Vec v, PetscInt *n);
PetscErrorCode VecSet(Vec v, PetscScalar s);
PetscErrorCode VecCopy(Vec x, Vec y);
PetscErrorCode VecAXPY(Vec y, PetscScalar alpha, Vec x);
PetscErrorCode VecNorm2(Vec v, PetscReal *nrm);
PetscErrorCode VecPermute(Vec v, IS is, PetscBool inv); /* inv=FALSE: v[i] <- v[is[i]] ; TRUE: the inverse */

/* IS */
PetscErrorCode ISCreateGeneral(PetscInt n, const PetscInt *idx, IS *is);
PetscErrorCode ISCreateStride(PetscInt n, PetscInt first, PetscInt step, IS *is);
PetscErrorCode ISDestroy(IS *is);
PetscErrorCode ISGetIndices(IS is, PetscInt *n, const PetscInt **idx);

/* orderings: PetscErrorCode f(Mat, MatOrderingType, IS *row, IS *col)   (testbed2.c:52-54) */
typedef PetscErrorCode (*MatOrderingFn)(Mat, MatOrderingType, IS *, IS *);
PetscErrorCode MatOrderingRegister(const char *name, MatOrderingFn fn);
PetscErrorCode MatGetOrdering(Mat A, MatOrderingType type, IS *row, IS *col);
PetscErrorCode MatGetOrdering_WBM(Mat A, MatOrderingType type, IS *row, IS *col);
PetscErrorCode MatGetOrdering_AWBM(Mat A, MatOrderingType type, IS *row, IS *col);
PetscErrorCode MatGetOrdering_Fiedler(Mat A, MatOrderingType type, IS *row, IS *col);
PetscErrorCode MatGetOrdering_Natural(Mat A, MatOrderingType type, IS *row, IS *col);
PetscErrorCode MatGetOrdering_RCM(Mat A, MatOrderingType type, IS *row, IS *col);

/* PC */
typedef PetscErrorCode (*PCCreateFn)(PC);
PetscErrorCode PCRegister(const char *name, PCCreateFn fn);
PetscErrorCode PCCreate(PC *pc);
PetscErrorCode PCSetType(PC pc, PCType type);
PetscErrorCode PCSetOptionsPrefix(PC pc, const char *prefix);
PetscErrorCode PCAppendOptionsPrefix(PC pc, const char *prefix);
PetscErrorCode PCSetOperators(PC pc, Mat A, Mat P);
PetscErrorCode PCSetFromOptions(PC pc);
PetscErrorCode PCSetUp(PC pc);
PetscErrorCode PCApply(PC pc, Vec x, Vec y);
PetscErrorCode PCReset(PC pc);
PetscErrorCode PCDestroy(PC *pc);
PetscErrorCode PCView(PC pc, FILE *viewer);
PetscErrorCode PCGetDiagonalScale(PC pc, PetscBool *flag);
PetscErrorCode PCCreate_Banded(PC pc);
PetscErrorCode PCCreate_Spike(PC pc);
PetscErrorCode PCCreate_None(PC pc);
PetscErrorCode PCBandedSetMaxHalfBandwidth(PC pc, PetscInt kmax);
PetscErrorCode PCBandedSetNormFraction(PC pc, PetscReal frac);
PetscErrorCode PCBandedGetInfo(PC pc, PetscInt *k, PetscReal *f, PetscInt *kmax, PetscReal *frac);
/* the engine handle behind a PC (banded -> its inner spike PC), or NULL */
PetscErrorCode PCGetSpikeHandle(PC pc, void **handle);

/* KSP */
typedef PetscErrorCode (*KSPCreateFn)(KSP);
PetscErrorCode KSPRegister(const char *name, KSPCreateFn fn);
PetscErrorCode KSPCreate(KSP *ksp);
PetscErrorCode KSPSetType(KSP ksp, KSPType type);
PetscErrorCode KSPSetOptionsPrefix(KSP ksp, const char *prefix);
PetscErrorCode KSPAppendOptionsPrefix(KSP ksp, const char *prefix);
PetscErrorCode KSPSetOperators(KSP ksp, Mat A, Mat M);
PetscErrorCode KSPGetPC(KSP ksp, PC *pc);
PetscErrorCode KSPSetTolerances(KSP ksp, PetscReal rtol, PetscInt maxits);
PetscErrorCode KSPSetFromOptions(KSP ksp);
PetscErrorCode KSPSetUp(KSP ksp);
PetscErrorCode KSPSolve(KSP ksp, Vec b, Vec x);
PetscErrorCode KSPGetConvergedReason(KSP ksp, KSPConvergedReason *reason);
PetscErrorCode KSPGetIterationNumber(KSP ksp, PetscInt *its);
PetscErrorCode KSPGetResidualNorm(KSP ksp, PetscReal *rnorm);
PetscErrorCode KSPView(KSP ksp, FILE *viewer);
PetscErrorCode KSPDestroy(KSP *ksp);
PetscErrorCode KSPCreate_Reorder(KSP ksp);
PetscErrorCode KSPCreate_GMRES(KSP ksp);
PetscErrorCode KSPReorderGetOrdering(KSP ksp, IS *row, IS *col); /* borrowed references */

/* the ordering kernels themselves (plain C, no PETSc types): include/spike_orderings.h */
PetscErrorCode MatGetOrdering_FiedlerHalves(Mat A, MatOrderingType type, IS *row, IS *col);
#ifdef __cplusplus
}
#endif
#endif
