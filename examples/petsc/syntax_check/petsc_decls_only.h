/*
 * petsc_decls_only.h -- DECLARATIONS ONLY, for `gcc -fsyntax-only` of the two glue files next to this directory (tests/test_petsc_glue_syntax.py).
 *
 * PETSc is installed on neither box, so the glue files cannot be compiled against it here.  This header declares just
 * the PETSc names those two files use (types, the PetscCall family as macros, prototypes as of PETSc 3.19-3.21), so that
 * the compiler can at least check spelling, argument counts and -- the point of round 3 -- that no PetscInt array is
 * handed to an entry point of the wrong index width (build with and without -DPETSC_USE_64BIT_INDICES).
 * It is a TYPO CHECK: nothing is linked or run with it, it is not an oracle, and it is not a stand-in for PETSc.
 */
#ifndef PETSC_DECLS_ONLY_H
#define PETSC_DECLS_ONLY_H
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#if defined(PETSC_USE_64BIT_INDICES)
typedef int64_t PetscInt;
#define PetscInt_FMT "lld"
#else
typedef int PetscInt;
#define PetscInt_FMT "d"
#endif
typedef double PetscReal;
typedef double PetscScalar;
typedef int PetscMPIInt;
typedef int PetscErrorCode;
typedef enum { PETSC_FALSE, PETSC_TRUE } PetscBool;
typedef int MPI_Comm;
typedef int MPI_Datatype;
#define MPI_BYTE 1
#define MPI_DOUBLE 2
#define PETSC_SUCCESS 0
#define PETSC_COMM_SELF 1
#define PETSC_COMM_WORLD 2
#define PETSC_ERR_SUP 56
#define PETSC_ERR_LIB 76
#define PETSC_ERR_PLIB 77
#define PETSC_ERR_ARG_OUTOFRANGE 63
#define PETSC_EXTERN extern
#define PetscAbsInt(a) (((a) < 0) ? -(a) : (a))

typedef struct _p_PetscObject { const char *type_name; } *PetscObject;
typedef struct _p_Mat *Mat;
typedef struct _p_Vec *Vec;
typedef struct _p_IS *IS;
typedef struct _p_PetscViewer *PetscViewer;
typedef struct _p_PetscOptionItems PetscOptionItems;
typedef struct _n_PetscFunctionList *PetscFunctionList;
typedef const char *MatOrderingType;
typedef const char *MatType;
typedef enum { KSP_CONVERGED_ITERATING = 0 } KSPConvergedReason;
typedef enum { KSP_NORM_PRECONDITIONED = 1, KSP_NORM_UNPRECONDITIONED = 2 } KSPNormType;
typedef enum { PC_LEFT = 0 } PCSide;
typedef enum { DIFFERENT_NONZERO_PATTERN, SUBSET_NONZERO_PATTERN, SAME_NONZERO_PATTERN } MatStructure;
typedef enum { MAT_INITIAL_MATRIX, MAT_REUSE_MATRIX } MatReuse;
typedef enum { MAT_FINAL_ASSEMBLY = 0 } MatAssemblyType;
typedef enum { NOT_SET_VALUES, INSERT_VALUES, ADD_VALUES } InsertMode;
typedef enum { PETSC_COPY_VALUES, PETSC_OWN_POINTER, PETSC_USE_POINTER } PetscCopyMode;
#define MATMPIAIJ "mpiaij"
#define MATAIJ "aij"
#define MATORDERINGNATURAL "natural"
#define PETSCVIEWERASCII "ascii"

typedef struct _p_PC *PC;
typedef struct _p_KSP *KSP;
struct _PCOps {
  PetscErrorCode (*setup)(PC);
  PetscErrorCode (*apply)(PC, Vec, Vec);
  PetscErrorCode (*applyrichardson)(PC, Vec, Vec, Vec, PetscReal, PetscReal, PetscReal, PetscInt, PetscBool, PetscInt *, int *);
  PetscErrorCode (*applytranspose)(PC, Vec, Vec);
  PetscErrorCode (*setfromoptions)(PC, PetscOptionItems *);
  PetscErrorCode (*reset)(PC);
  PetscErrorCode (*destroy)(PC);
  PetscErrorCode (*view)(PC, PetscViewer);
  PetscErrorCode (*applysymmetricleft)(PC, Vec, Vec);
  PetscErrorCode (*applysymmetricright)(PC, Vec, Vec);
};
struct _p_PC { struct _PCOps *ops; void *data; Mat mat, pmat; PetscInt setupcalled; MatStructure flag; };
struct _KSPOps {
  PetscErrorCode (*setup)(KSP);
  PetscErrorCode (*solve)(KSP);
  PetscErrorCode (*destroy)(KSP);
  PetscErrorCode (*buildsolution)(KSP, Vec, Vec *);
  PetscErrorCode (*buildresidual)(KSP, Vec, Vec, Vec *);
  PetscErrorCode (*view)(KSP, PetscViewer);
  PetscErrorCode (*setfromoptions)(KSP, PetscOptionItems *);
};
struct _p_KSP { struct _KSPOps *ops; void *data; Vec vec_sol, vec_rhs; PC pc; KSPConvergedReason reason; PetscInt its; };

#define PetscFunctionBegin do { } while (0)
#define PetscFunctionReturn(x) return (x)
#define PetscCall(...) do { PetscErrorCode ierr_ = (__VA_ARGS__); if (ierr_) return ierr_; } while (0)
#define PetscCallMPI(...) do { int ierr_ = (__VA_ARGS__); if (ierr_) return 98; } while (0)
PetscErrorCode PetscErrorDecl(MPI_Comm, PetscErrorCode, const char *, ...);
#define SETERRQ(comm, code, ...) return PetscErrorDecl(comm, code, __VA_ARGS__)
#define PetscCheck(cond, comm, code, ...) do { if (!(cond)) return PetscErrorDecl(comm, code, __VA_ARGS__); } while (0)
#define PetscOptionsHeadBegin(obj, title) do { (void)(obj); } while (0)
#define PetscOptionsHeadEnd() do { } while (0)
PetscErrorCode PetscMallocDecl(size_t, void *);
#define PetscNew(p) PetscMallocDecl(sizeof(**(p)), (void *)(p))
#define PetscMalloc1(n, p) PetscMallocDecl((size_t)(n) * sizeof(**(p)), (void *)(p))
#define PetscMalloc2(n1, p1, n2, p2) (PetscMalloc1(n1, p1) || PetscMalloc1(n2, p2))
#define PetscMalloc3(n1, p1, n2, p2, n3, p3) (PetscMalloc1(n1, p1) || PetscMalloc1(n2, p2) || PetscMalloc1(n3, p3))
#define PetscCalloc2(n1, p1, n2, p2) PetscMalloc2(n1, p1, n2, p2)
#define PetscCalloc3(n1, p1, n2, p2, n3, p3) PetscMalloc3(n1, p1, n2, p2, n3, p3)
PetscErrorCode PetscFreeDecl(void *);
#define PetscFree(p) PetscFreeDecl((void *)(p))
#define PetscFree2(a, b) (PetscFree(a) || PetscFree(b))
#define PetscFree3(a, b, c) (PetscFree(a) || PetscFree(b) || PetscFree(c))

int MPI_Comm_size(MPI_Comm, int *);
int MPI_Comm_rank(MPI_Comm, int *);
int MPI_Bcast(void *, int, MPI_Datatype, int, MPI_Comm);
int MPI_Allgather(const void *, int, MPI_Datatype, void *, int, MPI_Datatype, MPI_Comm);

MPI_Comm PetscObjectComm(PetscObject);
PetscErrorCode PetscObjectComposeFunction_Decl(PetscObject, const char *, void (*)(void));
#define PetscObjectComposeFunction(obj, name, fn) PetscObjectComposeFunction_Decl(obj, name, (void (*)(void))(fn))
PetscErrorCode PetscObjectTypeCompare(PetscObject, const char *, PetscBool *);
PetscErrorCode PetscObjectReference(PetscObject);
PetscErrorCode PetscObjectGetOptionsPrefix(PetscObject, const char **);
PetscErrorCode PetscInfo(void *, const char *, ...);
PetscErrorCode PetscSNPrintf(char *, size_t, const char *, ...);
PetscErrorCode PetscStrncpy(char *, const char *, size_t);
PetscErrorCode PetscOptionsInt(const char *, const char *, const char *, PetscInt, PetscInt *, PetscBool *);
PetscErrorCode PetscOptionsReal(const char *, const char *, const char *, PetscReal, PetscReal *, PetscBool *);
PetscErrorCode PetscOptionsBool(const char *, const char *, const char *, PetscBool, PetscBool *, PetscBool *);
PetscErrorCode PetscOptionsString(const char *, const char *, const char *, const char *, char *, size_t, PetscBool *);
PetscErrorCode PetscOptionsFList(const char *, const char *, const char *, PetscFunctionList, const char *, char *, size_t, PetscBool *);
PetscErrorCode PetscViewerASCIIPrintf(PetscViewer, const char *, ...);
PetscErrorCode PetscViewerASCIIPushTab(PetscViewer);
PetscErrorCode PetscViewerASCIIPopTab(PetscViewer);

PetscErrorCode MatGetSize(Mat, PetscInt *, PetscInt *);
PetscErrorCode MatGetOwnershipRange(Mat, PetscInt *, PetscInt *);
PetscErrorCode MatMPIAIJGetLocalMat(Mat, MatReuse, Mat *);
PetscErrorCode MatGetRowIJ(Mat, PetscInt, PetscBool, PetscBool, PetscInt *, const PetscInt *[], const PetscInt *[], PetscBool *);
PetscErrorCode MatRestoreRowIJ(Mat, PetscInt, PetscBool, PetscBool, PetscInt *, const PetscInt *[], const PetscInt *[], PetscBool *);
PetscErrorCode MatSeqAIJGetArray(Mat, PetscScalar *[]);
PetscErrorCode MatSeqAIJRestoreArray(Mat, PetscScalar *[]);
PetscErrorCode MatGetRow(Mat, PetscInt, PetscInt *, const PetscInt *[], const PetscScalar *[]);
PetscErrorCode MatRestoreRow(Mat, PetscInt, PetscInt *, const PetscInt *[], const PetscScalar *[]);
PetscErrorCode MatCreate(MPI_Comm, Mat *);
PetscErrorCode MatSetSizes(Mat, PetscInt, PetscInt, PetscInt, PetscInt);
PetscErrorCode MatSetType(Mat, MatType);
PetscErrorCode MatXAIJSetPreallocation(Mat, PetscInt, const PetscInt[], const PetscInt[], const PetscInt[], const PetscInt[]);
PetscErrorCode MatSetValues(Mat, PetscInt, const PetscInt[], PetscInt, const PetscInt[], const PetscScalar[], InsertMode);
PetscErrorCode MatAssemblyBegin(Mat, MatAssemblyType);
PetscErrorCode MatAssemblyEnd(Mat, MatAssemblyType);
PetscErrorCode MatDestroy(Mat *);
PetscErrorCode MatGetOrdering(Mat, MatOrderingType, IS *, IS *);
PetscErrorCode MatGetOrderingList(PetscFunctionList *);
PetscErrorCode MatOrderingRegister(const char[], PetscErrorCode (*)(Mat, MatOrderingType, IS *, IS *));
PetscErrorCode MatPermute(Mat, IS, IS, Mat *);
PetscErrorCode ISCreateStride(MPI_Comm, PetscInt, PetscInt, PetscInt, IS *);
PetscErrorCode ISCreateGeneral(MPI_Comm, PetscInt, const PetscInt[], PetscCopyMode, IS *);
PetscErrorCode ISDestroy(IS *);
PetscErrorCode VecPermute(Vec, IS, PetscBool);
PetscErrorCode VecHIPGetArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecHIPRestoreArrayRead(Vec, const PetscScalar **);
PetscErrorCode VecHIPGetArrayWrite(Vec, PetscScalar **);
PetscErrorCode VecHIPRestoreArrayWrite(Vec, PetscScalar **);
PetscErrorCode PCRegister(const char[], PetscErrorCode (*)(PC));
PetscErrorCode PCGetDiagonalScale(PC, PetscBool *);
PetscErrorCode KSPRegister(const char[], PetscErrorCode (*)(KSP));
PetscErrorCode KSPCreate(MPI_Comm, KSP *);
PetscErrorCode KSPDestroy(KSP *);
PetscErrorCode KSPDestroyDefault(KSP);
PetscErrorCode KSPGetOperators(KSP, Mat *, Mat *);
PetscErrorCode KSPSetOperators(KSP, Mat, Mat);
PetscErrorCode KSPSetUp(KSP);
PetscErrorCode KSPSolve(KSP, Vec, Vec);
PetscErrorCode KSPView(KSP, PetscViewer);
PetscErrorCode KSPSetFromOptions(KSP);
PetscErrorCode KSPGetConvergedReason(KSP, KSPConvergedReason *);
PetscErrorCode KSPGetIterationNumber(KSP, PetscInt *);
PetscErrorCode KSPSetSupportedNorm(KSP, KSPNormType, PCSide, PetscInt);
PetscErrorCode KSPSetOptionsPrefix(KSP, const char[]);
PetscErrorCode KSPAppendOptionsPrefix(KSP, const char[]);
PetscErrorCode KSPBuildSolutionDefault(KSP, Vec, Vec *);
PetscErrorCode KSPBuildResidualDefault(KSP, Vec, Vec, Vec *);
#endif
