/*
 * testbed2.c -- the reference's production driver, /root/reference/src/testbed2.c:77-142, rebuilt on the host mirror
 * (include/spike_petsc_host.h): same call order, same options, same printed check.
 *
 *   ./testbed2 -mat A.mtx -ksp_type reorder -mat_ordering_type wbm -mat_wbm_rows 1 \
 *              -reorder_ksp_type reorder -reorder_mat_ordering_type fiedler \
 *              -reorder_reorder_ksp_type gmres -reorder_reorder_ksp_rtol 1e-5 -reorder_reorder_ksp_max_it 500 \
 *              -reorder_reorder_pc_type banded
 *
 * Options it understands itself (testbed2.c:37-47): -mat <file> (.mtx = MatrixMarket, otherwise PETSc binary),
 * -random_exact_sol.  Everything else goes to the options database and is consumed by KSPSetFromOptions.
 */
#include <spike_petsc_host.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHKERRQ(e) do { PetscErrorCode ierr_ = (e); if (ierr_) { fprintf(stderr, "error %d: %s (%s:%d)\n", ierr_, SpikeHostLastError(), __FILE__, __LINE__); return ierr_; } } while (0)

int main(int argc, char **args)
{
    const char *matFilename = NULL;
    int randomSol = 0;
    Mat A, B;
    Vec x, b, u;
    KSP ksp;
    PetscReal error;
    PetscInt n, its;
    KSPConvergedReason reason;

    for (int i = 1; i < argc; ++i) { /* ProcessOptions, testbed2.c:20-50 */
        if (args[i][0] != '-') continue;
        const char *val = (i + 1 < argc && args[i + 1][0] != '-') ? args[i + 1] : "1";
        if (!strcmp(args[i], "-mat")) matFilename = val;
        else if (!strcmp(args[i], "-random_exact_sol")) randomSol = 1;
        else CHKERRQ(PetscOptionsSetValue(args[i], val));
    }
    if (!matFilename) { fprintf(stderr, "Must provide an input matrix using -mat <file>\n"); return PETSC_ERR_ARG_WRONG; } /* :38 */
    CHKERRQ(SpikePetscRegisterAll()); /* LoadModules, :61-73 */
    /* Load matrix, :93-96 */
    const size_t l = strlen(matFilename);
    if (l > 4 && !strcmp(matFilename + l - 4, ".mtx")) CHKERRQ(MatLoadMatrixMarket(matFilename, &A));
    else CHKERRQ(MatLoad(matFilename, &A));
    B = A; /* :105-108: the preconditioner is built from A itself */
    /* Create problem, :110-122 */
    CHKERRQ(MatGetSize(A, &n, NULL));
    CHKERRQ(VecCreateSeq(n, &u)); CHKERRQ(VecCreateSeq(n, &b)); CHKERRQ(VecCreateSeq(n, &x));
    if (randomSol) {
        PetscScalar *ua;
        unsigned long long s = 88172645463325252ULL;
        CHKERRQ(VecGetArray(u, &ua));
        for (PetscInt i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; ua[i] = (double)(s >> 11) / 9007199254740992.0; }
    } else CHKERRQ(VecSet(u, 1.0));
    CHKERRQ(MatMult(A, u, b));
    /* Create linear solver, :125-128 */
    CHKERRQ(KSPCreate(&ksp));
    CHKERRQ(KSPSetOperators(ksp, A, B));
    CHKERRQ(KSPSetFromOptions(ksp));
    CHKERRQ(KSPSolve(ksp, b, x));
    CHKERRQ(KSPGetIterationNumber(ksp, &its));
    CHKERRQ(KSPGetConvergedReason(ksp, &reason));
    CHKERRQ(KSPView(ksp, stdout));
    /* Check the error, :130-132 */
    CHKERRQ(VecAXPY(x, -1.0, u));
    CHKERRQ(VecNorm2(x, &error));
    printf("Iterations: %lld reason: %d\n", (long long)its, (int)reason);
    printf("Error in solution: %g\n", error);
    /* Cleanup, :134-140 */
    CHKERRQ(KSPDestroy(&ksp));
    CHKERRQ(VecDestroy(&u)); CHKERRQ(VecDestroy(&x)); CHKERRQ(VecDestroy(&b));
    CHKERRQ(MatDestroy(&A));
    return 0;
}
