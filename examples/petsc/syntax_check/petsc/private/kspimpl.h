#include "../../petsc_decls_only.h" /* declarations only: see that file */
