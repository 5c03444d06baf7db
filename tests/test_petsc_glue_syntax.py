"""TYPO CHECK of the PETSc glue (examples/petsc/*.c): `gcc -fsyntax-only -Werror` against the declarations-only header
examples/petsc/syntax_check/petsc_decls_only.h, once with PETSc's default 32-bit PetscInt and once with
-DPETSC_USE_64BIT_INDICES.  PETSc is installed on neither box, so nothing is linked or run: this checks spelling, argument
counts and pointer types -- in particular that no PetscInt array reaches an entry point of the other index width (round 2's
glue cast `const PetscInt *` to `const int64_t *`, which corrupts the heap under a default PETSc build).  Not an oracle."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLUE = ["pcbanded_spike.c", "kspreorder_spike.c"]


@pytest.mark.parametrize("src", GLUE)
@pytest.mark.parametrize("idx", ["", "-DPETSC_USE_64BIT_INDICES"])
def test_glue_passes_the_compiler_front_end(src, idx):
    cmd = ["gcc", "-std=gnu11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter",
           "-I" + os.path.join(ROOT, "examples", "petsc", "syntax_check"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "petsc", src)]
    if idx:
        cmd.insert(1, idx)
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]


@pytest.mark.parametrize("src", GLUE)
def test_glue_never_casts_petscint_arrays_to_a_fixed_width(src):
    text = open(os.path.join(ROOT, "examples", "petsc", src)).read()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    assert not re.search(r"\(\s*(const\s+)?int64_t\s*\*\s*\)\s*&?\s*(ia|ja|perm|order|num)\b", code)
    assert "SpikeIdxMatchesPetscInt" in code          # the compile-time width check is in place


def test_width_mismatch_is_a_compile_error(tmp_path):
    """the guard works: forcing the 64-bit entry points onto a 32-bit PetscInt must not compile"""
    src = open(os.path.join(ROOT, "examples", "petsc", "kspreorder_spike.c")).read()
    bad = src.replace("#if defined(PETSC_USE_64BIT_INDICES)", "#if 1", 1)
    p = tmp_path / "bad.c"
    p.write_text(bad)
    out = subprocess.run(["gcc", "-std=gnu11", "-fsyntax-only", "-I" + os.path.join(ROOT, "examples", "petsc", "syntax_check"),
                          "-I" + os.path.join(ROOT, "include"), str(p)], capture_output=True, text=True)
    assert out.returncode != 0 and "SpikeIdxMatchesPetscInt" in out.stderr
