"""Import shim: the package directory is ``spike-petsc_amd/`` (a hyphen cannot be imported),
so ``import spike_petsc_amd`` resolves to this stub, which runs that directory's __init__."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "spike-petsc_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
