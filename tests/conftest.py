import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.lib()
    return O


def _ensure_built():
    """A fresh checkout has no .so files (they are git-ignored): compile them once (hipcc cross-compiles without a GPU).
    This builds the product; it never substitutes anything for it -- if the build fails the tests fail."""
    import spike_petsc_amd as S
    host = os.path.join(ROOT, "spike-petsc_amd", "libspike_petsc_host.so")
    if not (os.path.exists(S.LIB_PATH) and os.path.exists(host)):
        S.build()


@pytest.fixture(scope="session")
def spike():
    """The product library through its C-ABI.  Fails (does not skip) when it cannot be built or loaded."""
    _ensure_built()
    import spike_petsc_amd as S
    S.lib()
    return S
