import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "hostbox: host-only oracle comparison (MC64, AWBM, known answers) that must ALSO be "
                                       "in the GPU box's record: runs under -m 'not gpu' here and is added to -m gpu there")


@pytest.hookimpl(tryfirst=True)
def pytest_collection_modifyitems(config, items):
    """`-m gpu` is the driver's recorded run on the MI355X box.  The host-side oracle comparisons need no GPU, but a
    record that lacks them says nothing about a-8/a-9/f-2 parity, so under `-m gpu` they carry the gpu marker too."""
    if (config.getoption("markexpr") or "").strip() == "gpu":
        for it in items:
            if it.get_closest_marker("hostbox") is not None:
                it.add_marker(pytest.mark.gpu)


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
