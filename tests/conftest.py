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
    """CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def fa():
    """The product: ctypes-bound HIP library.  Must already be built (or buildable) -- no fallback."""
    import flashattention_kernel_project_amd as fa_
    if not os.path.exists(fa_.capi.LIB_PATH):
        fa_.build()
    fa_.lib()
    return fa_


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
