import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One dxtex context on cuda:0 for the whole session. Fails loudly (no skip, no fallback) if the HIP
    library or the GPU is missing: GPU tests must never pass on a silent CPU path."""
    import directxtex_amd as dx
    c = dx.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    if not o.have_ref():
        pytest.fail("oracle/_ref/libdxtex_ref.so is missing: run `make -C oracle ref` where /root/reference exists")
    return o
