import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def vmlib():
    """The product library; built in-tree if missing (hipcc cross-compiles on CPU)."""
    from videomorphing_amd import build, capi
    if not os.path.exists(capi.LIB_PATH):
        build.build()
    return capi.load()


@pytest.fixture(scope="session")
def gpu_ctx(vmlib):
    from videomorphing_amd import morph
    return morph.Context(0)
