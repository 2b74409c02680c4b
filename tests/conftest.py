import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """The product library; built on demand (hipcc cross-compiles without a GPU)."""
    from ngs_amd import build, ffi
    build.build(verbose=False)
    return ffi.load_library()


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle_py
    oracle_py.load()
    return oracle_py


@pytest.fixture(scope="session")
def gpu_lib(lib):
    if lib.ngsq_device_count() < 1:
        pytest.fail("a test marked gpu ran without a HIP device: the HIP path has no fallback")
    return lib
