import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session", autouse=True)
def _built_artifacts():
    """Built files travel with the snapshot; if they do not (fresh checkout), build them once.
    hipcc cross-compiles without a GPU, so this is safe on the CPU box as well."""
    import broadphase_rs_amd as bp
    lib = bp.library_path()
    exe = os.path.join(os.path.dirname(lib), "test_facade.bin")
    if not (os.path.exists(lib) and os.path.exists(exe)):
        import __graft_entry__
        __graft_entry__.build()
    yield
