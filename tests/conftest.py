import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Test-harness hook (the product library reads no environment): run the suite against another build of the C-ABI library,
    # e.g. the tuning build or a compile-time variant -- SRGPT_TEST_LIB=spatialrgpt_amd/libsrgpt_hip_tuning.so pytest -m gpu ...
    alt = os.environ.get("SRGPT_TEST_LIB")
    if alt:
        from spatialrgpt_amd import _lib
        _lib.LIB_PATH = os.path.abspath(alt)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
