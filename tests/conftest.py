import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_bin():
    """Path of the oracle CLI (CPU restatement of the reference; test infrastructure)."""
    path = os.path.join(ROOT, "oracle", "_build", "hal_oracle")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    assert os.path.exists(path)
    return path


@pytest.fixture(scope="session")
def hal():
    """The hal_amd package (ctypes binding of libhgx.so); builds the library if it is missing."""
    lib = os.path.join(ROOT, "hal_amd", "libhgx.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()
    import hal_amd
    return hal_amd
