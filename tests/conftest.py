import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A process that uses both PyTorch and libpslfe on the GPU (tests/test_gather_gpu.py, as bench.py) must load PyTorch - and
    with it PyTorch's copy of the HIP runtime - FIRST: loaded after libpslfe it finds no device."""
    m = session.config.getoption("-m") or ""
    if ("gpu" in m and "not gpu" not in m) or (os.path.exists("/dev/kfd") and "not gpu" not in m):   # a GPU box, however the tests were selected
        try:
            import torch
            torch.cuda.is_available()
        except Exception:
            pass


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def ctx():
    import psl_slam_amd as P
    return P.default_context()
