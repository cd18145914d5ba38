import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "real_limit: (test_gpu_sparse.py) keep the dense limit of the combination grids where it is")


def _ensure_built():
    """libscg.so and liboracle.so are build products; make them if the tree is fresh."""
    lib = os.path.join(ROOT, "screencounter_amd", "libscg.so")
    ora = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "screencounter_amd", "csrc")])
    if not os.path.exists(ora):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "liboracle.so")])


_ensure_built()


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


class _Both:
    """The package, plus the tests' R-level mirror (tests/rlevel.py: countSingleBarcodes, matrixOf* ...), under one name --
    the reference's test files call both layers the same way."""

    def __init__(self, package, rlevel):
        self._package, self._rlevel = package, rlevel

    def __getattr__(self, name):
        if hasattr(self._package, name):
            return getattr(self._package, name)
        return getattr(self._rlevel, name)


@pytest.fixture(scope="session")
def sc():
    import screencounter_amd
    from tests import rlevel
    screencounter_amd.load()
    return _Both(screencounter_amd, rlevel)


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
