import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def fpr():
    """The product: host mirror + libfpr_hip.so on cuda:0 (gpu tests only)."""
    import fpr_amd

    return fpr_amd.load()
