"""pytest configuration: the ``gpu`` marker and common paths.

``-m "not gpu"`` runs everywhere (oracle vs golden vectors, host logic, C-ABI symbol
check, CPU emulation of the kernels, gloo multi-process tests).  ``-m gpu`` needs a real
MI355X and calls the HIP path through the C-ABI library.
"""

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
