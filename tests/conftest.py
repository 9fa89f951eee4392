import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name: str):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """Make sure the C checker exists (gcc, seconds).  Building the checker is not using it."""
    so = os.path.join(ROOT, "oracle", "libvoxel_oracle.so")
    src = os.path.join(ROOT, "oracle", "voxel_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    yield
