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


def golden_error(out, g) -> float:
    """Max abs error of `out` ([..., d], numpy or torch) against a fixture.  Full fixtures hold `out`; sliced ones
    (cases.slice_rows: large reference-geometry outputs) hold the first rows and every 16th row elementwise plus the L2 norm and
    the sum of EVERY row -- those two are scaled back to a per-element figure (|d norm| <= sqrt(d) max|err|, |d sum| <= d max|err|),
    so a row that is wrong anywhere still trips the same tolerance."""
    import cases
    a = out.detach().cpu().numpy() if hasattr(out, "detach") else np.asarray(out)
    if "out" in g.files:
        assert a.shape == g["out"].shape, (a.shape, g["out"].shape)
        return float(np.abs(a - g["out"]).max())
    a = a.reshape(-1, a.shape[-1]).astype(np.float32)
    d = a.shape[-1]
    assert a.shape[0] == g["norm"].shape[0], (a.shape, g["norm"].shape)
    s = cases.slice_rows(a)
    e = max(float(np.abs(s["head"] - g["head"]).max()), float(np.abs(s["strided"] - g["strided"]).max()))
    e = max(e, float(np.abs(s["norm"] - g["norm"]).max()) / np.sqrt(d), float(np.abs(s["rsum"] - g["rsum"]).max()) / d)
    return e


def golden_absmax(g) -> float:
    return float(np.abs(g["out"]).max()) if "out" in g.files else float(max(np.abs(g["head"]).max(), np.abs(g["strided"]).max()))


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """Make sure the C checker exists (gcc, seconds).  Building the checker is not using it."""
    so = os.path.join(ROOT, "oracle", "libvoxel_oracle.so")
    src = os.path.join(ROOT, "oracle", "voxel_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    yield


@pytest.fixture
def tune():
    """Explicit kernel-family choices for one test (include/lvq.h: lvq_tuning through _ffi.set_tuning): `tune(attn_nsplit=1)` changes
    fields of the record in force; the record implied by the environment is restored afterwards."""
    from lidar_vision_vqa_amd import _ffi

    def apply(**fields):
        cur = _ffi.get_tuning()
        cur.update(fields)
        _ffi._push(cur)

    yield apply
    _ffi.set_tuning()
