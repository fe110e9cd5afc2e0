import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# libtemx.so is a build artefact (git-ignored): build it in-tree when a fresh checkout runs the
# tests.  hipcc cross-compiles for gfx950 without a GPU.
_LIB = os.path.join(ROOT, "pytemdiags_amd", "libtemx.so")
if not os.path.exists(_LIB):
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "pytemdiags_amd", "csrc")], check=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def fieldnorm_err(x, ref):
    """max|x - ref| / max|ref|  (field-normalised error)."""
    import numpy as np
    x = np.asarray(x, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    den = float(np.max(np.abs(ref)))
    if den == 0.0:
        return float(np.max(np.abs(x)))
    return float(np.max(np.abs(x - ref))) / den
