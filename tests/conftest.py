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


def fieldnorm_err(x, ref):
    """max|x - ref| / max|ref|  (field-normalised error)."""
    import numpy as np
    x = np.asarray(x, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    den = float(np.max(np.abs(ref)))
    if den == 0.0:
        return float(np.max(np.abs(x)))
    return float(np.max(np.abs(x - ref))) / den
