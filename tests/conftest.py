import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def expected():
    """Outputs of the reference on the seeded cases (tests/make_golden.py)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "expected.npz")
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def rel_err(result, truth):
    """max_i ||result_i - truth_i||_2 / max_i ||truth_i||_2 over rows that are finite in the
    truth (metrics.py:53-56 norm, made relative)."""
    result = np.asarray(result, dtype=np.float64)
    truth = np.asarray(truth, dtype=np.float64)
    fin = np.isfinite(truth).all(axis=-1)
    if not fin.any():
        return 0.0
    scale = np.max(np.sqrt(np.sum(truth[fin] ** 2, axis=-1)))
    err = np.max(np.sqrt(np.sum((result[fin] - truth[fin]) ** 2, axis=-1)))
    return float(err / (scale if scale > 0 else 1.0))
