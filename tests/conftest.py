import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    """max-norm relative error |a-b|_inf / max(|b|_inf, tiny)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def l2_rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))


L2_TOL = {"fp32": 1e-5, "bf16": 1e-2}


def both_errs(name, a, b):
    """{name: max-norm relative error, name + '/l2': L2-relative error}: the max-norm figure bounds the largest entries,
    the L2 one sees the many small-magnitude entries a max-norm bound lets through."""
    return {name: rel_err(a, b), name + "/l2": l2_rel_err(a, b)}


def failing(errs, tol, l2_tol):
    """entries of a both_errs-style dict above their bound (keys ending in '/l2' against l2_tol, the rest against tol)."""
    return {k: v for k, v in errs.items() if not v < (l2_tol if k.endswith("/l2") else tol)}
