import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def make_data(seed, n, d, k, skew=False, scale_between=0.0):
    """Seeded synthetic data in the reference's usage distribution (uniform [0,1) rows,
    README.md:54-55, tests/pldatest.py:10-11), optionally with real speaker structure."""
    rng = np.random.default_rng(seed)
    if skew:
        y = np.concatenate([np.arange(k), rng.integers(0, k, n - k)]).astype(np.uint64)
    else:
        y = (np.arange(n) % k).astype(np.uint64)
    x = rng.random((n, d))
    if scale_between:
        x = x + scale_between * rng.standard_normal((k, d))[y.astype(np.int64)]
    return x, y


def score_tol(ref):
    """|delta| <= 1e-4 * max(|ref|, mean|ref|)  (north_star: 1e-4 relative; scores cross zero)."""
    ref = np.asarray(ref, np.float64)
    return 1e-4 * np.maximum(np.abs(ref), np.abs(ref).mean())


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.build()
    return binding


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = ["tiny_unequal", "pldatest_shape", "c1_readme"]


def load_golden(name):
    """Fixture + its inputs (stored, or regenerated from the stored seed exactly as
    tests/golden/make_golden.py did)."""
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    n, d = int(g["N"]), int(g["D"])
    if "X" not in g:
        rng = np.random.default_rng(int(g["seed"]))
        g["X"] = rng.random((n, d))
    return g
