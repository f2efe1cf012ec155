import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden", "hotpath_golden.npz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The reference's own code (oracle/_ref/libtce_ref.so); skipped where that prebuilt file is absent."""
    from oracle import oracle as O
    if not O.have_ref():
        pytest.skip("oracle/_ref/libtce_ref.so not built (needs /root/reference at build time)")
    return O.Reference()


@pytest.fixture(scope="session")
def golden():
    return dict(np.load(GOLDEN))


def w4a16_close(got_f16: np.ndarray, ref_f32: np.ndarray, rel: float = 1e-3):
    """North-star tolerance for the W4A16 path: |gpu - ref| <= 1e-3 * max(|ref|, eps) per element, where
    eps = 2^-6 * rms(ref) guards outputs that are tiny only because large terms cancel (fp32 accumulation order and
    the final fp16 rounding, 2^-11 relative, are the two error sources -- SURVEY App. B).  Returns (ok, worst ratio)."""
    got = got_f16.astype(np.float64)
    ref = ref_f32.astype(np.float64)
    eps = float(np.sqrt(np.mean(ref * ref))) / 64.0
    tol = rel * np.maximum(np.abs(ref), eps)
    ratio = np.abs(got - ref) / tol
    return bool(np.all(ratio <= 1.0)), float(ratio.max())
