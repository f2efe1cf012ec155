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


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests/` on a host without a GPU skips the gpu-marked tests instead of failing them.  On the GPU box
    (-m gpu) nothing is skipped: a missing device or a missing libtce_hip.so fails loudly in the fixtures."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        have = False
    if have or "gpu" in (config.getoption("-m") or "").replace("not gpu", ""):
        return
    skip = pytest.mark.skip(reason="no GPU on this host (run with -m gpu on an MI355X)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


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


def w4a16_report(got_f16: np.ndarray, ref_f32: np.ndarray, rel: float = 1e-3) -> dict:
    """What the floor in w4a16_close hides.  For every element: err = |gpu - ref|.
      frac_over_plain : share of elements with err > rel * |ref|  (they pass, if at all, only through the rms/64 floor)
      frac_fail       : share with err > rel * max(|ref|, rms/64)  (these fail w4a16_close)
      frac_small      : share of elements with |ref| < rms/64       (where the floor is the active bound at all)
      worst_plain     : max err / (rel * |ref|) over elements with |ref| >= rms/64 (the floor is not involved there)
      err_rms_over_rms: rms(err) / rms(ref)  -- the noise level of the kernel + the fp16 store, size-independent
    A result rounded to binary16 carries up to 2^-12 relative error by itself (half an ulp), so an element whose fp32 sum sits
    within a few 1e-4 relative of a rounding boundary can exceed 1e-3 * |ref| only when |ref| is small by cancellation."""
    got = got_f16.astype(np.float64)
    ref = ref_f32.astype(np.float64)
    rms = float(np.sqrt(np.mean(ref * ref)))
    eps = rms / 64.0
    err = np.abs(got - ref)
    big = np.abs(ref) >= eps
    over_plain = err > rel * np.abs(ref)
    return {
        "n": int(ref.size),
        "frac_over_plain": float(over_plain.mean()),
        "frac_fail": float((err > rel * np.maximum(np.abs(ref), eps)).mean()),
        "frac_small": float((~big).mean()),
        "worst_plain": float((err[big] / (rel * np.abs(ref[big]))).max()) if big.any() else 0.0,
        "err_rms_over_rms": float(np.sqrt(np.mean(err * err)) / rms) if rms > 0 else 0.0,
    }


def record_parity(name: str, rep: dict) -> None:
    """Appends one line to gpurun_out/parity_report.jsonl (merged back from the GPU box; summarised in DESIGN.md section 4)."""
    import json
    out = os.path.join(REPO, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report.jsonl"), "a") as f:
            f.write(json.dumps({"case": name, **rep}) + "\n")
    except OSError:
        pass


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
