import os
import sys

import pytest

try:
    # PyTorch-ROCm bundles its own libamdhip64; when torch shares a process with libhisstools_amd.so it has to be
    # loaded FIRST so both resolve to one HIP runtime.
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for everything except the HBM-resident tests
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle bindings (builds oracle/libhcv_oracle.so on first use)."""
    from oracle import oracle as O
    O.lib("port")
    return O


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")
    return np.load(path)


def rel_err(y, ref):
    import numpy as np
    ref = np.asarray(ref, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    peak = np.abs(ref).max()
    return float(np.abs(y - ref).max() / (peak if peak > 0 else 1.0))
