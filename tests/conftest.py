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


def pytest_sessionstart(session):
    # HCV_NATIVE_BACKTRACE=1: should the process die of a signal, the faulting thread's native call stack goes to stderr in front of
    # faulthandler's Python stacks (a crash in a runtime thread shows no Python frame at all)
    if os.environ.get("HCV_NATIVE_BACKTRACE"):
        import faulthandler
        faulthandler.enable()
        import hisstools_library_amd as H
        H.load().hcv_debug_native_backtrace_on_crash()


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
