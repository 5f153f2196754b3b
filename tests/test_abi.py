"""The C-ABI library loads on a machine without a GPU and exports exactly what include/hisstools_amd.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "hisstools_amd.h")).read()
    return sorted(set(re.findall(r"HCV_API[^;]*?\b(hcv_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from hisstools_library_amd import _lib
    lib = _lib.load()                       # raises ImportError if the .so was not built
    declared = header_symbols()
    assert len(declared) >= 55
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in hisstools_amd.h but not exported"
    # and the Python binding table covers the header one-to-one
    assert sorted(_lib.SIGNATURES) == declared


def test_version_and_device_query_do_not_need_a_gpu():
    from hisstools_library_amd import _lib
    lib = _lib.load()
    assert lib.hcv_version().startswith(b"hisstools_amd")
    assert lib.hcv_device_count() >= 0


def test_no_cpu_fallback():
    """Without a GPU every constructor must fail loudly instead of computing on the host."""
    import hisstools_library_amd as H
    if H.load().hcv_device_count() > 0:
        pytest.skip("a GPU is present")
    for make in (lambda: H.Convolver(2, 2, 0), lambda: H.MonoConvolve(16384, latency=0), lambda: H.PartitionedConvolve(4096, 8192, 0, 0),
                 lambda: H.TimeDomainConvolve(0, 128), lambda: H.NToMonoConvolve(2, 16384, 0)):
        with pytest.raises(RuntimeError, match="no HIP device"):
            make()
    with pytest.raises(RuntimeError):
        H.hisstools_rfft([0.0] * 64, 6)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "hisstools_library_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower() or f == "__init__.py" and False, f"{f} mentions the oracle"


def test_mono_ctor_errors_surface_without_gpu():
    # the size validation of MonoConvolve::setPartitions happens before any device work
    import hisstools_library_amd as H
    with pytest.raises(RuntimeError, match="invalid FFT size or order"):
        H.MonoConvolve(1000, zeroLatency=True, A=1024, B=256)
    with pytest.raises(RuntimeError, match="no valid FFT sizes given"):
        H.MonoConvolve(1000, zeroLatency=False, A=0)
