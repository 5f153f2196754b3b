"""API behaviour checklist: reference == oracle port == EXPECTED table, on CPU."""
import functools
import types

import numpy as np
import pytest

from semantics import EXPECTED, checklist


def namespace(O, backend):
    return types.SimpleNamespace(**{name: functools.partial(getattr(O, name), backend=backend)
                                    for name in ("PartitionedConvolve", "TimeDomainConvolve", "MonoConvolve", "NToMonoConvolve", "Convolver")})


def test_port_matches_expected(oracle):
    obs, (y, ya) = checklist(namespace(oracle, "port"))
    assert obs == EXPECTED
    assert np.allclose(ya, y + 1.0, atol=1e-6)              # accumulate=True adds to the existing out


def test_reference_matches_expected(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref (the compiled reference) is not present on this machine")
    obs, (y, ya) = checklist(namespace(oracle, "ref"))
    assert obs == EXPECTED
    assert np.allclose(ya, y + 1.0, atol=1e-6)
