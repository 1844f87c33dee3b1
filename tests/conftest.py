import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bits_equal(a, b, what=""):
    """Bit-for-bit equality of float arrays (NaN == NaN when both are NaN)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    eq = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
    if not eq.all():
        bad = np.argwhere(~eq)
        raise AssertionError("%s: %d of %d values differ, first at %s: %r vs %r" % (
            what, (~eq).sum(), eq.size, bad[0], a[tuple(bad[0])], b[tuple(bad[0])]))


@pytest.fixture(scope="session")
def oracle_cls():
    from oracle import pyoracle
    if not os.path.exists(pyoracle.ORACLE_SO):
        pyoracle.build(ref=False)
    return pyoracle.Oracle
