"""Shared pytest config: registers the `gpu` marker and puts the package directory on sys.path.

`-m "not gpu"` runs here on CPU (oracle vs golden vectors, host logic, C-ABI symbol checks,
gloo world_size-2 tests); `-m gpu` are the parity tests proper and run on a real MI355X.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "music-generation-emotion-adaptive_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture
def tune():
    """Set A/B switches of the native library for one test (mgea_tune_set; tools/README.md) and restore them afterwards."""
    from mgea import _lib
    saved = {}

    def set_(name, value):
        old = _lib.tune_set(name, value)
        saved.setdefault(name, old)
    yield set_
    for name, old in saved.items():
        _lib.tune_set(name, old)
