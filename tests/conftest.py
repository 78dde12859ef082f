import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + '.npz'))
        return cache[name]
    return load


def rel_err(got, want):
    """max|got-want| / max|want| — the per-tensor relative error SURVEY §7 defines for linear outputs."""
    got, want = np.asarray(got), np.asarray(want)
    wide = np.complex128 if (np.iscomplexobj(got) or np.iscomplexobj(want)) else np.float64
    got, want = got.astype(wide), want.astype(wide)
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-300))
