import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)   # golden_inputs.py (recomputable inputs of the fixtures)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a ROCm GPU (run on the MI355X box with -m gpu)')


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'))
    return load


@pytest.fixture(scope='session')
def oracle():
    from oracle import lgcn_oracle
    lgcn_oracle.build()
    return lgcn_oracle


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('a test marked gpu ran without a GPU: the HIP path has no CPU fallback')
    return torch.device('cuda:0')


def bits(a):
    """fp32 array -> uint32 view for bit-exact comparison"""
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def normwise(a, ref):
    """max|a - ref| / max|ref| (the 1e-4 bar is normwise)"""
    ref = np.asarray(ref, dtype=np.float64)
    a = np.asarray(a, dtype=np.float64)
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(a), fin)
    assert np.array_equal(a[~fin], ref[~fin])
    den = np.abs(ref[fin]).max() if fin.any() else 1.0
    return float(np.abs(a[fin] - ref[fin]).max() / max(den, 1e-30)) if fin.any() else 0.0
