import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Native libraries are built in-tree by __graft_entry__.build(); build them if this checkout has none yet."""
    from rfw_rs_amd import HIP_LIB, HOST_LIB
    from oracle.bindings import ORACLE_LIB
    if not (os.path.exists(HIP_LIB) and os.path.exists(HOST_LIB) and os.path.exists(ORACLE_LIB)):
        import __graft_entry__
        __graft_entry__.build()


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def rel_l2(a, b):
    import numpy as np
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))
