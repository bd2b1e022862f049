import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def gpu_available():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def ref_available():
    import oracle

    return oracle.ref_available()


def sort_rows(*arrays):
    """Sort rows of the first array lexicographically and apply the same permutation to the rest."""
    a = arrays[0]
    idx = np.lexsort(a.T[::-1])
    return tuple(x[idx] for x in arrays)


def synthetic_frames(config, start, count, **kw):
    from pyslam_amd.synthetic import SyntheticRGBD

    s = SyntheticRGBD(config, **kw)
    return s, [s[i] for i in range(start, start + count)]
