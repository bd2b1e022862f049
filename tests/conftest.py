import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a GPU: gpu-marked tests are skipped instead of failing in hv_create."""
    if gpu_available():
        return
    skip = pytest.mark.skip(reason="needs a gfx950 GPU (run on the GPU box with -m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


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
    return s, s.frames(start, count)


def canonical_mesh(verts, tris, cols):
    """Order-free form of a triangle mesh: vertices rounded to 1e-9 and sorted; triangles as tuples of the re-indexed
    vertices, rotated so that the smallest index comes first (orientation preserved), then sorted."""
    key = np.round(verts, 9)
    order = np.lexsort(key.T[::-1])
    rank = np.empty(len(order), np.int64)
    rank[order] = np.arange(len(order))
    t = rank[tris]
    if len(t):
        rot = np.argmin(t, axis=1)
        t = t[np.arange(len(t))[:, None], (rot[:, None] + np.arange(3)[None, :]) % 3]
        t = t[np.lexsort(t.T[::-1])]
    return verts[order], cols[order], t
