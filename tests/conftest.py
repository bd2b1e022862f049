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


FOLD_TSDF_TOL = 5e-6  # fold form of the sweep vs the oracle's per-frame float chain (contract: 1e-4; measured <= 1e-6)


def sweep_is_bitwise():
    """HV_TSDF_SWEEP=2 selects the form of the multi-frame sweep that replays the reference's running mean frame by
    frame (tsdf bit-identical to the oracle); the production form (4: whole voxel columns, the default) folds a batch
    per voxel (tsdf within FOLD_TSDF_TOL)."""
    return os.environ.get("HV_TSDF_SWEEP", "4") == "2"


def assert_tsdf_parity(ta, tb, swept=True):
    """tsdf planes of the HIP volume vs the oracle: bitwise for the online path and the bitwise sweep forms, within
    FOLD_TSDF_TOL when the volume went through the fold form of the sweep."""
    if not swept or sweep_is_bitwise():
        np.testing.assert_array_equal(ta.view(np.uint32), tb.view(np.uint32))
        return 0.0
    worst = 0.0
    for lo in range(0, len(ta), 512):
        worst = max(worst, float(np.abs(ta[lo:lo + 512] - tb[lo:lo + 512]).max()))
    assert worst <= FOLD_TSDF_TOL, worst
    return worst


def assert_dumps_match(da, db, swept=True):
    """(keys, tsdf, weight, colour) of two volumes: keys, weights and colours identical, tsdf per assert_tsdf_parity."""
    np.testing.assert_array_equal(da[0], db[0])
    np.testing.assert_array_equal(da[2], db[2])
    np.testing.assert_array_equal(da[3], db[3])
    assert_tsdf_parity(da[1], db[1], swept)


@pytest.fixture(params=["fold", "bitwise"])
def sweep_form(request, monkeypatch):
    """Runs a test once per production form of the multi-frame sweep (the switch is read per call)."""
    monkeypatch.setenv("HV_TSDF_SWEEP", "4" if request.param == "fold" else "2")
    return request.param


def sort_rows(*arrays):
    """Sort rows of the first array lexicographically and apply the same permutation to the rest."""
    a = arrays[0]
    idx = np.lexsort(a.T[::-1])
    return tuple(x[idx] for x in arrays)


def synthetic_frames(config, start, count, **kw):
    from pyslam_amd.synthetic import SyntheticRGBD

    s = SyntheticRGBD(config, **kw)
    return s, s.frames(start, count)


def _lex_less(a, b):
    """Row-wise lexicographic a < b for two [n, k] arrays."""
    d = a != b
    idx = d.argmax(axis=1)
    rows = np.arange(len(a))
    return d.any(axis=1) & (a[rows, idx] < b[rows, idx])


def canonical_mesh(verts, tris, cols):
    """Order-free form of a triangle mesh: vertices (rounded to 1e-9) sorted, with their colours; triangles as rows of
    their three vertex POSITIONS [n, 9], each rotated to its lexicographically smallest rotation (orientation preserved)
    and the rows sorted.  Positions instead of vertex indices: marching cubes emits coincident vertices on different
    edges when a tsdf value is exactly 0 at a voxel corner, and ranks of coincident vertices are arbitrary."""
    key = np.round(verts, 9)
    order = np.lexsort(key.T[::-1])
    t = np.zeros((0, 9))
    if len(tris):
        p = key[tris]  # [n, 3, 3]
        rots = [np.concatenate([p[:, (r + k) % 3] for k in range(3)], axis=1) for r in range(3)]
        t = rots[0]
        for r in rots[1:]:
            less = _lex_less(r, t)
            t = np.where(less[:, None], r, t)
        t = t[np.lexsort(t.T[::-1])]
    return verts[order], cols[order], t
