"""The reference's own known-answer tests for the semantic voxels (cpp/test_volumetric_voxel_semantic.py),
replayed through any grid factory ``make(kind, voxel_size)`` whose grids offer integrate(points, colors,
class_ids, instance_ids, depths) and get_voxels(min_count, min_confidence) -> (pts, cols, cls, obj, conf).
The reference runs them on the direct voxel grids; the payloads (and so the answers) are the same on the
block grids used here."""
import numpy as np
import pytest

VOTE, PROB = 0, 1


def _z(n):
    return np.zeros((n, 3), np.float64), np.zeros((n, 3), np.uint8)


def run_reference_kats(make):
    # :40-57 majority label
    g = make(PROB, 0.1)
    p, c = _z(4)
    g.integrate(p, c, np.array([5, 5, 5, 6], np.int32), np.array([1, 1, 1, 2], np.int32), None)
    _, _, cls, obj, conf = g.get_voxels(1, 0.0)
    assert len(obj) == 1 and obj[0] == 1 and cls[0] == 5 and conf[0] > 0.5
    # :60-80 depth decay down-weights the far observation
    g = make(PROB, 0.1)
    p, c = _z(2)
    g.integrate(p, c, np.array([7, 8], np.int32), np.array([3, 4], np.int32), np.array([1.0, 20.0], np.float32))
    _, _, cls, obj, conf = g.get_voxels(1, 0.0)
    assert len(obj) == 1 and obj[0] == 3 and cls[0] == 7 and conf[0] > 0.5
    # :104-125 strong majority 12:1
    g = make(PROB, 0.1)
    p, c = _z(13)
    g.integrate(p, c, np.array([5] * 12 + [6], np.int32), np.array([1] * 12 + [2], np.int32), None)
    _, _, cls, obj, conf = g.get_voxels(1, 0.0)
    assert obj[0] == 1 and cls[0] == 5 and conf[0] > 0.7
    # :128-153 label noise, probabilistic
    rng = np.random.default_rng(0)
    g = make(PROB, 0.2)
    pts = rng.uniform(0.0, 0.05, (55, 3)).astype(np.float64)
    cls_in = np.array([11] * 50 + [12] * 5, np.int32)
    inst_in = np.array([111] * 50 + [222] * 5, np.int32)
    perm = rng.permutation(55)
    g.integrate(pts[perm], np.zeros((55, 3), np.uint8), cls_in[perm], inst_in[perm], None)
    _, _, cls, obj, conf = g.get_voxels(1, 0.0)
    assert len(obj) == 1 and obj[0] == 111 and cls[0] == 11 and conf[0] > 0.75
    # :156-185 label noise, voting: confidence = (majority - noise) / total
    rng = np.random.default_rng(1)
    g = make(VOTE, 0.2)
    pts = rng.uniform(0.0, 0.05, (33, 3)).astype(np.float64)
    cls_in = np.array([21] * 30 + [22] * 3, np.int32)
    inst_in = np.array([210] * 30 + [220] * 3, np.int32)
    perm = rng.permutation(33)
    g.integrate(pts[perm], np.zeros((33, 3), np.uint8), cls_in[perm], inst_in[perm], None)
    _, _, cls, obj, conf = g.get_voxels(1, 0.0)
    assert len(obj) == 1 and obj[0] == 210 and cls[0] == 21
    assert conf[0] == pytest.approx(27 / 33.0, abs=1e-2)
    # :188-215 the joint distribution beats the marginals
    g = make(PROB, 0.1)
    pairs = [(1, 10)] * 3 + [(1, 11)] * 3 + [(2, 10)] * 4
    p, c = _z(len(pairs))
    perm = np.random.default_rng(42).permutation(len(pairs))
    g.integrate(p, c, np.array([q[1] for q in pairs], np.int32)[perm], np.array([q[0] for q in pairs], np.int32)[perm], None)
    _, _, cls, obj, conf = g.get_voxels(1, 0.0)
    assert len(obj) == 1 and (obj[0], cls[0]) == (2, 10)
    base_log = 0.10536051565782628  # softmax over the log-evidence, :226-230
    lp = np.array([4 * base_log, 3 * base_log, 3 * base_log])
    assert conf[0] == pytest.approx(np.exp(lp[0]) / np.exp(lp).sum(), rel=1e-4, abs=1e-4)
