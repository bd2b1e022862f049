"""CPU: the voting semantic payload restatement (oracle/semantic_oracle.c) against
(a) the reference's own known-answer tests for the voting voxel — the only numeric pins the reference
    holds for this path (cpp/test_volumetric_voxel_semantic.py:20-36 label switch + confidence 0.5,
    :79-97 labels preserved across voxels), replayed on the block grid;
(b) the compiled reference (oracle/_ref) on seeded random streams, all dtype/optional-array variants;
(c) the committed golden fixture generated from the compiled reference (tests/golden/semantic_vote.npz)."""
import os

import numpy as np
import pytest

import oracle
from oracle.semantic import PortSemGrid, RefSemGrid

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "semantic_vote.npz")


def stream(seed, n, pos_dtype=np.float32):
    rng = np.random.default_rng(seed)
    pts = ((rng.random((n, 3)) - 0.5) * 2.0).astype(pos_dtype)
    cols = rng.integers(0, 255, (n, 3)).astype(np.uint8)
    cls = rng.integers(0, 5, n).astype(np.int32)
    inst = rng.integers(0, 4, n).astype(np.int32)
    dep = (rng.random(n) * 20).astype(np.float32)
    return pts, cols, cls, inst, dep


def srt(v):
    i = np.lexsort(v[0].T[::-1])
    return tuple(a[i] for a in v)


def test_reference_kat_label_switch_and_confidence():
    g = PortSemGrid(0.1, 8)
    g.integrate(np.zeros((2, 3), np.float64), np.zeros((2, 3), np.uint8), np.array([1, 2], np.int32), np.array([1, 2], np.int32))
    pts, cols, cls, obj, conf = g.get_voxels(min_count=1, min_confidence=0.0)
    assert len(obj) == 1 and obj[0] == 2 and cls[0] == 2
    assert conf[0] == pytest.approx(0.5, abs=1e-3)


def test_reference_kat_labels_preserved_across_voxels():
    g = PortSemGrid(0.1, 8)
    g.integrate(np.array([[0.0, 0.0, 0.0], [0.2, 0.0, 0.0]]), np.zeros((2, 3), np.uint8), np.array([10, 20], np.int32),
                np.array([101, 202], np.int32))
    pts, cols, cls, obj, conf = srt(g.get_voxels(1, 0.0))
    assert len(obj) == 2
    assert (obj[0], cls[0]) == (101, 10) and (obj[1], cls[1]) == (202, 20)


def test_depth_gate_and_default_object_id():
    g = PortSemGrid(0.1, 8)
    # far first observation does not initialise the label; the near one then does
    g.integrate(np.zeros((2, 3), np.float32), np.zeros((2, 3), np.uint8), np.array([7, 8], np.int32), None,
                np.array([20.0, 1.0], np.float32))
    _, _, cls, obj, conf = g.get_voxels(1, 0.0)
    assert cls[0] == 8 and obj[0] == 0 and conf[0] == pytest.approx(0.5)


@pytest.mark.skipif(not oracle.ref_available(), reason="compiled reference not available")
@pytest.mark.parametrize("pos_dtype", [np.float32, np.float64])
@pytest.mark.parametrize("use_inst,use_depth", [(True, True), (True, False), (False, True), (False, False)])
def test_port_matches_compiled_reference(pos_dtype, use_inst, use_depth):
    p, r = PortSemGrid(0.05, 8), RefSemGrid(0.05, 8)
    for g in (p, r):  # the threshold is a static of the payload class: whatever ran before in this process must not leak in
        g.set_depth_threshold(10.0)
    for it in range(3):
        pts, cols, cls, inst, dep = stream(100 + it, 20000, pos_dtype)
        c = cols if it != 1 else (cols / 255.0).astype(np.float32)
        for g in (p, r):
            g.integrate(pts, c, cls, inst if use_inst else None, dep if use_depth else None)
    for a, b in zip(p.dump(), r.dump()):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(srt(p.get_voxels(2, 0.5)), srt(r.get_voxels(2, 0.5))):
        np.testing.assert_array_equal(a, b)


def test_port_matches_reference_golden():
    z = np.load(GOLD)
    g = PortSemGrid(0.05, 8)
    for it in range(2):
        pts, cols, cls, inst, dep = stream(200 + it, 6000)
        g.integrate(pts, cols, cls, inst, dep)
    keys, ints, pos, col = g.dump()
    occ = ints[..., 0] > 0
    np.testing.assert_array_equal(keys, z["keys"])
    np.testing.assert_array_equal(ints[occ], z["ints"])
    np.testing.assert_array_equal(pos[occ], z["pos"])
    np.testing.assert_array_equal(col[occ], z["col"])
    v = srt(g.get_voxels(2, 0.6))
    for a, name in zip(v, ("v_pts", "v_cols", "v_cls", "v_obj", "v_conf")):
        np.testing.assert_array_equal(a, z[name])
