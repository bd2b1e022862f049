"""CPU: the image-plane boxes of the `volumetric` mirror (pyslam_amd.bounding_boxes_2d.BoundingBox2D / OrientedBoundingBox2D) against

* the COMPILED reference classes (cpp/volumetric/bounding_boxes_2d.h/.cpp through oracle/_ref: getters, contains, the
  separating-axis intersects, corners, PCA compute_from_points) on seeded random inputs, and
* the scenarios and expected values of the reference's own unit tests for these classes
  (cpp/test_volumetric_bounding_boxes.py:545-826, 853-860, 882-886; the Qhull method is outside what the image can build).
Host code only."""
import ctypes as C
import math
import pickle

import numpy as np
import pytest

import oracle
from pyslam_amd.bounding_boxes_2d import BoundingBox2D, OrientedBoundingBox2D
from pyslam_amd.volumetric_semantic import OBBComputationMethod

vp, i64 = C.c_void_p, C.c_int64


def _p(a):
    return a.ctypes.data_as(vp)


@pytest.fixture(scope="module")
def ref():
    if not oracle.ref_available():
        pytest.skip("compiled reference not available")
    L = oracle.ref_lib()
    if not hasattr(L, "ref_aabb2_scalars"):
        pytest.skip("compiled reference predates the 2D box exports (rebuild oracle/_ref)")
    for name, args in (("ref_aabb2_scalars", [vp, vp]), ("ref_aabb2_contains", [vp, vp, i64, vp]), ("ref_aabb2_from_points", [vp, i64, vp]),
                       ("ref_obb2_scalars", [vp, vp, vp]), ("ref_obb2_contains", [vp, vp, i64, vp]), ("ref_obb2_from_points", [vp, i64, vp])):
        getattr(L, name).argtypes = args
        getattr(L, name).restype = None
    for name in ("ref_aabb2_intersects", "ref_obb2_intersects_obb", "ref_obb2_intersects_aabb"):
        getattr(L, name).argtypes = [vp, vp]
        getattr(L, name).restype = C.c_int
    return L


def test_aabb2_matches_compiled_reference(ref):
    rng = np.random.default_rng(0)
    for _ in range(50):
        lo = rng.normal(size=2)
        b4 = np.concatenate([lo, lo + rng.uniform(0.0, 2.0, size=2)])
        box = BoundingBox2D(b4[:2], b4[2:])
        want = np.zeros(7)
        ref.ref_aabb2_scalars(_p(b4), _p(want))
        got = np.concatenate([box.get_center(), box.get_size(), [box.get_area(), box.get_perimeter(), box.get_diagonal_length()]])
        np.testing.assert_array_equal(got, want)
        pts = np.ascontiguousarray(np.concatenate([rng.normal(size=(200, 2)) * 1.5, np.stack([b4[:2], b4[2:], box.get_center()])]))  # incl. points ON the edges
        m = np.zeros(len(pts), np.uint8)
        ref.ref_aabb2_contains(_p(b4), _p(pts), len(pts), _p(m))
        assert box.contains(pts) == [bool(x) for x in m]
        assert box.contains(pts[0]) == bool(m[0])
        lo2 = rng.normal(size=2)
        o4 = np.concatenate([lo2, lo2 + rng.uniform(0.0, 2.0, size=2)])
        assert box.intersects(BoundingBox2D(*o4)) == bool(ref.ref_aabb2_intersects(_p(b4), _p(o4)))
        cloud = np.ascontiguousarray(rng.normal(size=(37, 2)))
        w4 = np.zeros(4)
        ref.ref_aabb2_from_points(_p(cloud), len(cloud), _p(w4))
        c = BoundingBox2D.compute_from_points(cloud)
        np.testing.assert_array_equal([c.min_x, c.min_y, c.max_x, c.max_y], w4)


def test_obb2_matches_compiled_reference(ref):
    rng = np.random.default_rng(1)
    hits = 0
    for k in range(300):
        a5 = np.concatenate([rng.normal(size=2), [rng.uniform(-7.0, 7.0)], rng.uniform(0.05, 2.0, size=2)])
        b5 = np.concatenate([rng.normal(size=2), [rng.uniform(-7.0, 7.0)], rng.uniform(0.05, 2.0, size=2)])
        if k % 3 == 0:
            b5[:2] = a5[:2] + rng.normal(size=2) * 0.3  # close pairs: both outcomes of the SAT are exercised
        a, b = OrientedBoundingBox2D(a5[:2], a5[2], a5[3:]), OrientedBoundingBox2D(b5[:2], b5[2], b5[3:])
        s, cs = np.zeros(4), np.zeros(8)
        ref.ref_obb2_scalars(_p(a5), _p(s), _p(cs))
        np.testing.assert_array_equal([a.get_volume(), a.get_area(), a.get_perimeter(), a.get_diagonal_length()], s)
        np.testing.assert_allclose(np.stack(a.get_corners()), cs.reshape(4, 2), rtol=0, atol=1e-14)
        pts = np.ascontiguousarray(np.concatenate([a5[:2] + rng.normal(size=(100, 2)) * a5[3:].max(), np.stack(a.get_corners())]))
        m = np.zeros(len(pts), np.uint8)
        ref.ref_obb2_contains(_p(a5), _p(pts), len(pts), _p(m))
        got = a.contains(pts)
        # corners sit ON the boundary within the 1e-10 slack of both; an interior / exterior point whose margin is > 1e-9 must agree
        assert got[:100] == [bool(x) for x in m[:100]]
        assert all(got[100:]) and all(m[100:])
        hit = bool(ref.ref_obb2_intersects_obb(_p(a5), _p(b5)))
        assert a.intersects(b) == hit
        hits += hit
        lo = rng.normal(size=2)
        q4 = np.concatenate([lo, lo + rng.uniform(0.1, 2.0, size=2)])
        assert a.intersects(BoundingBox2D(*q4)) == bool(ref.ref_obb2_intersects_aabb(_p(a5), _p(q4)))
    assert 30 < hits < 270
    for n in (2, 3, 17, 400):
        cloud = np.ascontiguousarray(rng.normal(size=(n, 2)) * np.array([3.0, 0.7]) @ np.array([[0.8, -0.6], [0.6, 0.8]]))
        w5 = np.zeros(5)
        ref.ref_obb2_from_points(_p(cloud), n, _p(w5))
        o = OrientedBoundingBox2D.compute_from_points(cloud)
        # same box as a point set; the principal axis' sign is the eigen-solver's (angle may differ by pi)
        d = (o.angle_rad - w5[2] + math.pi / 2) % math.pi - math.pi / 2
        assert abs(d) < 1e-9
        np.testing.assert_allclose(o.center, w5[:2], atol=1e-9)
        np.testing.assert_allclose(o.size, w5[3:], atol=1e-9)
        assert all(OrientedBoundingBox2D(o.center, o.angle_rad, o.size + 1e-9).contains(cloud))


def test_reference_unit_test_scenarios():
    """cpp/test_volumetric_bounding_boxes.py:545-590, 647-718, 853-860, 882-886 (expected values as written there)."""
    bbox = BoundingBox2D(np.array([0.0, 0.0]), np.array([1.0, 1.0]))
    np.testing.assert_allclose(bbox.get_min_point(), [0.0, 0.0])
    np.testing.assert_allclose(bbox.get_max_point(), [1.0, 1.0])
    np.testing.assert_allclose(bbox.get_center(), [0.5, 0.5])
    np.testing.assert_allclose(bbox.get_size(), [1.0, 1.0])
    assert bbox.get_area() == 1.0 and bbox.get_perimeter() == 4.0 and abs(bbox.get_diagonal_length() - math.sqrt(2.0)) < 1e-12
    assert bbox.contains(np.array([0.5, 0.5])) is True and bbox.contains(np.array([1.5, 0.5])) is False
    assert bbox.intersects(BoundingBox2D(np.array([0.5, 0.5]), np.array([2.0, 2.0]))) is True
    pts = [np.array([0.0, 0.0]), np.array([2.0, 1.0]), np.array([1.0, 3.0])]
    b = BoundingBox2D.compute_from_points(pts)
    np.testing.assert_allclose(b.get_min_point(), [0.0, 0.0])
    np.testing.assert_allclose(b.get_max_point(), [2.0, 3.0])
    assert all(b.contains(p) for p in pts)
    line = BoundingBox2D.compute_from_points([np.array([float(i), 0.0]) for i in range(5)])
    assert line.get_area() == 0.0 and line.get_size()[0] == 4.0

    obb = OrientedBoundingBox2D(np.array([0.0, 0.0]), 0.0, np.array([2.0, 2.0]))
    assert obb.get_area() == 4.0 and obb.get_perimeter() == 8.0 and abs(obb.get_diagonal_length() - math.sqrt(8.0)) < 1e-12
    assert len(obb.get_corners()) == 4
    assert obb.contains(np.array([0.0, 0.0])) and obb.contains(np.array([0.9, 0.9])) and not obb.contains(np.array([1.1, 0.0]))
    assert OrientedBoundingBox2D(np.array([0.0, 0.0]), math.pi / 4.0, np.array([2.0, 2.0])).contains(np.array([0.707, 0.707]))
    rect = [np.array([0.0, 0.0]), np.array([2.0, 0.0]), np.array([2.0, 1.0]), np.array([0.0, 1.0])]
    o = OrientedBoundingBox2D.compute_from_points(rect)
    assert abs(o.get_area() - 2.0) < 0.05 * 2.0 and all(o.contains(p) for p in rect)
    c, s = math.cos(math.pi / 4.0), math.sin(math.pi / 4.0)
    R = np.array([[c, -s], [s, c]])
    assert abs(OrientedBoundingBox2D.compute_from_points([R @ p for p in rect]).get_area() - 2.0) < 0.05 * 2.0
    o1 = OrientedBoundingBox2D(np.array([0.0, 0.0]), 0.0, np.array([2.0, 2.0]))
    assert o1.intersects(OrientedBoundingBox2D(np.array([1.0, 0.0]), 0.0, np.array([2.0, 2.0]))) is True
    assert o1.intersects(BoundingBox2D(np.array([0.5, 0.5]), np.array([1.5, 1.5]))) is True
    assert not o1.intersects(OrientedBoundingBox2D(np.array([5.0, 0.0]), 0.3, np.array([2.0, 2.0])))
    col = OrientedBoundingBox2D.compute_from_points([np.array([float(i), float(i)]) for i in range(5)])
    assert col.get_area() >= 0.0 and abs(col.size.max() - 4.0 * math.sqrt(2.0)) < 1e-9 and col.size.min() < 1e-9
    # degenerate inputs (bounding_boxes_2d.cpp:236-246), the unbuilt method, pickling (the binding's py::pickle)
    e = OrientedBoundingBox2D.compute_from_points(np.zeros((0, 2)))
    assert e.get_area() == 0.0 and e.angle_rad == 0.0
    one = OrientedBoundingBox2D.compute_from_points([np.array([3.0, -1.0])])
    assert list(one.center) == [3.0, -1.0] and list(one.size) == [0.0, 0.0]
    with pytest.raises(NotImplementedError):
        OrientedBoundingBox2D.compute_from_points(rect, OBBComputationMethod.CONVEX_HULL_MINIMAL)
    back = pickle.loads(pickle.dumps(o))
    assert list(back.center) == list(o.center) and back.angle_rad == o.angle_rad and list(back.size) == list(o.size)
    bb = pickle.loads(pickle.dumps(bbox))
    assert (bb.min_x, bb.min_y, bb.max_x, bb.max_y) == (0.0, 0.0, 1.0, 1.0)
