"""CPU: the 3D box classes of the `volumetric` mirror (pyslam_amd.volumetric.BoundingBox3D,
pyslam_amd.volumetric_semantic.OrientedBoundingBox3D: what get_voxels_in_bb takes and get_object_segments returns) against

* the COMPILED reference classes (cpp/volumetric/bounding_boxes_3d.h/.cpp through oracle/_ref: getters, contains, the
  separating-axis intersects of box pairs, corners, matrices, PCA compute_from_points) on seeded random inputs, and
* the scenarios and expected values of the reference's own unit tests for these classes
  (cpp/test_volumetric_bounding_boxes.py:67-543, 865-1250: the 3D / PCA cases; its 2D boxes: tests/test_bounding_boxes_2d_cpu.py; the Qhull method is outside
  the path).
Host code only - the PCA entry point hv_compute_obb_pca is a host function of the C ABI."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle
from pyslam_amd.volumetric import BoundingBox3D
from pyslam_amd.volumetric_semantic import OBBComputationMethod, OrientedBoundingBox3D

vp, i64 = C.c_void_p, C.c_int64


def _p(a):
    return a.ctypes.data_as(vp)


@pytest.fixture(scope="module")
def ref():
    if not oracle.ref_available():
        pytest.skip("compiled reference not available")
    L = oracle.ref_lib()
    for name, args in (("ref_aabb3_scalars", [vp, vp]), ("ref_aabb3_contains", [vp, vp, i64, vp]), ("ref_aabb3_from_points", [vp, i64, vp]),
                       ("ref_obb3_scalars", [vp, vp, vp, vp, vp]), ("ref_obb3_contains", [vp, vp, i64, vp]),
                       ("ref_obb3_from_points", [vp, i64, vp])):
        getattr(L, name).argtypes = args
        getattr(L, name).restype = None
    for name, args in (("ref_aabb3_intersects", [vp, vp]), ("ref_obb3_intersects_obb", [vp, vp]), ("ref_obb3_intersects_aabb", [vp, vp])):
        getattr(L, name).argtypes = args
        getattr(L, name).restype = C.c_int
    return L


def random_obb(rng, scale=1.0):
    q = rng.normal(size=4)
    return np.concatenate([rng.normal(size=3) * scale, q * rng.uniform(0.2, 3.0), rng.uniform(0.05, 2.0, size=3) * scale])  # un-normalised quaternion on purpose


def test_aabb_matches_compiled_reference(ref):
    rng = np.random.default_rng(0)
    for _ in range(50):
        lo = rng.normal(size=3)
        b6 = np.concatenate([lo, lo + rng.uniform(0.0, 2.0, size=3)])
        box = BoundingBox3D(b6[:3], b6[3:])
        want = np.zeros(9)
        ref.ref_aabb3_scalars(_p(b6), _p(want))
        got = np.concatenate([box.get_center(), box.get_size(), [box.get_volume(), box.get_surface_area(), box.get_diagonal_length()]])
        np.testing.assert_array_equal(got, want)
        pts = np.ascontiguousarray(np.concatenate([rng.normal(size=(200, 3)) * 1.5, np.stack([b6[:3], b6[3:], box.get_center()])]))  # incl. points ON the faces
        m = np.zeros(len(pts), np.uint8)
        ref.ref_aabb3_contains(_p(b6), _p(pts), len(pts), _p(m))
        assert box.contains(pts) == [bool(x) for x in m]
        assert box.contains(pts[0]) == bool(m[0])
        lo2 = rng.normal(size=3)
        o6 = np.concatenate([lo2, lo2 + rng.uniform(0.0, 2.0, size=3)])
        assert box.intersects(BoundingBox3D(*o6)) == bool(ref.ref_aabb3_intersects(_p(b6), _p(o6)))
        cloud = np.ascontiguousarray(rng.normal(size=(37, 3)))
        w6 = np.zeros(6)
        ref.ref_aabb3_from_points(_p(cloud), len(cloud), _p(w6))
        np.testing.assert_array_equal(BoundingBox3D.compute_from_points(cloud).as_array(), w6)
    assert BoundingBox3D.compute_from_points(np.zeros((0, 3))).as_array().tolist() == [0.0] * 6
    assert BoundingBox3D().as_array().tolist() == [0.0] * 6


def test_obb_matches_compiled_reference(ref):
    rng = np.random.default_rng(1)
    n_hit = 0
    for k in range(300):
        a10, b10 = random_obb(rng), random_obb(rng)
        if k % 3 == 0:
            b10[:3] = a10[:3] + rng.normal(size=3) * 0.3  # close pairs: both outcomes of the SAT are exercised
        A = OrientedBoundingBox3D(a10[:3], a10[3:7], a10[7:])
        B = OrientedBoundingBox3D(b10[:3], b10[3:7], b10[7:])
        sc, M, Mi, cs = np.zeros(3), np.zeros(16), np.zeros(16), np.zeros(24)
        ref.ref_obb3_scalars(_p(a10), _p(sc), _p(M), _p(Mi), _p(cs))
        np.testing.assert_allclose([A.get_volume(), A.get_surface_area(), A.get_diagonal_length()], sc, rtol=1e-15)
        np.testing.assert_allclose(A.get_matrix().ravel(), M, atol=1e-14)
        np.testing.assert_allclose(A.get_inverse_matrix().ravel(), Mi, atol=1e-14)
        np.testing.assert_allclose(A.get_corners().ravel(), cs, atol=1e-13)  # same corner ORDER
        pts = np.ascontiguousarray(np.concatenate([a10[:3] + rng.normal(size=(150, 3)) * 0.8, A.get_corners(), a10[None, :3]]))
        m = np.zeros(len(pts), np.uint8)
        ref.ref_obb3_contains(_p(a10), _p(pts), len(pts), _p(m))
        mine = A.contains(pts)
        # a point within 1e-9 of a face may fall on either side of the two roundings; everything else must agree
        q = np.abs((pts - A.center) @ A.get_rotation_matrix()) - A.size / 2.0
        clear = np.abs(q).min(axis=1) > 1e-9
        assert [x for x, c in zip(mine, clear) if c] == [bool(x) for x, c in zip(m, clear) if c]
        assert clear.sum() > 100
        hit = bool(ref.ref_obb3_intersects_obb(_p(a10), _p(b10)))
        assert A.intersects(B) == hit
        n_hit += hit
        lo = a10[:3] + rng.normal(size=3) * 0.5
        c6 = np.concatenate([lo, lo + rng.uniform(0.1, 1.5, size=3)])
        assert A.intersects(BoundingBox3D(c6[:3], c6[3:])) == bool(ref.ref_obb3_intersects_aabb(_p(a10), _p(c6)))
    assert 50 < n_hit < 250


def test_obb_pca_matches_compiled_reference(ref):
    """compute_from_points (PCA) through the C ABI's host entry point against the reference's: same box as geometry (the
    eigenvector signs / order of equal eigenvalues are free: compared through volume, centre and containment)."""
    rng = np.random.default_rng(2)
    for n in (1, 2, 3, 8, 50, 500):
        R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        cloud = np.ascontiguousarray((rng.uniform(-1, 1, size=(n, 3)) * np.array([3.0, 1.0, 0.3])) @ R.T + rng.normal(size=3))
        want = np.zeros(10)
        ref.ref_obb3_from_points(_p(cloud), n, _p(want))
        got = OrientedBoundingBox3D.compute_from_points(cloud)
        W = OrientedBoundingBox3D(want[:3], want[3:7], want[7:])
        np.testing.assert_allclose(got.center, W.center, atol=1e-9)
        np.testing.assert_allclose(np.sort(got.size), np.sort(W.size), atol=1e-9)
        if n >= 3:
            assert all(got.contains(cloud)) and all(W.contains(cloud))
            np.testing.assert_allclose(np.sort(got.get_corners(), axis=0), np.sort(W.get_corners(), axis=0), atol=1e-8)
    with pytest.raises(NotImplementedError):
        OrientedBoundingBox3D.compute_from_points(np.zeros((4, 3)), OBBComputationMethod.CONVEX_HULL_MINIMAL)


# ---- the reference's own unit-test scenarios (cpp/test_volumetric_bounding_boxes.py) ----------------------------------
def box_points():
    return np.array([[x, y, z] for x in (0.0, 2.0) for y in (0.0, 1.0) for z in (0.0, 0.5)])


def rot_z(deg):
    a = math.radians(deg)
    return np.array([[math.cos(a), -math.sin(a), 0.0], [math.sin(a), math.cos(a), 0.0], [0.0, 0.0, 1.0]])


def test_reference_unit_test_scenarios_aabb():
    """:67-216, :518-543."""
    bbox = BoundingBox3D(np.array([0.0, 0.0, 0.0]), np.array([1.0, 2.0, 3.0]))
    np.testing.assert_allclose(bbox.get_min_point(), [0, 0, 0])
    np.testing.assert_allclose(bbox.get_max_point(), [1, 2, 3])
    np.testing.assert_allclose(bbox.get_center(), [0.5, 1.0, 1.5])
    np.testing.assert_allclose(bbox.get_size(), [1, 2, 3])
    assert bbox.get_volume() == pytest.approx(6.0) and bbox.get_surface_area() == pytest.approx(22.0)
    assert bbox.get_diagonal_length() == pytest.approx(math.sqrt(14.0))
    assert bbox.contains(np.array([0.5, 1.0, 1.5])) and bbox.contains(np.array([0.0, 0.0, 0.0])) and bbox.contains(np.array([1.0, 2.0, 3.0]))
    assert not bbox.contains(np.array([1.5, 1.0, 1.5])) and not bbox.contains(np.array([-0.1, 1.0, 1.5]))
    assert bbox.contains([np.array([0.5, 1.0, 1.5]), np.array([1.5, 1.0, 1.5]), np.array([0.0, 0.0, 0.0])]) == [True, False, True]
    b6 = BoundingBox3D(0.0, 0.0, 0.0, 1.0, 2.0, 3.0)
    assert b6.as_array().tolist() == bbox.as_array().tolist()
    pts = [np.array(p) for p in ([0.0, 0.0, 0.0], [1.0, 2.0, 3.0], [-1.0, 0.5, 1.5], [0.5, -0.5, 2.0])]
    c = BoundingBox3D.compute_from_points(pts)
    assert c.as_array().tolist() == [-1.0, -0.5, 0.0, 1.0, 2.0, 3.0]
    a = BoundingBox3D(np.zeros(3), np.ones(3))
    assert a.intersects(BoundingBox3D(np.full(3, 0.5), np.full(3, 1.5)))      # overlapping
    assert not a.intersects(BoundingBox3D(np.full(3, 2.0), np.full(3, 3.0)))  # disjoint
    assert a.intersects(BoundingBox3D(np.array([1.0, 0.0, 0.0]), np.array([2.0, 1.0, 1.0])))  # touching faces count


def test_reference_unit_test_scenarios_obb():
    """:220-435, :829-863, :865-1250 (3D / PCA cases)."""
    obb = OrientedBoundingBox3D(np.zeros(3), np.array([1.0, 0.0, 0.0, 0.0]), np.full(3, 2.0))
    assert obb.get_volume() == pytest.approx(8.0) and obb.get_surface_area() == pytest.approx(24.0)
    assert obb.get_diagonal_length() == pytest.approx(math.sqrt(12.0))
    corners = obb.get_corners()
    assert len(corners) == 8
    np.testing.assert_allclose(corners.mean(axis=0), 0.0, atol=1e-5)
    assert obb.contains(np.zeros(3)) and obb.contains(np.full(3, 0.9)) and not obb.contains(np.array([1.1, 0.0, 0.0]))
    a = math.pi / 4.0
    rot = OrientedBoundingBox3D(np.zeros(3), np.array([math.cos(a / 2), 0.0, 0.0, math.sin(a / 2)]), np.full(3, 2.0))
    assert rot.contains(np.array([0.707, 0.707, 0.0]))
    pts = box_points()
    pca = OrientedBoundingBox3D.compute_from_points(pts)
    assert pca.get_volume() == pytest.approx(1.0, abs=0.05) and all(pca.contains(pts))
    grid = np.array([[x, y, z] for x in (0.0, 1.0, 2.0) for y in (0.0, 1.0) for z in (0.0, 0.5)])
    for deg in (15, 30, 45, 60, 90):
        r = grid @ rot_z(deg).T
        o = OrientedBoundingBox3D.compute_from_points(r)
        assert o.get_volume() == pytest.approx(1.0, abs=0.05) and all(o.contains(r))
    ax, ay = math.radians(30), math.radians(45)
    Rx = np.array([[1, 0, 0], [0, math.cos(ax), -math.sin(ax)], [0, math.sin(ax), math.cos(ax)]])
    Ry = np.array([[math.cos(ay), 0, math.sin(ay)], [0, 1, 0], [-math.sin(ay), 0, math.cos(ay)]])
    r3 = grid @ (Ry @ Rx).T
    o3 = OrientedBoundingBox3D.compute_from_points(r3)
    assert o3.get_volume() == pytest.approx(1.0, abs=0.05) and all(o3.contains(r3))
    big = OrientedBoundingBox3D.compute_from_points(grid * 10.0)
    assert big.get_volume() == pytest.approx(1000.0, abs=50.0)
    sphere = np.array([[math.sin(math.pi * i / 20) * math.cos(2 * math.pi * i / 20), math.sin(math.pi * i / 20) * math.sin(2 * math.pi * i / 20),
                        math.cos(math.pi * i / 20)] for i in range(20)])
    os_ = OrientedBoundingBox3D.compute_from_points(sphere)
    assert 0.0 < os_.get_volume() < 8.0 and all(os_.contains(sphere))
    elong = np.array([[x, y, z] for x in np.linspace(-5.0, 5.0, 20) for y in (-0.1, 0.1) for z in (-0.1, 0.1)])
    oe = OrientedBoundingBox3D.compute_from_points(elong)
    assert max(oe.size) > 5.0 and min(oe.size) < 1.0 and all(oe.contains(elong))
    np.random.seed(42)
    rnd = np.array([np.random.randn(3) * 5.0 for _ in range(50)])
    orr = OrientedBoundingBox3D.compute_from_points(rnd)
    assert orr.get_volume() > 0.0 and all(orr.contains(rnd))
    # intersects (:829-863): overlapping, separated, and against an axis-aligned box
    ident = np.array([1.0, 0.0, 0.0, 0.0])
    o1 = OrientedBoundingBox3D(np.zeros(3), ident, np.full(3, 2.0))
    assert o1.intersects(OrientedBoundingBox3D(np.full(3, 0.5), ident, np.full(3, 2.0)))
    assert not o1.intersects(OrientedBoundingBox3D(np.full(3, 5.0), ident, np.full(3, 2.0)))
    assert o1.intersects(BoundingBox3D(np.full(3, 0.5), np.full(3, 1.5))) and not o1.intersects(BoundingBox3D(np.full(3, 3.0), np.full(3, 4.0)))
    # degenerate inputs (:865-900): no points, one point, two points
    e = OrientedBoundingBox3D.compute_from_points(np.zeros((0, 3)))
    assert e.get_volume() == 0.0
    one = OrientedBoundingBox3D.compute_from_points(np.array([[1.0, 2.0, 3.0]]))
    np.testing.assert_allclose(one.center, [1.0, 2.0, 3.0])
    assert one.get_volume() == 0.0
    two = OrientedBoundingBox3D.compute_from_points(np.array([[0.0, 0.0, 0.0], [2.0, 0.0, 0.0]]))
    np.testing.assert_allclose(two.center, [1.0, 0.0, 0.0], atol=1e-12)
    assert max(two.size) == pytest.approx(2.0)
