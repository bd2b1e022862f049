"""CPU: the host-side CameraFrustrum class (pyslam_amd.volumetric) against the compiled reference's (oracle/_ref, unmodified
cpp/volumetric/camera_frustrum.cpp) over the surface the module binds (camera_frustrum_module.h:50-130): the three constructors,
setters and the cache flag, K / R_cw / t_cw / orientation, the eight corners, the oriented box, is_in_bbox / is_in_obb / contains
with its ImagePoint - on seeded random poses and points, and the calls of the reference's own cpp/test_volumetric.py:263-412."""
import numpy as np
import pytest

import oracle
from pyslam_amd.volumetric import CameraFrustrum, ImagePoint

pytestmark = pytest.mark.skipif(not oracle.ref_available(), reason="compiled reference not available")


def random_pose(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = rng.uniform(-2, 2, 3)
    return T, q


def check(fr, ref, pts):
    np.testing.assert_allclose(np.array(fr.get_corners()), ref["corners"], rtol=0, atol=4e-15)  # (float64, 1-2 ulp: Eigen's product order)
    np.testing.assert_array_equal(fr.get_K(), ref["K"])
    np.testing.assert_array_equal(fr.get_R_cw(), ref["R_cw"])
    np.testing.assert_array_equal(fr.get_t_cw(), ref["t_cw"])
    q, qr = fr.get_orientation_cw().coeffs(), ref["orientation_cw"]
    np.testing.assert_allclose(q, qr, rtol=0, atol=1e-15)
    obb = fr.get_obb()
    np.testing.assert_allclose(obb.center, ref["obb"][0:3], rtol=0, atol=1e-12)
    np.testing.assert_allclose(obb.orientation, ref["obb"][3:7], rtol=0, atol=1e-15)
    np.testing.assert_allclose(obb.size, ref["obb"][7:10], rtol=0, atol=1e-12)
    bb = fr.get_bbox()
    np.testing.assert_allclose(bb.get_min_point(), ref["corners"].min(0), rtol=0, atol=4e-15)
    np.testing.assert_allclose(bb.get_max_point(), ref["corners"].max(0), rtol=0, atol=4e-15)
    inside = np.zeros(len(pts), np.uint8)
    uvd = np.zeros((len(pts), 3), np.float32)
    for i, p in enumerate(pts):
        ok, ip = fr.contains(p)
        inside[i] = ok
        uvd[i] = (ip.u, ip.v, ip.depth)
    np.testing.assert_array_equal(inside, ref["inside"])
    np.testing.assert_array_equal(uvd.view(np.uint32), ref["uvd"].view(np.uint32))
    # (a point within rounding distance of a face of a box may fall on either side: the boxes agree to 1e-14, not to the bit)
    assert (np.array([fr.is_in_bbox(p) for p in pts], np.uint8) != ref["in_bbox"]).mean() <= 0.002
    assert (np.array([fr.is_in_obb(p) for p in pts], np.uint8) != ref["in_obb"]).mean() <= 0.002
    return inside


def test_frustum_surface_equals_the_compiled_reference():
    rng = np.random.default_rng(11)
    seen_inside = 0
    for trial in range(12):
        T, q = random_pose(rng)
        intr = np.float32([rng.uniform(300, 700), rng.uniform(300, 700), rng.uniform(200, 400), rng.uniform(150, 300)])
        W, H = int(rng.integers(320, 1300)), int(rng.integers(240, 1000))
        dmax, dmin = float(np.float32(rng.uniform(3, 10))), float(np.float32(rng.uniform(0.01, 0.5)))
        # points: half drawn inside the viewing cone (camera frame -> world), half anywhere around the camera
        z = rng.uniform(0.0, dmax * 1.2, 2000)
        pc = np.stack([(rng.uniform(-0.2 * W, 1.2 * W, 2000) - intr[2]) / intr[0] * z, (rng.uniform(-0.2 * H, 1.2 * H, 2000) - intr[3]) / intr[1] * z, z], 1)
        pts = np.concatenate([(pc - T[:3, 3]) @ T[:3, :3], rng.uniform(-12, 12, (1000, 3))])
        if trial % 3 == 0:  # K constructor
            fr = CameraFrustrum(np.array([[intr[0], 0, intr[2]], [0, intr[1], intr[3]], [0, 0, 1]], np.float64), W, H, T, dmax, dmin)
            ref = oracle.ref_frustum_surface(intr, W, H, T, dmax, dmin, pts)
        elif trial % 3 == 1:  # (orientation, translation) constructor: the quaternion is normalised there
            q2 = q * 1.7
            fr = CameraFrustrum(*intr, W, H, orientation=q2, translation=T[:3, 3], depth_max=dmax, depth_min=dmin)
            ref = oracle.ref_frustum_surface(intr, W, H, None, dmax, dmin, pts, orientation=q2, translation=T[:3, 3])
            np.testing.assert_allclose(fr.get_R_cw(), ref["R_cw"], rtol=0, atol=1e-15)
            fr.set_T_cw(np.vstack([np.hstack([ref["R_cw"], ref["t_cw"][:, None]]), [0, 0, 0, 1]]))  # the same bits from here on
        else:
            fr = CameraFrustrum(*intr, W, H, T, depth_max=dmax, depth_min=dmin)
            ref = oracle.ref_frustum_surface(intr, W, H, T, dmax, dmin, pts)
        seen_inside += int(check(fr, ref, pts).sum())
    assert seen_inside > 5000


def test_setters_and_cache_flag_like_the_reference_class():
    T, _ = random_pose(np.random.default_rng(5))
    fr = CameraFrustrum(525.0, 525.0, 319.5, 239.5, 640, 480, T, 10.0, 0.01)
    assert not fr.is_cache_valid()  # (the reference constructor fills the cache; nothing observable depends on it)
    fr.get_corners()
    assert fr.is_cache_valid()
    pts = np.random.default_rng(6).uniform(-6, 6, (500, 3))
    for change in (lambda: fr.set_width(800), lambda: fr.set_height(600), lambda: fr.set_depth_max(4.0), lambda: fr.set_depth_min(0.3),
                   lambda: fr.set_intrinsics(500.0, 510.0, 400.0, 300.0), lambda: fr.set_intrinsics(np.array([[450.0, 0, 390.0], [0, 460.0, 310.0], [0, 0, 1]])),
                   lambda: fr.set_T_cw(np.eye(4))):
        change()
        assert not fr.is_cache_valid()
        ref = oracle.ref_frustum_surface(fr.intr, fr.get_width(), fr.get_height(), fr.get_T_cw(), fr.depth_max, fr.depth_min, pts)
        check(fr, ref, pts)
        assert fr.is_cache_valid()
    assert (fr.get_fx(), fr.get_fy(), fr.get_cx(), fr.get_cy()) == (450.0, 460.0, 390.0, 310.0)


def test_calls_of_the_reference_api_script():
    """cpp/test_volumetric.py:263-412: construct from scalars and from K, getters, a point in front of the camera is contained, one
    behind it is not and comes back as ImagePoint(-1, -1, -1)."""
    fr = CameraFrustrum(fx=500.0, fy=500.0, cx=320.0, cy=240.0, width=640, height=480, T_cw=np.eye(4), depth_max=10.0, depth_min=0.1)
    assert (fr.get_width(), fr.get_height()) == (640, 480) and fr.get_K()[0, 2] == 320.0
    ok, ip = fr.contains(np.array([0.0, 0.0, 5.0]))
    assert ok and (ip.u, ip.v, ip.depth) == (320.0, 240.0, 5.0)
    ok, ip = fr.contains(np.array([0.0, 0.0, -1.0]))
    assert not ok and (ip.u, ip.v, ip.depth) == (-1.0, -1.0, -1.0)
    assert fr.is_in_bbox(np.array([0.0, 0.0, 5.0])) and fr.is_in_obb(np.array([0.0, 0.0, 5.0]))
    assert len(fr.get_corners()) == 8 and fr.get_obb().size[2] == pytest.approx(9.9, abs=1e-6)
    p = ImagePoint(3, 4, 1.5)
    assert (p.u, p.v, p.depth) == (3.0, 4.0, 1.5)
