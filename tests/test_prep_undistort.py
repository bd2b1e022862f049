"""Undistortion.  CPU: the map restatement is self-consistent (distort o undistort = id,
zero distortion -> identity maps, alpha interpolation bounds).  GPU: remapping a *distorted* analytic
rendering with the maps reproduces the ideal pinhole rendering with the new camera matrix — the
geometric property cv2.initUndistortRectifyMap + cv2.remap exist to deliver (cv2 itself is absent)."""
import numpy as np
import pytest

from pyslam_amd import prep

# TUM1-like intrinsics and distortion (settings/TUM1.yaml:27-36)
K = np.array([[517.306408, 0, 318.643040], [0, 516.469215, 255.313989], [0, 0, 1.0]])
D = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314])
W, H = 640, 480


def test_distort_undistort_roundtrip():
    rng = np.random.default_rng(0)
    x, y = (rng.random(1000) - 0.5) * 0.9, (rng.random(1000) - 0.5) * 0.7
    xd, yd = prep.distort_normalized(x, y, D)
    u, v = K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]
    xr, yr = prep.undistort_points_normalized(u, v, K, D, iters=30)
    assert np.abs(xr - x).max() < 1e-6 and np.abs(yr - y).max() < 1e-6
    x5, y5 = prep.undistort_points_normalized(u, v, K, D)  # OpenCV's 5 iterations: close, not exact
    assert np.abs(x5 - x).max() < 5e-3


def test_zero_distortion_gives_identity_maps():
    mx, my = prep.init_undistort_rectify_map(K, np.zeros(5), K, (W, H))
    uu, vv = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    assert np.abs(mx - uu).max() < 1e-3 and np.abs(my - vv).max() < 1e-3


def test_optimal_new_camera_matrix_alpha_range():
    k0, _ = prep.get_optimal_new_camera_matrix(K, D, (W, H), 0.0, (W, H))
    k1, _ = prep.get_optimal_new_camera_matrix(K, D, (W, H), 1.0, (W, H))
    k7, _ = prep.get_optimal_new_camera_matrix(K, D, (W, H), 0.7, (W, H))
    assert k0[0, 0] > k1[0, 0] and k0[1, 1] > k1[1, 1]  # alpha=0 zooms in (valid pixels only), alpha=1 keeps all
    np.testing.assert_allclose(k7, 0.3 * k0 + 0.7 * k1, atol=1e-9)
    mx, my = prep.init_undistort_rectify_map(K, D, k0, (W, H))
    assert mx.min() >= -1.0 and mx.max() <= W and my.min() >= -1.0 and my.max() <= H  # alpha=0: all sources inside


def _render(Kmat, dist):
    """Analytic scene: depth of a slanted plane + colour stripes, through a (possibly distorted) camera."""
    uu, vv = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    if dist is None:
        x, y = (uu - Kmat[0, 2]) / Kmat[0, 0], (vv - Kmat[1, 2]) / Kmat[1, 1]
    else:
        x, y = prep.undistort_points_normalized(uu, vv, Kmat, dist, iters=30)
    depth = 2.0 / (1.0 + 0.3 * x - 0.2 * y)  # plane n.p = 2 with n = (0.3, -0.2, 1)
    rgb = np.stack([127 + 100 * np.sin(6 * x), 127 + 100 * np.cos(5 * y), 127 + 100 * np.sin(4 * (x + y))], -1)
    return depth.astype(np.float32), np.clip(np.rint(rgb), 0, 255).astype(np.uint8)


def test_oracle_remap_restatement_basics():
    """oracle/host_prep.py's numpy remaps (the checker of the device rectification, tests/test_gpu_tum.py): identity maps return the
    image, an integer shift moves it and fills the border with 0, half-pixel maps average neighbours with OpenCV's 1/32-pixel fixed
    point, nearest rounds half to even (cvRound)."""
    from oracle import host_prep as hp

    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (12, 16, 3)).astype(np.uint8)
    dep = rng.integers(0, 65536, (12, 16)).astype(np.uint16)
    xx, yy = np.meshgrid(np.arange(16, dtype=np.float32), np.arange(12, dtype=np.float32))
    np.testing.assert_array_equal(hp.remap_linear_u8(img, xx, yy), img)
    np.testing.assert_array_equal(hp.remap_nearest(dep, xx, yy), dep)
    sh = hp.remap_nearest(dep, xx + 3, yy - 2)
    np.testing.assert_array_equal(sh[2:, :13], dep[:10, 3:])
    assert (sh[:2] == 0).all() and (sh[:, 13:] == 0).all()
    half = hp.remap_linear_u8(img, xx + 0.5, yy)[:, :15].astype(np.int32)
    want = (img[:, :15].astype(np.int32) + img[:, 1:].astype(np.int32) + 1) >> 1
    assert np.abs(half - want).max() <= 1
    np.testing.assert_array_equal(hp.remap_nearest(dep, np.full_like(xx, 2.5), yy)[:, 0], dep[:, 2])  # 2.5 -> 2 (half to even)
    np.testing.assert_array_equal(hp.remap_nearest(dep, np.full_like(xx, 3.5), yy)[:, 0], dep[:, 4])  # 3.5 -> 4


def test_distorted_synthetic_stream_is_undone_by_the_maps():
    """The synthetic stream rendered through TUM1's lens model (SyntheticRGBD(distorted=True)), rectified with the maps of
    pyslam_amd/prep.py and the oracle's nearest remap, is the pinhole rendering at the NEW camera matrix up to resampling: the
    depth images agree within a few millimetres away from depth edges."""
    from oracle import host_prep as hp
    from pyslam_amd import prep
    from pyslam_amd.synthetic import SyntheticRGBD, render, trajectory_pose

    s = SyntheticRGBD("tum1_640x480_5mm", distorted=True, noise=False, invalid_frac=0.0)
    depth, _rgb, _T = s[5]
    K = np.array([[s.fx, 0.0, s.cx], [0.0, s.fy, s.cy], [0.0, 0.0, 1.0]])
    new_K = prep.get_optimal_new_camera_matrix(K, s.dist, (s.width, s.height), 0.7, (s.width, s.height))[0]
    mx, my = prep.init_undistort_rectify_map(K, s.dist, new_K, (s.width, s.height))
    rect = hp.remap_nearest(depth, mx, my)
    ideal = render(trajectory_pose(5)[1], s.width, s.height, new_K[0, 0], new_K[1, 1], new_K[0, 2], new_K[1, 2])[0]
    ok = rect > 0
    assert ok.mean() > 0.9
    err = np.abs(rect - ideal)[ok]
    assert np.median(err) < 2e-3 and (err < 0.02).mean() > 0.97


@pytest.mark.gpu
def test_gpu_remap_undistorts_analytic_scene():
    from pyslam_amd.volumetric import VoxelBlockGrid

    vol = VoxelBlockGrid(0.05, 8, max_blocks=1 << 10, max_points=1 << 20)
    und = prep.Undistorter(vol, K, D, W, H, use_optimal_new_K=True, alpha=0.7)
    depth_d, rgb_d = _render(K, D)                 # what the distorted sensor sees
    depth_i, rgb_i = _render(und.new_K, None)      # ideal pinhole view with the new camera matrix
    depth_u, rgb_u = und.depth(depth_d), und.color(rgb_d)
    inside = (und.map_x > 1) & (und.map_x < W - 2) & (und.map_y > 1) & (und.map_y < H - 2)
    assert inside.mean() > 0.6
    assert np.abs(depth_u - depth_i)[inside].max() < 5e-3       # nearest sampling of a smooth depth field
    assert np.abs(rgb_u.astype(int) - rgb_i.astype(int))[inside].max() <= 6  # bilinear colour, 8-bit rounding
    assert (depth_u[~((und.map_x >= -0.5) & (und.map_x < W - 0.5) & (und.map_y >= -0.5) & (und.map_y < H - 0.5))] == 0).all()
    lab = (np.arange(H * W, dtype=np.int32).reshape(H, W) % 41)
    lab_u = und.labels(lab)
    sx, sy = np.rint(und.map_x).astype(int), np.rint(und.map_y).astype(int)
    ok = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H)
    np.testing.assert_array_equal(lab_u[ok], lab[sy[ok], sx[ok]])  # exact nearest-neighbour lookup


CV2_FIXTURE = __import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "remap_cv2.npz")


@pytest.mark.skipif(not __import__("os").path.exists(CV2_FIXTURE), reason="tests/golden/remap_cv2.npz not generated yet (needs cv2: tools/make_golden_remap.py)")
def test_maps_match_opencv_fixture():
    """PIN of rows P2 / N1 - active once tools/make_golden_remap.py has run where OpenCV exists: the restated
    getOptimalNewCameraMatrix / initUndistortRectifyMap against cv2's (new K to 1e-6, maps to 1e-3 px)."""
    z = np.load(CV2_FIXTURE)
    W_, H_ = (int(x) for x in z["size"])
    new_K, _ = prep.get_optimal_new_camera_matrix(z["K"], z["D"], (W_, H_), 0.0, (W_, H_))
    np.testing.assert_allclose(new_K, z["new_K"], atol=1e-6)
    mx, my = prep.init_undistort_rectify_map(z["K"], z["D"], z["new_K"], (W_, H_))
    assert np.abs(mx - z["map_x"]).max() < 1e-3 and np.abs(my - z["map_y"]).max() < 1e-3


@pytest.mark.gpu
@pytest.mark.skipif(not __import__("os").path.exists(CV2_FIXTURE), reason="tests/golden/remap_cv2.npz not generated yet (needs cv2: tools/make_golden_remap.py)")
def test_gpu_remap_matches_opencv_fixture():
    """hv_remap against cv2.remap on OpenCV's own maps: nearest-neighbour results identical, bilinear colour within 1/255
    (cv2 interpolates with 5-bit fixed-point weights)."""
    from pyslam_amd.volumetric import VoxelBlockGrid

    z = np.load(CV2_FIXTURE)
    vol = VoxelBlockGrid(0.05, 8, max_blocks=1 << 10, max_points=1 << 16)
    np.testing.assert_array_equal(vol.remap(z["depth"], z["map_x"], z["map_y"], linear=False), z["depth_nearest"])
    np.testing.assert_array_equal(vol.remap(z["labels"], z["map_x"], z["map_y"], linear=False), z["labels_nearest"])
    got = vol.remap(z["img"], z["map_x"], z["map_y"], linear=True).astype(np.int32)
    assert np.abs(got - z["img_linear"].astype(np.int32)).max() <= 1
