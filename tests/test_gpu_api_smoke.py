"""GPU: API-surface smoke test in the spirit of the reference's cpp/test_volumetric.py (:421-576,
"every binding is callable, nothing throws") for the classes this package mirrors, plus shape/dtype
checks the reference script does not make."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_voxel_block_grid_surface():
    from pyslam_amd.volumetric import BoundingBox3D, CameraFrustrum, TBBUtils, VoxelBlockGrid

    TBBUtils.set_max_threads(2)
    rng = np.random.default_rng(0)
    points = rng.random((1000, 3)) * 2.0  # float64 like the reference script (:462) -> narrowed to float32
    colors = rng.integers(0, 255, (1000, 3)).astype(np.uint8)  # (:463)
    g = VoxelBlockGrid(voxel_size=0.1, block_size=8)
    assert g.get_block_size() == 8 and g.empty() and g.num_blocks() == 0
    g.integrate(points, colors)
    g.integrate(points.astype(np.float32))
    assert not g.empty() and g.num_blocks() > 0 and g.size() == g.get_total_voxel_count() > 0
    v = g.get_voxels()
    assert v.points.shape == v.colors.shape == (g.size(), 3) and v.points.dtype == np.float32
    assert g.get_points().shape == g.get_colors().shape == v.points.shape
    bb = BoundingBox3D([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
    inside = g.get_voxels_in_bb(bb, min_count=1)
    assert 0 < len(inside.points) < len(v.points)
    assert (inside.points >= 0).all() and (inside.points <= 1.0).all()
    fr = CameraFrustrum(500.0, 500.0, 320.0, 240.0, 640, 480, np.eye(4), depth_max=10.0, depth_min=0.01)
    fr.set_T_cw(np.eye(4))
    seen = g.get_voxels_in_camera_frustrum(fr)
    assert 0 < len(seen.points) <= len(v.points)
    depth = (rng.random((480, 640)) * 5 + 0.5).astype(np.float32)  # (:574-576)
    g.carve(fr, depth, 1e-2)
    g.carve(fr, depth[:10], 1e-2)  # wrong size: reference prints and returns
    g.remove_low_count_voxels(2)
    g.remove_low_confidence_voxels(0.5)
    g.clear()
    assert g.empty()
    g.integrate(points, colors)
    g.reset()
    assert g.size() == 0


def test_tsdf_volume_surface():
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume

    s = SyntheticRGBD("tiny_160x120_2cm")
    vol = ScalableTSDFVolume(voxel_length=4.0 / 512.0 * 4, sdf_trunc=0.12)  # test/open3d/test_volume_integration.py:105-109 scale
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    for i in range(0, 30, 10):  # every 10th frame (:275-284)
        depth, rgb, T = s[i]
        vol.integrate(RGBDImage.create_from_color_and_depth(rgb, depth, depth_scale=1.0, depth_trunc=4.0,
                                                            convert_rgb_to_intensity=False), K, T)
    mesh = vol.extract_triangle_mesh()
    assert mesh.vertices.dtype == np.float64 and mesh.triangles.dtype == np.int32 and mesh.vertex_colors.shape == mesh.vertices.shape
    pc = vol.extract_point_cloud()
    assert pc.points.shape == pc.colors.shape and len(pc.points) > 0
    vol.reset()
    assert vol.num_blocks() == 0 and vol.extract_triangle_mesh().triangles.shape == (0, 3)


def test_semantic_grid_surface():
    from pyslam_amd.volumetric_semantic import VoxelBlockSemanticGrid

    rng = np.random.default_rng(1)
    g = VoxelBlockSemanticGrid(0.1, 8)
    pts = rng.random((500, 3))
    g.integrate(pts, rng.integers(0, 255, (500, 3)).astype(np.uint8), rng.integers(0, 3, 500).astype(np.int32),
                rng.integers(0, 3, 500).astype(np.int32), (rng.random(500) * 3).astype(np.float32))
    v = g.get_voxels(1, 0.0)
    assert v.points.dtype == np.float64 and len(v.class_ids) == len(v.object_ids) == len(v.confidences) == len(v.points) > 0
    assert ((v.confidences >= 0) & (v.confidences <= 1)).all()
    g.reset()
    assert g.empty()
