"""GPU: the block pool grows by itself (hv_reserve_blocks / auto-grow at data-returning calls) and growth never
changes the fused result: a volume that starts tiny ends bit-identical to one created large."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_tsdf_pool_grows_and_matches_large_pool():
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume

    s = SyntheticRGBD("tiny_160x120_2cm")
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    small = ScalableTSDFVolume(0.02, 0.08, max_blocks=512)
    big = ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 14)
    grew = 0
    for i in range(0, 40, 2):
        depth, rgb, T = s[i]
        img = RGBDImage.create_from_color_and_depth(rgb, depth, 1.0, 4.0, False)
        for v in (small, big):
            v.integrate(img, K, T)
        before = small.max_blocks()
        small.num_blocks()  # a data-returning call: the growth point
        grew += small.max_blocks() > before
    assert grew >= 1 and small.max_blocks() > 512
    for a, b in zip(small.dump(), big.dump()):
        np.testing.assert_array_equal(a, b)
    # batch path after growth, explicit reserve
    small.reserve_blocks(small.max_blocks() * 2)
    d, c, Ts = s.batch(40, 4)
    for v in (small, big):
        v.integrate_batch(d, c, K, Ts, depth_scale=1.0, depth_trunc=4.0)
    for a, b in zip(small.dump(), big.dump()):
        np.testing.assert_array_equal(a, b)
    ma, mb = small.extract_triangle_mesh(), big.extract_triangle_mesh()
    assert ma.triangles.shape == mb.triangles.shape and ma.vertices.shape == mb.vertices.shape


@pytest.mark.parametrize("mode", ["grid", "vote", "prob"])
def test_grid_pools_grow(mode):
    from oracle import host_prep as hp
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import VoxelBlockGrid
    from pyslam_amd.volumetric_semantic import VoxelBlockSemanticGrid, VoxelBlockSemanticProbabilisticGrid

    cls = {"grid": VoxelBlockGrid, "vote": VoxelBlockSemanticGrid, "prob": VoxelBlockSemanticProbabilisticGrid}[mode]
    s = SyntheticRGBD("tiny_160x120_2cm")
    small, big = cls(0.02, 8, max_blocks=1024, max_points=1 << 16), cls(0.02, 8, max_blocks=1 << 15, max_points=1 << 16)
    for i in range(0, 30, 3):
        depth, rgb, T = s[i]
        pts, cols, valid = hp.frame_to_world_f32(depth, rgb, *s.intrinsics, T, 4.0)
        lab = s.labels(i)[valid].astype(np.int32)
        for v in (small, big):
            if mode == "grid":
                v.integrate(pts, cols)
            else:
                v.integrate(pts, cols, lab, lab % 3)
        small.num_blocks()
    assert small.max_blocks() > 1024
    for a, b in zip(small.dump(), big.dump()):
        np.testing.assert_array_equal(a, b)
