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


def test_long_batch_replay_into_small_pool_loses_nothing():
    """VERDICT r01 #8 / ADVICE: a replay made of batch calls only (no data-returning call in between, as rebuild() does) into
    a pool that is far too small: the pool grows inside the integrate calls - in time, or after a verified claim pass - and
    the result is bit-identical to a volume that was created large.  No call returns an error, no unit is dropped."""
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, ScalableTSDFVolume

    s = SyntheticRGBD("tiny_160x120_2cm")
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    small = ScalableTSDFVolume(0.02, 0.08, max_blocks=64)
    big = ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 14)
    for lo in range(0, 120, 8):  # 15 batch calls back to back, the camera moves on: new units in every call
        d, c, Ts = s.batch(lo * 4, 8)
        for v in (small, big):
            v.integrate_batch(d, c, K, Ts, depth_scale=1.0, depth_trunc=4.0)
    assert small.dropped_points() == 0
    assert small.max_blocks() >= small.num_blocks() > 64
    for a, b in zip(small.dump(), big.dump()):
        np.testing.assert_array_equal(a, b)


def test_pool_that_cannot_grow_fails_before_fusing(monkeypatch):
    """HV_AUTO_GROW=0: the call whose units do not fit returns the capacity error BEFORE its frame is fused, the volume keeps
    exactly what it had, later calls keep failing (nothing is silently dropped) until reserve_blocks() makes room."""
    from pyslam_amd._lib import HipVolError
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume

    monkeypatch.setenv("HV_AUTO_GROW", "0")
    s = SyntheticRGBD("tiny_160x120_2cm")
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    frames = [s[i] for i in (0, 40, 80, 120)]
    probe = ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 13)
    probe.integrate(RGBDImage.create_from_color_and_depth(frames[0][1], frames[0][0], 1.0, 4.0, False), K, frames[0][2])
    first = probe.num_blocks()
    vol = ScalableTSDFVolume(0.02, 0.08, max_blocks=first + 8)  # room for the first view, not for a second one
    vol.integrate(RGBDImage.create_from_color_and_depth(frames[0][1], frames[0][0], 1.0, 4.0, False), K, frames[0][2])
    before = vol.dump()
    with pytest.raises(HipVolError, match="NOT fused"):
        vol.integrate(RGBDImage.create_from_color_and_depth(frames[1][1], frames[1][0], 1.0, 4.0, False), K, frames[1][2])
    with pytest.raises(HipVolError, match="pool exhausted"):  # sticky: the batch path refuses as well
        d, c, Ts = np.stack([f[0] for f in frames[2:]]), np.stack([f[1] for f in frames[2:]]), np.stack([f[2] for f in frames[2:]])
        vol.integrate_batch(d, c, K, Ts, depth_scale=1.0, depth_trunc=4.0)
    vol.reserve_blocks(1 << 13)  # explicit growth is still possible: the table is rebuilt without the failed claims
    for a, b in zip(vol.dump(), before):
        np.testing.assert_array_equal(a, b)  # nothing of the refused frames went in, nothing of the first one was lost
    for f in frames[1:]:
        vol.integrate(RGBDImage.create_from_color_and_depth(f[1], f[0], 1.0, 4.0, False), K, f[2])
        probe.integrate(RGBDImage.create_from_color_and_depth(f[1], f[0], 1.0, 4.0, False), K, f[2])
    for a, b in zip(vol.dump(), probe.dump()):
        np.testing.assert_array_equal(a, b)


def test_failed_claim_rollback_keeps_imported_units(monkeypatch):
    """units that arrive through import_numerators (a gather onto the root) are not announced by a publishing
    kernel; a later integrate call whose claims do not fit - and whose pool cannot grow - must roll back to a state that
    still contains them.  The rollback works in place (no second pool)."""
    from pyslam_amd._lib import HipVolError
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume

    monkeypatch.setenv("HV_AUTO_GROW", "0")
    s = SyntheticRGBD("tiny_160x120_2cm")
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    src = ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 13)
    d0, c0, T0 = s[0]
    src.integrate(RGBDImage.create_from_color_and_depth(c0, d0, 1.0, 4.0, False), K, T0)
    keys = src.unit_keys()
    payload = src.export_numerators(keys)
    vol = ScalableTSDFVolume(0.02, 0.08, max_blocks=len(keys) + 8)
    vol.import_numerators(keys, payload)
    before = vol.dump()
    assert len(before[0]) == len(keys)
    d1, c1, T1 = s[80]  # another view: far more new units than the 8 that are free
    with pytest.raises(HipVolError, match="NOT fused"):
        vol.integrate(RGBDImage.create_from_color_and_depth(c1, d1, 1.0, 4.0, False), K, T1)
    for a, b in zip(vol.dump(), before):
        np.testing.assert_array_equal(a, b)  # every imported unit is still there, bit for bit
    vol.reserve_blocks(1 << 13)
    vol.integrate(RGBDImage.create_from_color_and_depth(c1, d1, 1.0, 4.0, False), K, T1)
    assert vol.num_blocks() > len(keys)
