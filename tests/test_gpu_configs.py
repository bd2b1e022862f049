"""GPU parity at the other BASELINE.json configurations (they are parity-test cases, not bench lines):
Replica-shaped 1200x680 @ 4 mm, EuRoC-shaped 752x480 @ 10 mm, ScanNet-shaped 1296x968 @ 2 mm — voxel
sizes whose float32 reciprocals are inexact — one frame each against the oracle,
plus size-independent properties over a short stream."""
import numpy as np
import pytest

import oracle
from oracle import host_prep as hp
from tests.conftest import FOLD_TSDF_TOL, sweep_is_bitwise, synthetic_frames

pytestmark = pytest.mark.gpu

CASES = [("replica_1200x680_4mm", 0.004, 0.04, 1 << 15), ("euroc_752x480_10mm", 0.010, 0.04, 1 << 13),
         ("scannet_1296x968_2mm", 0.002, 0.016, 1 << 16)]


@pytest.mark.parametrize("config,voxel,trunc,max_blocks", CASES)
def test_tsdf_frame_matches_oracle(config, voxel, trunc, max_blocks):
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume

    s, frames = synthetic_frames(config, 3, 1)
    depth, rgb, T = frames[0]
    gpu = ScalableTSDFVolume(voxel, trunc, max_blocks=max_blocks, max_points=s.width * s.height)
    cpu = oracle.PortTsdf(voxel, trunc, threads=8)
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    gpu.integrate(RGBDImage.create_from_color_and_depth(rgb, depth, 1.0, 4.0, False), K, T)
    cpu.integrate(depth, rgb, K.as_array(), T, 1.0, 4.0)
    np.testing.assert_array_equal(gpu.touched_keys(), cpu.touched_keys())
    assert gpu.num_blocks() == cpu.num_units()
    perm = np.arange(4096).reshape(16, 16, 16).transpose(2, 0, 1).reshape(-1)  # internal index z*256+x*16+y -> oracle index x*256+y*16+z

    def compare_sample(swept=False):
        """A deterministic sample of ~200 units in full (a full dump at 2 mm is several GB on the host): weights exact,
        tsdf bitwise (as the f32 numerator tsdf*w both sides form identically), colour <= 1e-4 on [0, 1]."""
        ka = gpu.unit_keys()
        assert len(ka) == cpu.num_units()
        kb, tb, wb, cb = cpu.dump()
        sel = np.arange(0, len(kb), max(1, len(kb) // 200))
        payload = gpu.export_numerators(kb[sel])
        wb_i, tb_i, cb_i = wb[sel][:, perm], tb[sel][:, perm], cb[sel][:, perm]
        np.testing.assert_array_equal(payload[..., 1], wb_i)
        if swept and not sweep_is_bitwise():  # the production sweep folds a batch per voxel: tsdf to FOLD_TSDF_TOL
            assert np.abs(payload[..., 0] / np.maximum(wb_i, 1.0) - tb_i).max() <= FOLD_TSDF_TOL
        else:
            np.testing.assert_array_equal(payload[..., 0], tb_i * wb_i)
        mean = payload[..., 2:5].astype(np.float64) / np.maximum(wb_i, 1.0)[..., None]
        assert np.abs(mean - cb_i).max() / 255.0 <= 1e-4
        return int(wb_i.max())

    assert compare_sample() == 1
    # the multi-frame sweep at this configuration: 8 further frames in one call, revisiting the first frame's units
    s2, more = synthetic_frames(config, 4, 8)
    gpu.integrate_batch(np.stack([f[0] for f in more]), np.stack([f[1] for f in more]), K, np.stack([f[2] for f in more]),
                        depth_scale=1.0, depth_trunc=4.0)
    for d, c, Tcw in more:
        cpu.integrate(d, c, K.as_array(), Tcw, 1.0, 4.0)
    assert gpu.num_blocks() == cpu.num_units()
    assert compare_sample(swept=True) >= 8
    assert gpu.dropped_points() == 0


@pytest.mark.parametrize("config,voxel", [("replica_1200x680_4mm", 0.004), ("scannet_1296x968_2mm", 0.002)])
def test_voxel_grid_frame_matches_reference(config, voxel):
    from pyslam_amd.volumetric import VoxelBlockGrid

    s, frames = synthetic_frames(config, 5, 1)
    depth, rgb, T = frames[0]
    gpu = VoxelBlockGrid(voxel, 8, max_blocks=1 << 18, max_points=s.width * s.height)
    cpu = oracle.PortGrid(voxel, 8)  # pinned to the compiled reference by the CPU suite
    pts, cols, _ = hp.frame_to_world_f32(depth, rgb, *s.intrinsics, T, 4.0)
    gpu.integrate_rgbd(depth, rgb, *s.intrinsics, T, max_depth=4.0)
    cpu.integrate(pts, cols)
    for a, b in zip(gpu.dump(), cpu.dump()):
        np.testing.assert_array_equal(a, b)


def test_replica_stream_mesh_every_10_frames():
    """BASELINE configs[2]: 4 mm TSDF + colour, marching cubes every 10 frames: extraction interleaved
    with fusion stays consistent (indices valid, colours in range)."""
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume

    s, frames = synthetic_frames("replica_1200x680_4mm", 0, 21)
    vol = ScalableTSDFVolume(0.004, 0.04, max_blocks=1 << 16, max_points=s.width * s.height)
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    sizes = []
    for i, (depth, rgb, T) in enumerate(frames):
        vol.integrate(RGBDImage.create_from_color_and_depth(rgb, depth, 1.0, 4.0, False), K, T)
        if i % 10 == 0:
            m = vol.extract_triangle_mesh()
            assert m.triangles.max() < len(m.vertices) and m.triangles.min() >= 0
            assert (m.vertex_colors >= 0).all() and (m.vertex_colors <= 1).all()
            sizes.append(len(m.vertices))
    # (the first, single-observation mesh is the largest: sensor noise of up to 4 voxels is not averaged yet)
    assert len(sizes) == 3 and min(sizes) > 100000
