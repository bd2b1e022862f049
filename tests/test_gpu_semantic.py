"""GPU parity: VOXEL_SEMANTIC_GRID (voting payload) vs the compiled reference / its C restatement:
labels, confidence counters, counts, float64 position sums and float colour sums bit-exact."""
import numpy as np
import pytest

import oracle
from oracle.semantic import PortSemGrid, RefSemGrid
from tests.test_semantic_oracle import srt, stream

pytestmark = pytest.mark.gpu


def make_oracle(voxel, bs=8):
    return RefSemGrid(voxel, bs) if oracle.ref_available() else PortSemGrid(voxel, bs)


def test_reference_kats_on_gpu():
    from pyslam_amd.volumetric_semantic import VoxelBlockSemanticGrid

    g = VoxelBlockSemanticGrid(0.1, 8, max_blocks=1 << 10, max_points=1 << 16)
    g.integrate(np.zeros((2, 3), np.float64), np.zeros((2, 3), np.uint8), np.array([1, 2], np.int32), np.array([1, 2], np.int32))
    v = g.get_voxels(min_count=1, min_confidence=0.0)
    assert len(v.object_ids) == 1 and v.object_ids[0] == 2 and v.class_ids[0] == 2
    assert v.confidences[0] == pytest.approx(0.5, abs=1e-3)  # cpp/test_volumetric_voxel_semantic.py:20-36
    g.clear()
    g.integrate(np.array([[0.0, 0.0, 0.0], [0.2, 0.0, 0.0]]), np.zeros((2, 3), np.uint8), np.array([10, 20], np.int32),
                np.array([101, 202], np.int32))
    v = g.get_voxels(1, 0.0)
    paired = sorted(zip(map(tuple, v.points), v.object_ids, v.class_ids))
    assert [p[1:] for p in paired] == [(101, 10), (202, 20)]  # :79-97


@pytest.mark.parametrize("pos_dtype", [np.float32, np.float64])
@pytest.mark.parametrize("use_inst,use_depth", [(True, True), (True, False), (False, True), (False, False)])
def test_stream_bit_exact(pos_dtype, use_inst, use_depth):
    from pyslam_amd.volumetric_semantic import VoxelBlockSemanticGrid

    gpu = VoxelBlockSemanticGrid(0.05, 8, max_blocks=1 << 14, max_points=1 << 17)
    cpu = make_oracle(0.05)
    for it in range(3):
        pts, cols, cls, inst, dep = stream(300 + it, 50000, pos_dtype)
        c = cols if it != 1 else (cols / 255.0).astype(np.float32)
        for g in (gpu, cpu):
            g.integrate(pts, c, cls, inst if use_inst else None, dep if use_depth else None)
    assert gpu.dropped_points() == 0
    for a, b in zip(gpu.dump(), cpu.dump()):
        np.testing.assert_array_equal(a, b)
    for mc, mconf in ((1, 0.0), (2, 0.5), (3, 0.6)):
        v = gpu.get_voxels(mc, mconf)
        got = srt((v.points, v.colors, v.class_ids, v.object_ids, v.confidences))
        for a, b in zip(got, srt(cpu.get_voxels(mc, mconf))):
            np.testing.assert_array_equal(a, b)


def test_depth_threshold_and_errors():
    from pyslam_amd.volumetric_semantic import VoxelBlockSemanticGrid

    gpu = VoxelBlockSemanticGrid(0.05, 8, max_blocks=1 << 12, max_points=1 << 16)
    cpu = PortSemGrid(0.05, 8)
    gpu.set_depth_threshold(5.0)
    cpu.set_depth_threshold(5.0)
    try:
        pts, cols, cls, inst, dep = stream(7, 20000)
        gpu.integrate(pts, cols, cls, inst, dep)
        cpu.integrate(pts, cols, cls, inst, dep)
        for a, b in zip(gpu.dump(), cpu.dump()):
            np.testing.assert_array_equal(a, b)
    finally:
        cpu.set_depth_threshold(10.0)
    with pytest.raises(RuntimeError, match="instance_ids but no class_ids"):
        gpu.integrate(pts, cols, None, inst)
    with pytest.raises(RuntimeError, match="same size"):
        gpu.integrate(pts, cols, cls[:-1])


@pytest.mark.parametrize("voxel", [0.1, 0.25, 0.5])
@pytest.mark.parametrize("pos_dtype", [np.float32, np.float64])
def test_big_buckets_bit_exact(voxel, pos_dtype):
    """The per-keyframe bucket path with buckets far beyond a wave's LDS window: 60 000 points in a 2 m cube with 0.8 / 2 / 4 m
    blocks = 2 000 ... 60 000 points per block (voxel-index ranges; at 0.5 m a range still overflows -> point-index windows)."""
    from pyslam_amd.volumetric_semantic import VoxelBlockSemanticGrid

    gpu = VoxelBlockSemanticGrid(voxel, 8, max_blocks=1 << 10, max_points=1 << 17)
    cpu = make_oracle(voxel)
    for it in range(2):
        pts, cols, cls, inst, dep = stream(900 + it, 60000, pos_dtype)
        for g in (gpu, cpu):
            g.integrate(pts, cols, cls, inst, dep)
    assert gpu.dropped_points() == 0
    for a, b in zip(gpu.dump(), cpu.dump()):
        np.testing.assert_array_equal(a, b)
