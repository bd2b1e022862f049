"""CPU, dev container only: the C restatement against the live compiled reference (oracle/_ref,
built from /root/reference by oracle/Makefile) on larger seeded inputs than the golden fixtures."""
import numpy as np
import pytest

import oracle
from oracle import host_prep as hp

pytestmark = pytest.mark.skipif(not oracle.ref_available(), reason="compiled reference not available")


def test_keys_random_and_adversarial():
    from tools.make_golden import golden_points

    for voxel in (0.005, 0.004, 0.002, 0.015, 0.05):
        pts, _ = golden_points(3, 100_000, voxel)
        for a, b in zip(oracle.keys(pts, voxel, 8, "port"), oracle.keys(pts, voxel, 8, "ref")):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("bs", [4, 8, 16])
def test_integrate_stream_bit_exact(bs):
    from pyslam_amd.synthetic import SyntheticRGBD

    s = SyntheticRGBD("tiny_160x120_2cm")
    port, ref = oracle.PortGrid(0.02, bs), oracle.RefGrid(0.02, bs)
    for i in range(3):
        depth, rgb, T = s[i]
        p, c, _ = hp.frame_to_world_f32(depth, rgb, *s.intrinsics, T, 4.0)
        for g in (port, ref):
            g.integrate(p, c if i != 1 else (c * 255).astype(np.uint8))
    for a, b in zip(port.dump(), ref.dump()):
        np.testing.assert_array_equal(a, b)
    assert port.size() == ref.size() and port.num_blocks() == ref.num_blocks()
    port.remove_low_count_voxels(2)
    ref.remove_low_count_voxels(2)
    for a, b in zip(port.dump(), ref.dump()):
        np.testing.assert_array_equal(a, b)
    port.clear()
    ref.clear()
    assert port.empty() and ref.empty()


def test_frustum_contains():
    from pyslam_amd.synthetic import SyntheticRGBD

    s = SyntheticRGBD("tiny_160x120_2cm")
    intr = np.array(s.intrinsics, np.float32)
    rng = np.random.default_rng(0)
    for f in range(3):
        T = s.pose(f * 17)
        for p in (rng.random((300, 3)) * np.array([6, 4, 3])).astype(np.float32):
            a = oracle.frustum_contains(intr, s.width, s.height, T, 8.0, 0.01, p, "port")
            b = oracle.frustum_contains(intr, s.width, s.height, T, 8.0, 0.01, p, "ref")
            assert a[0] == b[0]
            np.testing.assert_array_equal(a[1], b[1])
        np.testing.assert_array_equal(oracle.frustum_bbox(intr, s.width, s.height, T, 8.0, 0.01, "port"),
                                      oracle.frustum_bbox(intr, s.width, s.height, T, 8.0, 0.01, "ref"))


@pytest.mark.skipif(not oracle.ref_available(), reason="compiled reference not available")
def test_reference_double_overload_keys_in_double():
    """Pins what `integrate(points float64)` means in the reference (volumetric_grid_module.h:738-741 ->
    integrate_raw<double, ...>): the voxel index is floor(x * (double)inv_voxel_size_f32) in DOUBLE, the voxel sums take
    static_cast<float>(x).  Checked on points whose float32 narrowing lands in the neighbouring cell; the GPU path is held to
    the same compiled reference in tests/test_gpu_voxel_grid.py::test_float64_points_take_the_double_overload."""
    rng = np.random.default_rng(5)
    vs = float(np.float32(0.005))
    inv = float(np.float32(1.0) / np.float32(0.005))
    k = rng.integers(-4000, 4000, size=(20000, 3)).astype(np.float64)
    pts = np.concatenate([k * vs * (1.0 - 1e-12), k * vs * (1.0 + 1e-12), (rng.random((20000, 3)) - 0.5) * 40.0])
    g = oracle.RefGrid(0.005, 8)
    g.integrate(pts)
    keys, _, counts, sums = g.dump()
    # restatement: voxel index in double, block = floor_div(v, 8), local = v - 8 block, linear x + 8 y + 64 z
    v = np.floor(pts * inv).astype(np.int64)
    b = v >> 3
    lin = (v - 8 * b) @ np.array([1, 8, 64])
    order = {tuple(key): i for i, key in enumerate(keys)}
    want = np.zeros_like(counts)
    pos = np.zeros(counts.shape + (3,), np.float32)
    for row, (bk, l) in enumerate(zip(map(tuple, b), lin)):  # sequential float32 accumulation, point order
        i = order[bk]
        want[i, l] += 1
        pos[i, l] += pts[row].astype(np.float32)
    np.testing.assert_array_equal(counts, want)
    np.testing.assert_array_equal(sums[..., :3].view(np.uint32), pos.view(np.uint32))
    g32 = oracle.RefGrid(0.005, 8)
    g32.integrate(pts.astype(np.float32))
    assert g32.dump()[0].shape != keys.shape or not np.array_equal(g32.dump()[2], counts)  # and NOT the float overload's result
