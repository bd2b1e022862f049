"""GPU parity: VOXEL_GRID mode vs the compiled reference (oracle/_ref) / its C restatement.

Bar: block keys, BlockKeyHash, per-voxel count AND the float32 position/colour sums are bit-exact
(the GPU folds every voxel's points in point-index order, the reference's sequential order)."""
import numpy as np
import pytest

import oracle
from oracle import host_prep as hp
from tests.conftest import sort_rows, synthetic_frames

pytestmark = pytest.mark.gpu


def make_oracle(voxel, bs=8):
    return oracle.RefGrid(voxel, bs) if oracle.ref_available() else oracle.PortGrid(voxel, bs)


def assert_same_grid(gpu, cpu):
    a, b = gpu.dump(), cpu.dump()
    assert a[0].shape == b[0].shape, (a[0].shape, b[0].shape)
    np.testing.assert_array_equal(a[0], b[0])  # block keys
    np.testing.assert_array_equal(a[1], b[1])  # BlockKeyHash
    np.testing.assert_array_equal(a[2], b[2])  # counts
    np.testing.assert_array_equal(a[3].view(np.uint32), b[3].view(np.uint32))  # sums, bitwise


def adversarial_points(rng, voxel, n):
    """negative coords, exact voxel multiples, +/-0, denormals, block boundaries."""
    vs = np.float32(voxel)
    k = rng.integers(-4000, 4000, size=(n, 3)).astype(np.float32)
    exact = k * vs
    nudged = np.nextafter(exact, np.float32(np.inf) * rng.choice([-1, 1], size=(n, 3)).astype(np.float32))
    special = np.array(
        [[0.0, -0.0, 1e-45], [-1e-45, 1e-38, -1e-38], [-0.001, 0.5, 1.25], [8 * voxel, -8 * voxel, 16 * voxel],
         [-vs, vs, -2 * vs]], dtype=np.float32)
    rand = (rng.random((n, 3), dtype=np.float32) - np.float32(0.5)) * np.float32(40.0)
    return np.concatenate([exact, nudged, special, rand]).astype(np.float32)


@pytest.mark.parametrize("voxel", [0.005, 0.004, 0.002, 0.015, 0.05])
def test_keys_bit_exact(voxel):
    from pyslam_amd.volumetric import VoxelBlockGrid

    rng = np.random.default_rng(1)
    pts = adversarial_points(rng, voxel, 200_000)
    g = VoxelBlockGrid(voxel, 8, max_blocks=1 << 12, max_points=1 << 20)
    got = g.keys_from_points(pts)
    want = oracle.keys(pts, voxel, 8, "ref" if oracle.ref_available() else "port")
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("color_kind", ["f32", "u8", "none"])
def test_integrate_points_bit_exact(color_kind):
    from pyslam_amd.volumetric import VoxelBlockGrid

    s, frames = synthetic_frames("synthetic_640x480_5mm", 3, 3)
    gpu = VoxelBlockGrid(0.005, 8, max_blocks=1 << 17, max_points=1 << 20)
    cpu = make_oracle(0.005)
    for depth, rgb, T in frames:
        pts, cols, _ = hp.frame_to_world_f32(depth, rgb, *s.intrinsics, T, 4.0)
        c = {"f32": cols, "u8": (cols * 255).astype(np.uint8), "none": None}[color_kind]
        gpu.integrate(pts, c)
        cpu.integrate(pts, c)
    assert gpu.dropped_points() == 0
    assert gpu.num_blocks() == cpu.num_blocks()
    assert gpu.size() == cpu.size()
    assert_same_grid(gpu, cpu)
    # get_voxels as key-sorted sets (H8: the reference's row order is its hash-map order)
    for min_count in (1, 3):
        v = gpu.get_voxels(min_count)
        pa, ca = sort_rows(v.points, v.colors)
        pb, cb = sort_rows(*cpu.get_voxels(min_count))
        np.testing.assert_array_equal(pa, pb)
        np.testing.assert_array_equal(ca, cb)


def test_integrate_device_resident_inputs():
    import torch
    from pyslam_amd.volumetric import VoxelBlockGrid

    s, frames = synthetic_frames("tiny_160x120_2cm", 0, 2)
    gpu = VoxelBlockGrid(0.02, 8, max_blocks=1 << 14, max_points=1 << 18)
    cpu = make_oracle(0.02)
    for depth, rgb, T in frames:
        pts, cols, _ = hp.frame_to_world_f32(depth, rgb, *s.intrinsics, T, 4.0)
        gpu.integrate(torch.from_numpy(pts).cuda(), torch.from_numpy(cols).cuda())
        cpu.integrate(pts, cols)
    assert_same_grid(gpu, cpu)


def test_edge_cases_empty_ragged_duplicates():
    from pyslam_amd.volumetric import VoxelBlockGrid

    gpu = VoxelBlockGrid(0.05, 8, max_blocks=1 << 10, max_points=1 << 16)
    cpu = make_oracle(0.05)
    assert gpu.empty() and gpu.size() == 0 and gpu.num_blocks() == 0
    gpu.integrate(np.zeros((0, 3), np.float32))  # empty input is a no-op
    assert gpu.get_voxels().points.shape == (0, 3)
    one = np.array([[0.01, 0.02, 0.03]], np.float32)
    many = np.repeat(one, 1000, axis=0)  # 1000 hits of one voxel: long ordered run
    cols = np.linspace(0, 1, 3000, dtype=np.float32).reshape(1000, 3)
    for g in (gpu, cpu):
        g.integrate(one)
        g.integrate(many, cols)
    assert_same_grid(gpu, cpu)
    with pytest.raises(RuntimeError, match="Nx3"):
        gpu.integrate(np.zeros((4, 2), np.float32))
    with pytest.raises(RuntimeError, match="same size"):
        gpu.integrate(np.zeros((4, 3), np.float32), np.zeros((3, 3), np.float32))
    with pytest.raises(RuntimeError, match="uint8 or float32"):
        gpu.integrate(np.zeros((4, 3), np.float32), np.zeros((4, 3), np.float64))
    gpu.clear()
    cpu.clear()
    assert gpu.empty() and gpu.num_blocks() == 0
    gpu.integrate(many[:10], cols[:10])
    cpu.integrate(many[:10], cols[:10])
    assert_same_grid(gpu, cpu)


def test_block_size_variants():
    from pyslam_amd.volumetric import VoxelBlockGrid

    rng = np.random.default_rng(5)
    pts = ((rng.random((50_000, 3), dtype=np.float32) - np.float32(0.5)) * np.float32(3.0)).astype(np.float32)
    cols = rng.random((50_000, 3), dtype=np.float32)
    for bs in (4, 5, 8, 16):
        gpu = VoxelBlockGrid(0.03, bs, max_blocks=1 << 15, max_points=1 << 17)
        cpu = make_oracle(0.03, bs)
        gpu.integrate(pts, cols)
        cpu.integrate(pts, cols)
        assert_same_grid(gpu, cpu)


def test_fused_rgbd_matches_reference_prep():
    """hv_integrate_rgbd_points == depth2pointcloud + world transform + integrate (H1): keys/counts
    must match the oracle fed with the reference-style numpy points; sums to 1e-4 on the averages."""
    from pyslam_amd.volumetric import VoxelBlockGrid

    s, frames = synthetic_frames("synthetic_640x480_5mm", 10, 2)
    gpu = VoxelBlockGrid(0.005, 8, max_blocks=1 << 17, max_points=1 << 20)
    cpu = make_oracle(0.005)
    cpu_blas = make_oracle(0.005)
    for depth, rgb, T in frames:
        gpu.integrate_rgbd(depth, rgb, *s.intrinsics, T, max_depth=4.0)
        pts, cols, _ = hp.frame_to_world_f32(depth, rgb, *s.intrinsics, T, 4.0, blas=False)
        cpu.integrate(pts, cols)
        pts_b, cols_b, _ = hp.frame_to_world_f32(depth, rgb, *s.intrinsics, T, 4.0, blas=True)
        cpu_blas.integrate(pts_b, cols_b)
    assert_same_grid(gpu, cpu)  # same documented operation order -> bit-exact
    # against the literal numpy `R @ P.T` (BLAS order): report and bound the voxel-boundary ties
    a, b = gpu.dump(), cpu_blas.dump()
    if a[0].shape == b[0].shape and np.array_equal(a[0], b[0]):
        mism = int((a[2] != b[2]).sum())
    else:
        mism = -1
    total = int((b[2] > 0).sum())
    assert mism == -1 or mism <= max(4, total // 10000), (mism, total)


def test_queries_and_carve():
    from pyslam_amd.volumetric import BoundingBox3D, CameraFrustrum, VoxelBlockGrid

    s, frames = synthetic_frames("synthetic_640x480_5mm", 20, 2)
    gpu = VoxelBlockGrid(0.005, 8, max_blocks=1 << 17, max_points=1 << 20)
    cpu = make_oracle(0.005)
    for depth, rgb, T in frames:
        pts, cols, _ = hp.frame_to_world_f32(depth, rgb, *s.intrinsics, T, 4.0)
        gpu.integrate(pts, cols)
        cpu.integrate(pts, cols)
    depth, rgb, T = frames[1]
    intr = np.array(s.intrinsics, np.float32)
    fr = CameraFrustrum(*s.intrinsics, s.width, s.height, T, depth_max=3.0, depth_min=0.5)
    v = gpu.get_voxels_in_camera_frustrum(fr, min_count=2)
    pa, ca = sort_rows(v.points, v.colors)
    pb, cb = sort_rows(*cpu.get_voxels_in_camera_frustrum(intr, s.width, s.height, T, 3.0, 0.5, 2))
    np.testing.assert_array_equal(pa, pb)
    np.testing.assert_array_equal(ca, cb)
    bb = [2.0, 1.0, 0.2, 4.0, 3.0, 1.5]
    v = gpu.get_voxels_in_bb(BoundingBox3D(bb[:3], bb[3:]), min_count=1)
    pa, ca = sort_rows(v.points, v.colors)
    pb, cb = sort_rows(*cpu.get_voxels_in_bb(np.array(bb), 1))
    np.testing.assert_array_equal(pa, pb)
    np.testing.assert_array_equal(ca, cb)
    # carve with a depth image pushed back by 0.5 m on the left half: voxels there are "in front"
    dc = depth.copy()
    dc[:, : s.width // 2] += 0.5
    fr2 = CameraFrustrum(*s.intrinsics, s.width, s.height, T, depth_max=8.0, depth_min=0.01)
    gpu.carve(fr2, dc, 0.03)
    cpu.carve(intr, s.width, s.height, T, 8.0, 0.01, dc, 0.03)
    assert_same_grid(gpu, cpu)
    gpu.remove_low_count_voxels(2)
    cpu.remove_low_count_voxels(2)
    assert_same_grid(gpu, cpu)
    assert gpu.size() == cpu.size()


def test_full_size_properties():
    """BASELINE full size (640x480 @ 5 mm, 8 frames): size-independent properties."""
    from pyslam_amd.volumetric import VoxelBlockGrid

    s, frames = synthetic_frames("synthetic_640x480_5mm", 40, 8)
    gpu = VoxelBlockGrid(0.005, 8, max_blocks=1 << 18, max_points=1 << 20)
    total = 0
    for depth, rgb, T in frames:
        gpu.integrate_rgbd(depth, rgb, *s.intrinsics, T, max_depth=4.0)
        total += int(((depth > 0) & (depth < 4.0)).sum())
    keys, hashes, counts, sums = gpu.dump()
    assert counts.sum() == total  # every valid pixel lands in exactly one voxel
    assert len(np.unique(keys, axis=0)) == len(keys)  # block keys unique
    occ = counts > 0
    avg = sums[..., :3][occ] / counts[occ][:, None]
    # averaged position lies inside its voxel: floor(avg / voxel) reproduces the key
    bs = 8
    lin = np.nonzero(occ)
    lx, ly, lz = lin[1] % bs, (lin[1] // bs) % bs, lin[1] // (bs * bs)
    vox = keys[lin[0]] * bs + np.stack([lx, ly, lz], 1)
    inv = np.float32(1.0) / np.float32(0.005)
    got = np.floor(avg.astype(np.float64) * np.float64(inv) + 1e-6 * np.sign(avg)).astype(np.int64)
    assert (np.abs(got - vox) <= 1).all()
    col = sums[..., 3:][occ] / counts[occ][:, None]
    assert (col >= 0).all() and (col <= 1.0 + 1e-6).all()
    # idempotence of extraction + reset round trip
    n1 = gpu.size()
    assert n1 == int(occ.sum()) == gpu.get_voxels(1).points.shape[0]
    gpu.reset()
    assert gpu.size() == 0 and gpu.num_blocks() == 0


@pytest.mark.skipif(not oracle.ref_available(), reason="compiled reference not available")
def test_direct_voxel_grid_alias_matches_reference_voxelgrid():
    """V16: volumetric.VoxelGrid (the non-default direct voxel hash) — same rows as the compiled reference's
    VoxelGrid, including its dropped-uint8-colours quirk."""
    from pyslam_amd.volumetric import VoxelGrid

    rng = np.random.default_rng(5)
    pts = ((rng.random((120000, 3)) - 0.5) * 3).astype(np.float32)
    for cols in (rng.random((120000, 3)).astype(np.float32), rng.integers(0, 255, (120000, 3)).astype(np.uint8), None):
        gpu, ref = VoxelGrid(0.05, max_blocks=1 << 13, max_points=1 << 18), oracle.RefVoxelGrid(0.05)
        for g in (gpu, ref):
            g.integrate(pts, cols)
            g.integrate(pts[::2] * 0.5, cols[::2] if cols is not None else None)
        v = gpu.get_voxels(2)
        pa, ca = sort_rows(v.points, v.colors)
        pb, cb = sort_rows(*ref.get_voxels(2))
        np.testing.assert_array_equal(pa, pb)
        np.testing.assert_array_equal(ca, cb)
        assert gpu.size() == ref.size()


def test_batched_replay_is_bit_identical_to_per_frame():
    """hv_integrate_rgbd_points_batch: one sort per chunk of frames, same per-voxel fold order."""
    import torch
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import VoxelBlockGrid

    s = SyntheticRGBD("tiny_160x120_2cm")
    depth, rgb, T = s.batch(0, 7)
    npx = s.width * s.height
    one = VoxelBlockGrid(0.02, 8, max_blocks=1 << 13, max_points=npx)
    for f in range(7):
        one.integrate_rgbd(depth[f], rgb[f], *s.intrinsics, T[f], max_depth=4.0)
    for max_points, dev in ((3 * npx, False), (8 * npx, True)):  # chunks of 3 frames / one chunk; host and device inputs
        g = VoxelBlockGrid(0.02, 8, max_blocks=1 << 13, max_points=max_points)
        d, c = (torch.from_numpy(depth).cuda(), torch.from_numpy(rgb).cuda()) if dev else (depth, rgb)
        g.integrate_rgbd_batch(d, c, *s.intrinsics, T, max_depth=4.0)
        for a, b in zip(g.dump(), one.dump()):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("voxel,bs", [(0.005, 8), (0.05, 8), (0.1, 16)])
def test_bucket_path_equals_sort_path_and_reference(voxel, bs, monkeypatch):
    """Single frames take the per-call bin path (bin pass + LDS sort / ordered fold; hv_bins.h: "bucket" below), larger inputs and
    HV_VG_PATH=sort the device-wide radix sort.  Both fold a voxel's points in point order: identical bits, and identical
    to the compiled reference.  Coarse voxels (5 cm x 8 = 40 cm blocks, 10 cm x 16 = 1.6 m blocks) put far more than the
    4096-entry LDS window into one block: the windowed fold of oversized buckets is covered too."""
    from pyslam_amd.volumetric import VoxelBlockGrid

    s, frames = synthetic_frames("synthetic_640x480_5mm", 20, 3)
    ref = make_oracle(voxel, bs)
    grids = {}
    for path in ("bucket", "sort"):
        if path == "sort":
            monkeypatch.setenv("HV_VG_PATH", "sort")
        g = VoxelBlockGrid(voxel, bs, max_blocks=1 << 17, max_points=1 << 19)
        for depth, rgb, T in frames:
            g.integrate_rgbd(depth, rgb, *s.intrinsics, T, max_depth=4.0)
        grids[path] = g
    for depth, rgb, T in frames:
        pts, cols, _ = hp.frame_to_world_f32(depth, rgb, *s.intrinsics, T, 4.0)
        ref.integrate(pts, cols)
    assert_same_grid(grids["bucket"], ref)
    for a, b in zip(grids["bucket"].dump(), grids["sort"].dump()):
        np.testing.assert_array_equal(a, b)
    if voxel >= 0.05:  # blocks really caught more points than one LDS window: 3 frames x 300 k points over few blocks
        assert grids["bucket"].dump()[2].sum(axis=1).max() > 3 * 4096


def float64_border_points(rng, voxel, n):
    """float64 points whose float32 narrowing lands in the neighbouring cell: a hair below / above exact multiples of the
    (float32) voxel size, block borders, plus an ordinary cloud."""
    vs = float(np.float32(voxel))
    k = rng.integers(-4000, 4000, size=(n, 3)).astype(np.float64)
    below = k * vs * (1.0 - 1e-12)
    above = k * vs * (1.0 + 1e-12)
    blocks = (rng.integers(-500, 500, size=(n, 3)) * 8).astype(np.float64) * vs - 1e-13
    rand = (rng.random((n, 3)) - 0.5) * 40.0
    return np.concatenate([below, above, blocks, rand])


@pytest.mark.skipif(not oracle.ref_available(), reason="compiled reference not available")
@pytest.mark.parametrize("path", ["bucket", "sort"])
@pytest.mark.parametrize("bs", [8, 5])
def test_float64_points_take_the_double_overload(path, bs, monkeypatch):
    """VoxelBlockGrid.integrate(float64 points) == the binding's py::array_t<double> overload (volumetric_grid_module.h:738-741):
    keys from the doubles, float32-narrowed sums - bitwise against the compiled reference, on the per-frame bucket path and on
    the radix path, power-of-two and odd block sizes, numpy and device-resident inputs; and NOT what narrowing first gives."""
    import torch

    from pyslam_amd.volumetric import VoxelBlockGrid

    if path == "sort":
        monkeypatch.setenv("HV_VG_PATH", "sort")
    rng = np.random.default_rng(5)
    pts = float64_border_points(rng, 0.005, 50_000)
    cols = rng.integers(0, 256, size=(len(pts), 3), dtype=np.uint8)
    ref64, ref32 = oracle.RefGrid(0.005, bs), oracle.RefGrid(0.005, bs)
    ref64.integrate(pts, cols)
    ref32.integrate(pts.astype(np.float32), cols)
    assert not np.array_equal(ref64.dump()[2], ref32.dump()[2]) or not np.array_equal(ref64.dump()[0], ref32.dump()[0])
    g = VoxelBlockGrid(0.005, bs, max_blocks=1 << 18, max_points=1 << 18)
    g.integrate(pts, cols)
    assert_same_grid(g, ref64)
    g.integrate(torch.from_numpy(pts).cuda(), torch.from_numpy(cols).cuda())  # device-resident float64
    ref64.integrate(pts, cols)
    assert_same_grid(g, ref64)
    g32 = VoxelBlockGrid(0.005, bs, max_blocks=1 << 18, max_points=1 << 18)
    g32.integrate(pts.astype(np.float32), cols)
    assert_same_grid(g32, ref32)


def test_block_ownership_sharding_union_is_the_single_grid():
    """hv_set_owner on HIP grids: three 'ranks' (three grids on this GPU) see the same frames through every integrate entry point
    (fused RGB-D frame, point array, batched replay); their voxel sets are disjoint, agree with hv_block_owner, and their union is
    the unsharded grid bit for bit."""
    from pyslam_amd.distributed import block_owner
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import VoxelBlockGrid

    s = SyntheticRGBD("tiny_160x120_2cm")
    world = 3
    full = VoxelBlockGrid(0.02, 8, max_blocks=1 << 13, max_points=1 << 18)
    parts = [VoxelBlockGrid(0.02, 8, max_blocks=1 << 13, max_points=1 << 18) for _ in range(world)]
    for r, g in enumerate(parts):
        g.set_owner(r, world)
    frames = [s[i] for i in (0, 5, 9, 20)]
    for g in [full] + parts:
        d, c, T = frames[0]
        g.integrate_rgbd(d, c, *s.intrinsics, T, max_depth=4.0)
        pts, cols, _ = hp.frame_to_world_f32(frames[1][0], frames[1][1], *s.intrinsics, frames[1][2], 4.0)
        g.integrate(pts, cols)
        g.integrate_rgbd_batch(np.stack([f[0] for f in frames[2:]]), np.stack([f[1] for f in frames[2:]]), *s.intrinsics,
                               np.stack([f[2] for f in frames[2:]]), max_depth=4.0)
        assert g.dropped_points() == 0  # a foreign point is not a dropped point
    kf, hf, cf, sf = full.dump()
    dumps = [g.dump() for g in parts]
    for r, (k, _, _, _) in enumerate(dumps):
        assert (block_owner(k, world) == r).all() and len(k) > 0.2 * len(kf)
    keys = np.concatenate([d[0] for d in dumps])
    order = np.lexsort(keys.T[::-1])
    np.testing.assert_array_equal(keys[order], kf)
    np.testing.assert_array_equal(np.concatenate([d[2] for d in dumps])[order], cf)
    np.testing.assert_array_equal(np.concatenate([d[3] for d in dumps])[order].view(np.uint32), sf.view(np.uint32))


def test_mfma_unprojection_variant_fills_the_same_voxels(monkeypatch):
    """NS1 (tools/ns1_mfma.py): the unprojection with its rigid transform on the matrix cores (HV_VG_UNPROJECT=mfma, four
    v_mfma_f64_16x16x4_f64 per wave, radix path) against the default VALU form on posed synthetic frames: the same blocks, the same
    per-voxel counts; position sums within float32 rounding of a fused multiply-add (the matrix core rounds a row's four steps once
    each) - which is why it is NOT the default: the parity contract names the unfused order."""
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import VoxelBlockGrid

    s = SyntheticRGBD("tiny_160x120_2cm")
    monkeypatch.setenv("HV_VG_PATH", "sort")
    dumps = {}
    for form in ("valu", "mfma"):
        monkeypatch.setenv("HV_VG_UNPROJECT", form)
        g = VoxelBlockGrid(0.02, 8, max_blocks=1 << 13, max_points=1 << 16)
        for i in (0, 7, 19):
            d, c, T = s[i]
            g.integrate_rgbd(d, c, *s.intrinsics, T, max_depth=4.0)
        dumps[form] = g.dump()
    ka, _, ca, sa = dumps["valu"]
    kb, _, cb, sb = dumps["mfma"]
    np.testing.assert_array_equal(ka, kb)
    moved = int(np.abs(ca.astype(np.int64) - cb).sum()) // 2
    assert moved <= 2, moved  # a point exactly on a voxel face may move with the last bit of its coordinate; none does on these frames
    same = ca == cb
    np.testing.assert_allclose(sa[same][:, :3], sb[same][:, :3], rtol=0, atol=2e-6 * max(1, int(ca.max())))
    assert ca.sum() > 30000
