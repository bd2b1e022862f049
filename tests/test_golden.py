"""CPU: the C restatement (oracle/voxel_oracle.c) against the committed golden vectors that
tools/make_golden.py generated from the *compiled reference* (oracle/_ref).  These fixtures travel
to the GPU box, where /root/reference does not exist; the GPU tests use the same oracle."""
import os

import numpy as np
import pytest

import oracle
from oracle import host_prep as hp
from tools.make_golden import golden_points

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name))


def occupied(grid):
    keys, hashes, counts, sums = grid.dump()
    occ = counts > 0
    return keys, hashes, np.nonzero(occ)[0].astype(np.int32), np.nonzero(occ)[1].astype(np.int16), counts[occ], sums[occ]


def srt(a, b):
    i = np.lexsort(a.T[::-1])
    return a[i], b[i]


@pytest.mark.parametrize("voxel", [0.005, 0.004, 0.002, 0.015])
def test_keys_match_reference_golden(voxel):
    g = load(f"keys_v{int(voxel * 1000):03d}.npz")
    pts, _ = golden_points(11, 1500, voxel)
    vk, bk, lk, h = oracle.keys(pts, voxel, 8, "port")
    np.testing.assert_array_equal(vk, g["voxel_keys"])
    np.testing.assert_array_equal(bk, g["block_keys"])
    np.testing.assert_array_equal(lk, g["local_keys"])
    np.testing.assert_array_equal(h, g["hashes"])
    # SURVEY appendix C probe: key(-0.001, 0.5, 1.25 @ 5 mm) = (-1,100,250), block (-1,12,31), hash ~100
    if voxel == 0.005:
        v, b, l, hh = oracle.keys(np.array([[-0.001, 0.5, 1.25]], np.float32), 0.005, 8, "port")
        assert v.tolist() == [[-1, 100, 250]] and b.tolist() == [[-1, 12, 31]] and l.tolist() == [[7, 4, 2]]
        assert int(hh[0]) == 18446744073709551515


def test_integrate_random_matches_reference_golden():
    g = load("integrate_random.npz")
    pts, cols = golden_points(12, 2500, 0.02)
    grid = oracle.PortGrid(0.02, 8)
    grid.integrate(pts, cols)
    grid.integrate(pts[::3], (cols[::3] * 255).astype(np.uint8))
    grid.integrate(pts[::7])
    keys, hashes, ob, ov, counts, sums = occupied(grid)
    np.testing.assert_array_equal(keys, g["keys"])
    np.testing.assert_array_equal(hashes, g["hashes"])
    np.testing.assert_array_equal(ob, g["occ_block"])
    np.testing.assert_array_equal(ov, g["occ_voxel"])
    np.testing.assert_array_equal(counts, g["counts"])
    np.testing.assert_array_equal(sums.view(np.uint32), g["sums"].view(np.uint32))
    assert grid.size() == int(g["size"])


def test_frame_queries_carve_match_reference_golden():
    from pyslam_amd.synthetic import SyntheticRGBD

    g = load("frame_tiny.npz")
    s = SyntheticRGBD("tiny_160x120_2cm")
    grid = oracle.PortGrid(0.02, 8)
    for i in (0, 1):
        depth, rgb, T = s[i]
        p, c, _ = hp.frame_to_world_f32(depth, rgb, *s.intrinsics, T, 4.0)
        grid.integrate(p, c)
    keys, hashes, ob, ov, counts, sums = occupied(grid)
    np.testing.assert_array_equal(keys, g["keys"])
    np.testing.assert_array_equal(counts, g["counts"])
    np.testing.assert_array_equal(sums.view(np.uint32), g["sums"].view(np.uint32))
    intr = np.array(s.intrinsics, np.float32)
    depth, rgb, T = s[1]
    fp, fc = srt(*grid.get_voxels_in_camera_frustrum(intr, s.width, s.height, T, 3.0, 0.5, 2))
    np.testing.assert_array_equal(fp, g["frustum_points"])
    np.testing.assert_array_equal(fc, g["frustum_colors"])
    bp, bc = srt(*grid.get_voxels_in_bb(np.array([2.0, 1.0, 0.2, 4.0, 3.0, 1.5]), 1))
    np.testing.assert_array_equal(bp, g["bb_points"])
    np.testing.assert_array_equal(bc, g["bb_colors"])
    gv, gc = srt(*grid.get_voxels(3))
    np.testing.assert_array_equal(gv, g["voxels3_points"])
    np.testing.assert_array_equal(gc, g["voxels3_colors"])
    dc = depth.copy()
    dc[:, : s.width // 2] += 0.5
    grid.carve(intr, s.width, s.height, T, 8.0, 0.01, dc, 0.03)
    _, _, counts2, _ = grid.dump()
    assert int(counts2.sum()) == int(g["carved_total"])
    assert int((counts2 > 0).sum()) == int(g["carved_occupied"])


def test_frustum_contains_matches_reference_golden():
    from pyslam_amd.synthetic import SyntheticRGBD

    g = load("frustum_contains.npz")
    s = SyntheticRGBD("tiny_160x120_2cm")
    intr = np.array(s.intrinsics, np.float32)
    T = s.pose(1)
    rng = np.random.default_rng(13)
    P = (rng.random((400, 3)) * np.array([6, 4, 3])).astype(np.float32)
    for i, p in enumerate(P):
        ok, uvd = oracle.frustum_contains(intr, s.width, s.height, T, 8.0, 0.01, p, "port")
        assert ok == bool(g["inside"][i])
        np.testing.assert_array_equal(uvd, g["uvd"][i])
    np.testing.assert_array_equal(oracle.frustum_bbox(intr, s.width, s.height, T, 8.0, 0.01, "port"), g["bbox"])


# ---- host prep: pinned to the reference's own pyslam/utilities/depth.py -------------------
def test_host_prep_matches_reference_depth_fixture():
    """tests/golden/prep_depth.npz was produced by IMPORTING the reference's depth.py (tools/make_golden_prep.py): the
    restatement that travels to the GPU box must reproduce depth2pointcloud and filter_shadow_points bit for bit."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "prep_depth.npz"))
    fx, fy, cx, cy = z["intr"]
    pts, cols, valid = hp.depth2pointcloud(z["depth"], z["rgb"], fx, fy, cx, cy, float(z["max_depth"]), float(z["min_depth"]))
    np.testing.assert_array_equal(pts, z["points"])
    np.testing.assert_array_equal(cols, z["colors"])
    assert valid.sum() == len(z["points"])
    np.testing.assert_array_equal(hp.filter_shadow_points(z["depth"]), z["shadow_mad"])
    np.testing.assert_array_equal(hp.filter_shadow_points(z["depth"], delta_depth=0.05, delta_x=3, delta_y=1, fill_value=0.0), z["shadow_fixed"])
    assert (z["shadow_mad"] == -1).sum() > 0 and (z["shadow_fixed"] != z["depth"]).sum() > 0


@pytest.mark.skipif(not os.path.exists("/root/reference/pyslam/utilities/depth.py"), reason="reference tree not present (GPU box)")
def test_host_prep_matches_imported_reference():
    """Dev container: the reference file itself as the oracle (it needs nothing but numpy), on fresh random inputs."""
    from tools.make_golden_prep import load_reference_depth

    ref = load_reference_depth()
    rng = np.random.default_rng(11)
    for _ in range(3):
        depth = (0.3 + 3.0 * rng.random((37, 53))).astype(np.float32)
        depth[rng.random(depth.shape) < 0.1] = 0.0
        rgb = rng.integers(0, 255, (37, 53, 3)).astype(np.uint8)
        pc = ref.depth2pointcloud(depth, rgb, 60.0, 61.0, 26.0, 18.0, max_depth=3.0, min_depth=0.4)
        pts, cols, _ = hp.depth2pointcloud(depth, rgb, 60.0, 61.0, 26.0, 18.0, 3.0, 0.4)
        np.testing.assert_array_equal(pts, pc.points)
        np.testing.assert_array_equal(cols, pc.colors)
        for kw in ({}, {"delta_depth": 0.1, "delta_x": 1, "delta_y": 3, "fill_value": 0.0}):
            np.testing.assert_array_equal(hp.filter_shadow_points(depth, **kw), ref.filter_shadow_points(depth, **kw))
