"""GPU parity of the EXACT code path and configuration bench.py times.

bench.py's step = ``ScalableTSDFVolume.integrate_batch`` with B = 64 posed frames (32 in rounds 1-5 and in the ``batch32`` leg) of
``synthetic_640x480_5mm`` at voxel 0.005 m / sdf_trunc 0.04 m / depth_trunc 4 m, batches taken as a sliding
window over the stream (step k fuses frames Bk .. Bk+B-1 into the same volume).  These tests run that call
with those arguments against ``oracle.PortTsdf`` and compare the FULL dump: unit keys and weights exact, colour
<= 1e-4 (north-star tolerance; reference call sites pyslam/dense/volumetric_integrator_tsdf.py:215-223,260), tsdf
within ``FOLD_TSDF_TOL`` = 5e-6 for the production (fold) form of the sweep - which applies one running-mean step per
voxel and batch instead of one per frame - and bitwise for the forms that replay the reference's chain frame by
frame (``HV_TSDF_SWEEP=2``).  Mesh and point-cloud extraction are compared at the
same configuration and at BASELINE configs[2] (Replica-shaped 1200x680 @ 4 mm).
"""
import os

import numpy as np
import pytest

import oracle
from tests.conftest import (FOLD_TSDF_TOL, assert_dumps_match, assert_tsdf_parity, canonical_mesh, sort_rows, sweep_is_bitwise,
                            synthetic_frames)

pytestmark = pytest.mark.gpu

TOL = 1e-4
VOXEL, SDF_TRUNC, DEPTH_TRUNC, B = 0.005, 0.04, 4.0, 32  # == bench.py VOXEL / SDF_TRUNC / DEPTH_TRUNC; B: the batch32 leg (the default step of 64 frames: test_bench_step_sliding_window_matches_oracle[*-64])
THREADS = min(32, os.cpu_count() or 1)


def assert_same_volume_chunked(gpu, cpu):
    """Full comparison, evaluated a slice of units at a time (a 5 mm volume dump is ~1 GB per side)."""
    ka, ta, wa, ca = gpu.dump()
    kb, tb, wb, cb = cpu.dump()
    np.testing.assert_array_equal(ka, kb)  # unit indices bit-exact
    np.testing.assert_array_equal(wa, wb)  # weights exact
    assert_tsdf_parity(ta, tb)  # bitwise (forms 1, 2) / <= FOLD_TSDF_TOL (fold form)
    worst = 0.0
    for lo in range(0, len(ka), 512):
        worst = max(worst, float(np.abs(ca[lo:lo + 512] - cb[lo:lo + 512]).max()))
    assert worst / 255.0 <= TOL
    return len(ka), int(wa.max())


def batch_arrays(frames):
    return np.stack([f[0] for f in frames]), np.stack([f[1] for f in frames]), np.stack([f[2] for f in frames])


@pytest.mark.parametrize("batch", [32, 64])
def test_bench_step_sliding_window_matches_oracle(sweep_form, batch):
    """The bench's steps over frames 0..63 through integrate_batch, device-resident inputs as in bench.py - two consecutive steps of
    32 frames (the batch size of rounds 1-5, bench.py's ``batch32`` leg) or ONE step of 64 (bench.py's default since round 6: the width
    of the sweep's frame mask) - against the oracle fusing the same 64 frames one by one."""
    import torch
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, ScalableTSDFVolume

    s, frames = synthetic_frames("synthetic_640x480_5mm", 0, 64)
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    gpu = ScalableTSDFVolume(VOXEL, SDF_TRUNC, max_blocks=1 << 15)
    cpu = oracle.PortTsdf(VOXEL, SDF_TRUNC, threads=THREADS)
    for step in range(64 // batch):
        depth, rgb, T = batch_arrays(frames[step * batch:(step + 1) * batch])
        gpu.integrate_batch(torch.from_numpy(depth).cuda(), torch.from_numpy(rgb).cuda(), K, T, depth_scale=1.0,
                            depth_trunc=DEPTH_TRUNC)
        for f in frames[step * batch:(step + 1) * batch]:
            cpu.integrate(f[0], f[1], K.as_array(), f[2], 1.0, DEPTH_TRUNC)
        assert gpu.num_blocks() == cpu.num_units()
    n_units, w_max = assert_same_volume_chunked(gpu, cpu)
    assert n_units > 6000 and w_max >= 64 - 8  # revisits really happened across the 64 frames
    assert gpu.dropped_points() == 0


def test_bench_step_replay_equals_online(sweep_form):
    """The replay form of the step (the same 32 frames fused twice, bench.py --window replay) equals 64 online
    integrate() calls: keys, weights and colour sums identical, tsdf bitwise (form 2) / within FOLD_TSDF_TOL (fold)."""
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume

    s, frames = synthetic_frames("synthetic_640x480_5mm", 100, B)
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    a = ScalableTSDFVolume(VOXEL, SDF_TRUNC, max_blocks=1 << 15)
    b = ScalableTSDFVolume(VOXEL, SDF_TRUNC, max_blocks=1 << 15)
    depth, rgb, T = batch_arrays(frames)
    for _ in range(2):
        a.integrate_batch(depth, rgb, K, T, depth_scale=1.0, depth_trunc=DEPTH_TRUNC)
        for d, c, Tcw in frames:
            b.integrate(RGBDImage.create_from_color_and_depth(c, d, 1.0, DEPTH_TRUNC, False), K, Tcw)
    assert_dumps_match(a.dump(), b.dump())


def compare_extraction(gpu, cpu, min_triangles):
    m = gpu.extract_triangle_mesh()
    vb, tb, cb = cpu.extract_triangle_mesh()
    assert m.vertices.shape == vb.shape and m.triangles.shape == tb.shape
    assert len(tb) >= min_triangles
    va, ca, ta = canonical_mesh(m.vertices, m.triangles, m.vertex_colors)
    vb, cb, tb = canonical_mesh(vb, tb, cb)
    np.testing.assert_allclose(va, vb, rtol=0, atol=1e-9)
    np.testing.assert_allclose(ca, cb, rtol=0, atol=TOL)
    np.testing.assert_allclose(ta, tb, rtol=0, atol=1e-9)  # identical triangles, as vertex-position triples
    pc = gpu.extract_point_cloud()
    pb, qb = cpu.extract_point_cloud()
    assert pc.points.shape == pb.shape
    pa, qa = sort_rows(np.round(pc.points, 9), pc.colors)
    pb, qb = sort_rows(np.round(pb, 9), qb)
    np.testing.assert_allclose(pa, pb, rtol=0, atol=1e-9)
    np.testing.assert_allclose(qa, qb, rtol=0, atol=TOL)


def test_extraction_matches_oracle_at_headline_config(monkeypatch):
    """Marching cubes + point cloud of a 640x480 / 5 mm volume (8 frames through the bitwise form of the sweep, so that
    both sides extract from identical tsdf values) vs the oracle: identical vertex sets (<= 1e-9), colours <= 1e-4,
    identical triangles after canonical re-indexing."""
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, ScalableTSDFVolume

    monkeypatch.setenv("HV_TSDF_SWEEP", "2")

    s, frames = synthetic_frames("synthetic_640x480_5mm", 0, 8)
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    gpu = ScalableTSDFVolume(VOXEL, SDF_TRUNC, max_blocks=1 << 15)
    cpu = oracle.PortTsdf(VOXEL, SDF_TRUNC, threads=THREADS)
    depth, rgb, T = batch_arrays(frames)
    gpu.integrate_batch(depth, rgb, K, T, depth_scale=1.0, depth_trunc=DEPTH_TRUNC)
    for d, c, Tcw in frames:
        cpu.integrate(d, c, K.as_array(), Tcw, 1.0, DEPTH_TRUNC)
    compare_extraction(gpu, cpu, min_triangles=1_000_000)


def test_extraction_of_fold_form_volume_within_tolerance(monkeypatch):
    """The production (fold) form leaves tsdf within FOLD_TSDF_TOL of the oracle, so the extracted surface cannot be
    compared vertex for vertex: a zero crossing on an edge whose two values differ by 1e-3 moves by ~1e-5 m, and a value
    within 1e-6 of zero may change sign.  Held instead: vertex / triangle / point counts within 0.1 % of the oracle's,
    and 99.9 % of the vertices (points) within 1e-5 m of an oracle vertex (point), all within 1e-3 m (a fifth of a voxel)."""
    from scipy.spatial import cKDTree

    from pyslam_amd.volumetric import PinholeCameraIntrinsic, ScalableTSDFVolume

    monkeypatch.setenv("HV_TSDF_SWEEP", "4")
    s, frames = synthetic_frames("synthetic_640x480_5mm", 0, 8)
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    gpu = ScalableTSDFVolume(VOXEL, SDF_TRUNC, max_blocks=1 << 15)
    cpu = oracle.PortTsdf(VOXEL, SDF_TRUNC, threads=THREADS)
    depth, rgb, T = batch_arrays(frames)
    gpu.integrate_batch(depth, rgb, K, T, depth_scale=1.0, depth_trunc=DEPTH_TRUNC)
    for d, c, Tcw in frames:
        cpu.integrate(d, c, K.as_array(), Tcw, 1.0, DEPTH_TRUNC)
    m = gpu.extract_triangle_mesh()
    vb, tb, _ = cpu.extract_triangle_mesh()
    pc = gpu.extract_point_cloud()
    pb, _ = cpu.extract_point_cloud()
    for mine, theirs in ((m.vertices, vb), (pc.points, pb)):
        assert abs(len(mine) - len(theirs)) <= 1e-3 * len(theirs)
        dist, _ = cKDTree(theirs).query(mine, k=1)
        assert float((dist <= 1e-5).mean()) >= 0.999 and float(dist.max()) <= 1e-3, (float((dist <= 1e-5).mean()), float(dist.max()))
    assert abs(len(m.triangles) - len(tb)) <= 1e-3 * len(tb)


def test_extraction_matches_oracle_replica_4mm(monkeypatch):
    """BASELINE configs[2]: Replica-shaped 1200x680, 4 mm TSDF + colour: fuse (bitwise form of the sweep), extract, compare."""
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, ScalableTSDFVolume

    monkeypatch.setenv("HV_TSDF_SWEEP", "2")

    s, frames = synthetic_frames("replica_1200x680_4mm", 0, 4)
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    gpu = ScalableTSDFVolume(0.004, 0.04, max_blocks=1 << 16, max_points=s.width * s.height)
    cpu = oracle.PortTsdf(0.004, 0.04, threads=THREADS)
    depth, rgb, T = batch_arrays(frames)
    gpu.integrate_batch(depth, rgb, K, T, depth_scale=1.0, depth_trunc=DEPTH_TRUNC)
    for d, c, Tcw in frames:
        cpu.integrate(d, c, K.as_array(), Tcw, 1.0, DEPTH_TRUNC)
    assert gpu.num_blocks() == cpu.num_units()
    compare_extraction(gpu, cpu, min_triangles=1_000_000)


@pytest.fixture(scope="module")
def sweep_case():
    """16 frames of the headline stream + the oracle's volume after fusing them (computed once for the form tests)."""
    s, frames = synthetic_frames("synthetic_640x480_5mm", 40, 16)
    cpu = oracle.PortTsdf(VOXEL, SDF_TRUNC, threads=THREADS)
    K = np.array(s.intrinsics, dtype=np.float64)
    for d, c, Tcw in frames:
        cpu.integrate(d, c, K, Tcw, 1.0, DEPTH_TRUNC)
    return s, frames, cpu.dump()


SWEEP_FORMS = [
    {"HV_TSDF_SWEEP": "4"},                                # production: the batch folded per voxel on whole voxel columns (tsdf <= FOLD_TSDF_TOL, the rest exact)
    {"HV_TSDF_SWEEP": "4", "HV_TSDF_PIPELINE": "0"},     # every launch of a batch on the one stream (no touch / sweep overlap between batches)
    {"HV_TSDF_SWEEP": "4", "HV_TSDF_BATCH_GENERAL": "1"},  # the column kernel's rare-regime path everywhere: bitwise again
    {"HV_TSDF_SWEEP": "4", "HV_TSDF_SWEEP_ZS": "2"},      # z halves: 8 wave tasks per unit (multi-GPU shares; the upper half replays 8 z steps)
    {"HV_TSDF_SWEEP": "4", "HV_TSDF_SWEEP_ZS": "2", "HV_TSDF_BATCH_GENERAL": "1"},
    {"HV_TSDF_SWEEP": "2"},                              # bitwise form: the reference's running mean frame by frame
    {"HV_TSDF_SWEEP": "2", "HV_TSDF_PIPELINE": "0"},
    {"HV_TSDF_SWEEP": "2", "HV_TSDF_BATCH_GENERAL": "1"},  # EXACT evaluation with integer weights everywhere
]


@pytest.mark.parametrize("env", SWEEP_FORMS, ids=lambda e: ",".join(f"{k[8:]}={v}" for k, v in e.items()))
def test_sweep_forms_match_oracle(env, sweep_case, monkeypatch):
    """Every form of the multi-frame sweep (the switches are read per call) against the oracle at the bench
    configuration, two batches of 8 frames: keys and weights exact, colour <= 1e-4, tsdf bitwise for the forms that apply
    the running mean frame by frame (2, and the column kernel's rare-regime path) and <= FOLD_TSDF_TOL for the fold."""
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, ScalableTSDFVolume

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    s, frames, (kb, tb, wb, cb) = sweep_case
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    gpu = ScalableTSDFVolume(VOXEL, SDF_TRUNC, max_blocks=1 << 15)
    for lo in (0, 8):
        depth, rgb, T = batch_arrays(frames[lo:lo + 8])
        gpu.integrate_batch(depth, rgb, K, T, depth_scale=1.0, depth_trunc=DEPTH_TRUNC)
    ka, ta, wa, ca = gpu.dump()
    np.testing.assert_array_equal(ka, kb)
    np.testing.assert_array_equal(wa, wb)
    if env.get("HV_TSDF_BATCH_GENERAL") == "1":
        np.testing.assert_array_equal(ta.view(np.uint32), tb.view(np.uint32))
    else:
        worst = assert_tsdf_parity(ta, tb)
        print(f"[sweep form {env}] max |tsdf - oracle| = {worst:.3e}")
    assert max(float(np.abs(ca[lo:lo + 512] - cb[lo:lo + 512]).max()) for lo in range(0, len(ka), 512)) / 255.0 <= TOL


def test_batch_pipeline_with_interleaved_calls_matches_oracle(sweep_form):
    """The batch pipeline (touch + pack of batch k+1 on a second stream while batch k is swept, two scratch sets) under the
    call patterns that start, break and restart a chain: device-resident batches back to back, an online frame in between,
    an extraction in between, a host-resident batch, a long call that is cut into 64-frame chunks, a reset.  Full dump
    against the oracle fusing the same frames one by one; and bitwise equal to the same calls with the pipeline off."""
    import torch
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume

    s, frames = synthetic_frames("tiny_160x120_2cm", 0, 200)
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)

    def run(gpu, cpu=None):
        def batch(lo, hi, device=True):
            depth, rgb, T = batch_arrays(frames[lo:hi])
            if device:
                depth, rgb = torch.from_numpy(depth).cuda(), torch.from_numpy(rgb).cuda()
            gpu.integrate_batch(depth, rgb, K, T, depth_scale=1.0, depth_trunc=DEPTH_TRUNC)
            if cpu is not None:
                for f in frames[lo:hi]:
                    cpu.integrate(f[0], f[1], K.as_array(), f[2], 1.0, DEPTH_TRUNC)

        def online(i):
            d, c, T = frames[i]
            gpu.integrate(RGBDImage.create_from_color_and_depth(c, d, 1.0, DEPTH_TRUNC, False), K, T)
            if cpu is not None:
                cpu.integrate(d, c, K.as_array(), T, 1.0, DEPTH_TRUNC)

        batch(0, 8)
        gpu.reset()                       # a chain must not survive a reset
        if cpu is not None:
            cpu.reset()
        for lo in range(0, 48, 8):        # six batches back to back: the chain is running from the second on
            batch(lo, lo + 8)
        online(48)                        # breaks the chain
        batch(49, 57)
        batch(57, 65)
        n_pts = len(gpu.extract_point_cloud().points)  # a synchronous reader between two batches
        batch(65, 73)
        batch(73, 81, device=False)       # host-resident frames: staged on the main stream
        batch(81, 89)
        batch(89, 189)                    # 100 frames: two chunks inside one call
        batch(189, 200)
        return n_pts

    vol_size = 0.02, 0.08
    gpu = ScalableTSDFVolume(*vol_size, max_blocks=1 << 13)
    cpu = oracle.PortTsdf(*vol_size, threads=THREADS)
    n_pts = run(gpu, cpu)
    assert n_pts > 1000
    ka, ta, wa, ca = gpu.dump()
    kb, tb, wb, cb = cpu.dump()
    np.testing.assert_array_equal(ka, kb)
    np.testing.assert_array_equal(wa, wb)
    assert_tsdf_parity(ta, tb)
    assert float(np.abs(ca - cb).max()) / 255.0 <= TOL
    os.environ["HV_TSDF_PIPELINE"] = "0"
    try:
        plain = ScalableTSDFVolume(*vol_size, max_blocks=1 << 13)
        run(plain)
    finally:
        del os.environ["HV_TSDF_PIPELINE"]
    for x, y in zip(gpu.dump(), plain.dump()):
        np.testing.assert_array_equal(x, y)
