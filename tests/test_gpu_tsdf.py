"""GPU parity: TSDF mode vs oracle/tsdf_oracle.c (Open3D ScalableTSDFVolume semantics, restated;
PARITY UNPINNED by the reference — see the oracle header).

Bar (BASELINE.json north_star): unit indices bit-exact; tsdf/weight/colour within 1e-4.  What the
GPU path actually delivers and what is asserted: unit keys, touched sets and weights exact; tsdf
bit-identical (same IEEE op order); colour (exact integer sums / weight) within 1e-4 of the oracle's
double running mean on the [0,1] scale (1e-9 typical)."""
import numpy as np
import pytest

import oracle
from tests.conftest import assert_dumps_match, assert_tsdf_parity, canonical_mesh, sort_rows, synthetic_frames

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north-star tolerance on tsdf / weight / colour ([0,1] scale)


def make_pair(voxel, trunc, **kw):
    from pyslam_amd.volumetric import ScalableTSDFVolume

    return ScalableTSDFVolume(voxel, trunc, **kw), oracle.PortTsdf(voxel, trunc, threads=8)


def integrate_both(gpu, cpu, s, frames, depth_scale=1.0, depth_trunc=4.0):
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage

    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    for depth, rgb, T in frames:
        img = RGBDImage.create_from_color_and_depth(rgb, depth, depth_scale=depth_scale, depth_trunc=depth_trunc,
                                                    convert_rgb_to_intensity=False)
        gpu.integrate(img, K, T)
        cpu.integrate(depth, rgb, K.as_array(), T, depth_scale, depth_trunc)


def assert_same_volume(gpu, cpu, exact_tsdf=True, swept=False):
    """swept: the volume went through the multi-frame sweep, whose default (fold) form holds tsdf to FOLD_TSDF_TOL
    instead of bitwise (conftest.assert_tsdf_parity)."""
    ka, ta, wa, ca = gpu.dump()
    kb, tb, wb, cb = cpu.dump()
    np.testing.assert_array_equal(ka, kb)  # unit indices bit-exact
    np.testing.assert_array_equal(wa, wb)  # weights exact
    if exact_tsdf:
        assert_tsdf_parity(ta, tb, swept)
    assert np.abs(ta - tb).max() <= TOL
    assert np.abs(ca - cb).max() / 255.0 <= TOL


@pytest.mark.parametrize("config,voxel,trunc", [("tiny_160x120_2cm", 0.02, 0.08), ("synthetic_640x480_5mm", 0.005, 0.04),
                                                ("synthetic_640x480_5mm", 0.01, 0.04)])
def test_integrate_matches_oracle(config, voxel, trunc):
    s, frames = synthetic_frames(config, 5, 3)
    gpu, cpu = make_pair(voxel, trunc, max_blocks=1 << 15)
    for f in frames:
        integrate_both(gpu, cpu, s, [f])
        np.testing.assert_array_equal(gpu.touched_keys(), cpu.touched_keys())  # K7 parity per frame
    assert gpu.num_blocks() == cpu.num_units()
    assert_same_volume(gpu, cpu)


def test_headline_config_5mm():
    """BASELINE configs[1]: 640x480 @ 5 mm, sdf_trunc 0.04 — two frames against the oracle."""
    s, frames = synthetic_frames("synthetic_640x480_5mm", 30, 2)
    gpu, cpu = make_pair(0.005, 0.04, max_blocks=1 << 15)
    integrate_both(gpu, cpu, s, frames)
    assert_same_volume(gpu, cpu)


def test_u16_depth_and_scale_trunc():
    """TUM-style uint16 depth with DepthMapFactor 5000 and a depth_trunc that really truncates."""
    s, frames = synthetic_frames("tum1_640x480_5mm", 7, 2, depth_dtype="uint16")
    gpu, cpu = make_pair(0.02, 0.08, max_blocks=1 << 14)
    integrate_both(gpu, cpu, s, frames, depth_scale=5000.0, depth_trunc=2.5)
    assert_same_volume(gpu, cpu)


def test_device_resident_and_batch_equivalence(sweep_form):
    import torch
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, ScalableTSDFVolume

    s, frames = synthetic_frames("tiny_160x120_2cm", 0, 6)
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    a, cpu = make_pair(0.02, 0.08, max_blocks=1 << 13)
    integrate_both(a, cpu, s, frames)
    b = ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 13)
    depth = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
    rgb = torch.from_numpy(np.stack([f[1] for f in frames])).cuda()
    b.integrate_batch(depth, rgb, K, np.stack([f[2] for f in frames]), depth_scale=1.0, depth_trunc=4.0)
    assert_dumps_match(a.dump(), b.dump())
    assert_same_volume(b, cpu, swept=True)


def test_empty_and_invalid_inputs():
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume

    gpu = ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 10)
    K = PinholeCameraIntrinsic(64, 48, 60.0, 60.0, 31.5, 23.5)
    depth = np.zeros((48, 64), np.float32)  # no valid pixel: nothing allocated
    rgb = np.zeros((48, 64, 3), np.uint8)
    gpu.integrate(RGBDImage.create_from_color_and_depth(rgb, depth, 1.0, 4.0, False), K, np.eye(4))
    assert gpu.num_blocks() == 0
    assert gpu.extract_triangle_mesh().vertices.shape == (0, 3)
    assert gpu.extract_point_cloud().points.shape == (0, 3)
    with pytest.raises(RuntimeError, match="Unsupported image format"):
        gpu.integrate(RGBDImage.create_from_color_and_depth(rgb[:, :32], depth, 1.0, 4.0, False), K, np.eye(4))
    with pytest.raises(RuntimeError, match="Unsupported image format"):
        RGBDImage.create_from_color_and_depth(rgb, depth, 1.0, 4.0, True)
    # a pool that is too small grows before anything is fused (the first call verifies its claims) ...
    small = ScalableTSDFVolume(0.02, 0.08, max_blocks=4)
    d = np.full((48, 64), 1.0, np.float32)
    small.integrate(RGBDImage.create_from_color_and_depth(rgb, d, 1.0, 4.0, False), K, np.eye(4))
    assert small.max_blocks() > 4 and small.num_blocks() > 4


def test_marching_cubes_and_point_cloud_match_oracle():
    s, frames = synthetic_frames("tiny_160x120_2cm", 0, 5)
    gpu, cpu = make_pair(0.02, 0.08, max_blocks=1 << 13)
    integrate_both(gpu, cpu, s, frames)
    m = gpu.extract_triangle_mesh()
    vb, tb, cb = cpu.extract_triangle_mesh()
    assert m.vertices.shape == vb.shape and m.triangles.shape == tb.shape
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


def test_mesh_is_watertight_inside_view():
    """Property at full resolution (no oracle): interior mesh edges are shared by exactly two
    triangles with opposite orientation (2-manifold), vertices lie within the truncation band."""
    s, frames = synthetic_frames("synthetic_640x480_5mm", 60, 4)
    gpu, _ = make_pair(0.01, 0.04, max_blocks=1 << 15)
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage

    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    for depth, rgb, T in frames:
        gpu.integrate(RGBDImage.create_from_color_and_depth(rgb, depth, 1.0, 4.0, False), K, T)
    m = gpu.extract_triangle_mesh()
    assert len(m.triangles) > 10000
    assert m.triangles.min() >= 0 and m.triangles.max() < len(m.vertices)
    e = np.concatenate([m.triangles[:, [0, 1]], m.triangles[:, [1, 2]], m.triangles[:, [2, 0]]])
    und = np.sort(e, axis=1)
    _, inv, cnt = np.unique(und, axis=0, return_inverse=True, return_counts=True)
    assert cnt.max() <= 2  # never more than two triangles on an edge
    assert (cnt == 2).mean() > 0.9  # open only along the view/validity boundary
    # opposite orientation on shared edges
    sign = np.where(e[:, 0] < e[:, 1], 1, -1)
    tot = np.zeros(len(cnt), np.int64)
    np.add.at(tot, inv.ravel(), sign)
    assert (tot[cnt == 2] == 0).all()
    assert (m.vertex_colors >= 0).all() and (m.vertex_colors <= 1).all()


def test_incremental_extraction_equals_a_full_pass_at_every_tick(monkeypatch):
    """The per-unit extraction caches (hv_common.h: masks, marching-cubes classification, point counts kept between extractions and
    recomputed only around the units the new frames wrote to) against a full pass over the same volume at every tick of a running
    reconstruction: single keyframes, batches, mesh-only and points-only ticks, a pool that grows under the caches, numerators
    imported behind the stamps' back, a reset - and against the oracle at the end."""
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, ScalableTSDFVolume

    monkeypatch.setenv("HV_TSDF_SWEEP", "2")  # (bitwise form: the oracle comparison at the end is vertex for vertex)
    s, frames = synthetic_frames("tiny_160x120_2cm", 0, 24)
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    inc = ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 6)  # (small: the pool is rebuilt on the way)
    cpu = oracle.PortTsdf(0.02, 0.08)

    def fuse(vol, lo, hi, batch):
        from pyslam_amd.volumetric import RGBDImage

        if batch:
            d, c, T = (np.stack([f[k] for f in frames[lo:hi]]) for k in range(3))
            vol.integrate_batch(d, c, K, T, depth_scale=1.0, depth_trunc=4.0)
        else:
            for d, c, T in frames[lo:hi]:
                vol.integrate(RGBDImage(c, d, 1.0, 4.0), K, T)

    def full_pass(history):
        """a fresh volume with the same history, extracted once (its first extraction: every unit computed)"""
        ref = ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 13)
        for step in history:
            step(ref)
        return ref

    def same(a, b, mesh=True, points=True):
        if mesh:
            ma, mb = a.extract_triangle_mesh(), b.extract_triangle_mesh()
            assert ma.vertices.shape == mb.vertices.shape and ma.triangles.shape == mb.triangles.shape and len(ma.triangles) > 0
            for x, y in zip(canonical_mesh(ma.vertices, ma.triangles, ma.vertex_colors), canonical_mesh(mb.vertices, mb.triangles, mb.vertex_colors)):
                np.testing.assert_array_equal(x, y)
        if points:
            pa, pb = a.extract_point_cloud(), b.extract_point_cloud()
            assert pa.points.shape == pb.points.shape and len(pa.points) > 0
            for x, y in zip(sort_rows(pa.points, pa.colors), sort_rows(pb.points, pb.colors)):
                np.testing.assert_array_equal(x, y)

    history = []
    ticks = [(0, 4, True, True, True), (4, 5, False, True, True), (5, 6, False, False, True), (6, 8, False, True, False),
             (8, 16, True, True, True), (16, 17, False, True, True), (17, 24, True, True, True)]
    for lo, hi, batch, mesh, points in ticks:
        step = (lambda vol, lo=lo, hi=hi, batch=batch: fuse(vol, lo, hi, batch))
        step(inc)
        history.append(step)
        same(inc, full_pass(history), mesh, points)
    for d, c, T in frames:
        cpu.integrate(d, c, K.as_array(), T, 1.0, 4.0)
    m = inc.extract_triangle_mesh()
    vb, tb, cb = cpu.extract_triangle_mesh()
    va, ca, ta = canonical_mesh(m.vertices, m.triangles, m.vertex_colors)
    vb, cb, tb = canonical_mesh(vb, tb, cb)
    np.testing.assert_allclose(va, vb, rtol=0, atol=1e-9)
    np.testing.assert_allclose(ta, tb, rtol=0, atol=1e-9)
    # numerators imported behind the stamps' back (the multi-GPU gather writes voxels without touching a unit's stamp) ...
    keys = inc.unit_keys()
    payload = inc.export_numerators(keys)
    payload[:, :, 0] *= 0.5  # halves every tsdf: the surface moves in every unit
    inc.import_numerators(keys, payload)
    other = full_pass(history)
    other.import_numerators(keys, payload)
    same(inc, other)
    # ... and a reset followed by a different stream
    inc.reset()
    fuse(inc, 10, 14, True)
    same(inc, full_pass([lambda vol: fuse(vol, 10, 14, True)]))


def test_numerators_roundtrip_and_merge():
    """export -> import reproduces the volume; summing two disjoint-frame volumes' numerators equals
    fusing all frames in one volume (the multi-GPU merge identity)."""
    from pyslam_amd.volumetric import ScalableTSDFVolume

    s, frames = synthetic_frames("tiny_160x120_2cm", 0, 4)
    full, cpu = make_pair(0.02, 0.08, max_blocks=1 << 13)
    integrate_both(full, cpu, s, frames)
    a, _ = make_pair(0.02, 0.08, max_blocks=1 << 13)
    b, _ = make_pair(0.02, 0.08, max_blocks=1 << 13)
    integrate_both(a, oracle.PortTsdf(0.02, 0.08), s, frames[:2])
    integrate_both(b, oracle.PortTsdf(0.02, 0.08), s, frames[2:])
    keys = np.unique(np.concatenate([a.unit_keys(), b.unit_keys()]), axis=0)
    merged = a.export_numerators(keys) + b.export_numerators(keys)
    c = ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 13)
    c.import_numerators(keys, merged)
    kc, tc, wc, cc = c.dump()
    kf, tf, wf, cf = full.dump()
    np.testing.assert_array_equal(kc, kf)
    np.testing.assert_array_equal(wc, wf)
    assert np.abs(tc - tf).max() <= TOL
    assert np.abs(cc - cf).max() / 255.0 <= TOL
    # pure round trip is exact in weight/colour and 1-ulp-level in tsdf
    d = ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 13)
    d.import_numerators(kf, full.export_numerators(kf))
    kd, td, wd, cd = d.dump()
    np.testing.assert_array_equal(wd, wf)
    np.testing.assert_array_equal(cd, cf)
    assert np.abs(td - tf).max() <= 1e-6


def test_reset_and_reuse():
    s, frames = synthetic_frames("tiny_160x120_2cm", 0, 2)
    gpu, cpu = make_pair(0.02, 0.08, max_blocks=1 << 13)
    integrate_both(gpu, cpu, s, frames)
    gpu.reset()
    cpu.reset()
    assert gpu.num_blocks() == 0
    integrate_both(gpu, cpu, s, frames[::-1])
    assert_same_volume(gpu, cpu)


def test_sweep_camera_inside_touched_units(sweep_form):
    """Depth samples a few centimetres from the camera: touched units straddle the camera plane, so voxel columns cross
    pc2 = 0 and the multi-frame sweep has to leave its short division chain (EXACT evaluation).  Online and batch forms
    against the oracle, bit-exact."""
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, ScalableTSDFVolume

    s, frames = synthetic_frames("tiny_160x120_2cm", 0, 6)
    rng = np.random.default_rng(5)
    near = []
    for i, (depth, rgb, T) in enumerate(frames):
        d = (0.03 + 0.25 * rng.random(depth.shape)).astype(np.float32)
        d[rng.random(depth.shape) < 0.1] = 0.0
        near.append((d, rgb, T))
    gpu, cpu = make_pair(0.01, 0.04, max_blocks=1 << 14)
    integrate_both(gpu, cpu, s, near)
    assert_same_volume(gpu, cpu)
    b = ScalableTSDFVolume(0.01, 0.04, max_blocks=1 << 14)
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    b.integrate_batch(np.stack([f[0] for f in near]), np.stack([f[1] for f in near]), K, np.stack([f[2] for f in near]),
                      depth_scale=1.0, depth_trunc=4.0)
    assert_same_volume(b, cpu, swept=True)


@pytest.mark.parametrize("w0", [(1 << 24) - 70, (1 << 24) - 3])
def test_sweep_weights_near_2_pow_24(w0, sweep_form):
    """Voxels observed ~16.7 M times: the sweep's float-weight running mean stops being exact at 2^24, so it must switch to
    the integer-weight form (first across the 2^24 - 64 guard, then across 2^24 itself).  Batch == online, bit-exact."""
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, ScalableTSDFVolume

    s, frames = synthetic_frames("tiny_160x120_2cm", 0, 16)
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    seed, _ = make_pair(0.02, 0.08, max_blocks=1 << 13)
    integrate_both(seed, oracle.PortTsdf(0.02, 0.08), s, frames)
    keys = seed.unit_keys()
    payload = np.zeros((len(keys), 16 ** 3, 5), np.float32)
    payload[..., 1] = float(w0)
    payload[..., 0] = 0.25 * float(w0)
    payload[..., 2:] = 64.0 * float(w0)
    a = ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 13)
    b = ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 13)
    a.import_numerators(keys, payload)
    b.import_numerators(keys, payload)
    depth = np.stack([f[0] for f in frames])
    rgb = np.stack([f[1] for f in frames])
    T = np.stack([f[2] for f in frames])
    for lo in (0, 8):
        a.integrate_batch(depth[lo:lo + 8], rgb[lo:lo + 8], K, T[lo:lo + 8], depth_scale=1.0, depth_trunc=4.0)
    integrate_both(b, oracle.PortTsdf(0.02, 0.08), s, frames)
    da, db = a.dump(), b.dump()
    assert int(da[2].max()) > w0 + 4  # weights really moved past the guard
    assert_dumps_match(da, db)


def test_sweep_multiplier_table_follows_intrinsics():
    """The sweep's per-pixel multiplier table is keyed on (intrinsics, image size): batches from two different cameras
    into one volume, interleaved, against the oracle."""
    from pyslam_amd.volumetric import PinholeCameraIntrinsic

    sa, fa = synthetic_frames("tiny_160x120_2cm", 0, 6)
    sb, fb = synthetic_frames("tum1_640x480_5mm", 3, 4)
    gpu, cpu = make_pair(0.02, 0.08, max_blocks=1 << 14)
    for s, frames in ((sa, fa[:3]), (sb, fb[:2]), (sa, fa[3:]), (sb, fb[2:])):
        K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
        gpu.integrate_batch(np.stack([f[0] for f in frames]), np.stack([f[1] for f in frames]), K,
                            np.stack([f[2] for f in frames]), depth_scale=1.0, depth_trunc=4.0)
        for depth, rgb, T in frames:
            cpu.integrate(depth, rgb, K.as_array(), T, 1.0, 4.0)
    assert_same_volume(gpu, cpu, swept=True)


@pytest.mark.parametrize("box_bits", ["0", "8", "2048"])
def test_touch_pass_paths_open_the_same_units(box_bits, monkeypatch):
    """The touch pass enumerates a sample patch's units through an LDS bitmap of their bounding box, or - when the box
    is too large - sample by sample with ballot de-duplication.  HV_TSDF_TOUCH_BOX_BITS (read at volume creation)
    bounds the box: 0 forces the general path everywhere, 8 mixes both, 2048 is the default.  Same units, same volume,
    online and multi-frame, including a truncation band wider than a unit (27+ units per sample)."""
    from pyslam_amd.volumetric import PinholeCameraIntrinsic

    monkeypatch.setenv("HV_TSDF_TOUCH_BOX_BITS", box_bits)
    s, frames = synthetic_frames("tiny_160x120_2cm", 2, 4)
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    for voxel, trunc in ((0.02, 0.08), (0.01, 0.2)):  # unit 32 cm / band 8 cm; unit 16 cm / band 20 cm
        gpu, cpu = make_pair(voxel, trunc, max_blocks=1 << 14)
        integrate_both(gpu, cpu, s, frames[:2])
        np.testing.assert_array_equal(gpu.touched_keys(), cpu.touched_keys())
        gpu.integrate_batch(np.stack([f[0] for f in frames[2:]]), np.stack([f[1] for f in frames[2:]]), K,
                            np.stack([f[2] for f in frames[2:]]), depth_scale=1.0, depth_trunc=4.0)
        for depth, rgb, T in frames[2:]:
            cpu.integrate(depth, rgb, K.as_array(), T, 1.0, 4.0)
        assert gpu.dropped_points() == 0
        assert_same_volume(gpu, cpu, swept=True)


@pytest.mark.parametrize("u16", [False, True])
def test_integrate_frames_host_staging_matches_device_batches(u16, monkeypatch):
    """hv_tsdf_integrate_frames: host keyframes given one pageable array per frame go through the pipelined staging
    (worker threads -> two page-locked slots -> copy stream -> two device sets; 1 MB sub-chunks here so that slots and
    sets are reused several times, back-to-back calls, a call of one frame) and must give exactly the volume of the same
    frames fused from device-resident batches - and the oracle's."""
    import torch
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, ScalableTSDFVolume

    monkeypatch.setenv("HV_STAGE_CHUNK_MB", "1")
    s, frames = synthetic_frames("tiny_160x120_2cm", 0, 61)
    if u16:
        frames = [((f[0] * 5000.0).astype(np.uint16), f[1], f[2]) for f in frames]
    scale = 5000.0 if u16 else 1.0
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    a = ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 13)
    b = ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 13)
    cpu = oracle.PortTsdf(0.02, 0.08, threads=8)
    cuts = [0, 24, 25, 45, 61]  # 24 frames, 1 frame, 20 frames, 16 frames
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        part = frames[lo:hi]
        T = np.stack([f[2] for f in part])
        # separate allocations, deliberately not contiguous with each other
        a.integrate_frames([np.array(f[0]) for f in part], [np.array(f[1]) for f in part], K, T, depth_scale=scale, depth_trunc=4.0)
        b.integrate_batch(torch.from_numpy(np.stack([f[0] for f in part])).cuda(), torch.from_numpy(np.stack([f[1] for f in part])).cuda(),
                          K, T, depth_scale=scale, depth_trunc=4.0)
        for d, c, Tcw in part:
            cpu.integrate(d, c, K.as_array(), Tcw, scale, 4.0)
    for x, y in zip(a.dump(), b.dump()):
        np.testing.assert_array_equal(x, y)
    assert_same_volume(a, cpu, swept=True)


def test_registered_host_frames_are_fused_in_place_and_bgr_order():
    """The front's transport: keyframes lie in ONE page-locked host segment (hv_host_register - the shared-memory ring) in OpenCV's
    B, G, R order; hv_tsdf_integrate_frames DMAs them in place (no staging copy) and the pack kernel swaps the channels
    (hv_tsdf_set_color_order).  Must equal the pageable / R, G, B path bit for bit, for the multi-frame sweep and the online path."""
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume

    s, frames = synthetic_frames("tiny_160x120_2cm", 0, 40)
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    H, W = s.height, s.width
    slot = (H * W * 7 + 255) // 256 * 256
    ring = np.zeros(slot * len(frames) + 4096, np.uint8)
    base = (-ring.ctypes.data) % 4096  # page-aligned start inside the allocation
    a = ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 13)
    b = ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 13)
    a.set_color_order(bgr=True)
    a.register_host_memory(ring.ctypes.data + base, slot * len(frames))
    try:
        depths, colors = [], []
        for i, (d, c, T) in enumerate(frames):
            at = base + i * slot
            dv = ring[at:at + H * W * 4].view(np.float32).reshape(H, W)
            cv = ring[at + H * W * 4:at + H * W * 7].reshape(H, W, 3)
            dv[...] = d
            cv[...] = c[..., ::-1]  # B, G, R
            depths.append(dv)
            colors.append(cv)
        T = np.stack([f[2] for f in frames])
        for lo, hi in ((0, 24), (24, 25), (25, 40)):
            a.integrate_frames(depths[lo:hi], colors[lo:hi], K, T[lo:hi], depth_scale=1.0, depth_trunc=4.0)
            # the call returned: the slots may be overwritten at once (what the front does when it releases them)
            saved = [(depths[i].copy(), colors[i].copy()) for i in range(lo, hi)]
            for i in range(lo, hi):
                depths[i][...] = 0.0
                colors[i][...] = 0
            b.integrate_frames([x[0] for x in saved], [np.ascontiguousarray(x[1][..., ::-1]) for x in saved], K, T[lo:hi], depth_scale=1.0,
                               depth_trunc=4.0)
        for x, y in zip(a.dump(), b.dump()):
            np.testing.assert_array_equal(x, y)
        # online path, BGR order
        d, c, Tcw = frames[3]
        a.integrate(RGBDImage.create_from_color_and_depth(np.ascontiguousarray(c[..., ::-1]), d, depth_scale=1.0, depth_trunc=4.0,
                                                          convert_rgb_to_intensity=False), K, Tcw)
        b.integrate(RGBDImage.create_from_color_and_depth(c, d, depth_scale=1.0, depth_trunc=4.0, convert_rgb_to_intensity=False), K, Tcw)
        for x, y in zip(a.dump(), b.dump()):
            np.testing.assert_array_equal(x, y)
    finally:
        a.synchronize()
        a.unregister_host_memory(ring.ctypes.data + base)


def test_point_cloud_normals_match_oracle(tmp_path):
    """extract_point_cloud(normals=True): Open3D's GetNormalAt on the GPU vs the restatement (same double arithmetic on the same
    tsdf values: 1e-9), and the PLY the save path writes carries them (x y z nx ny nz r g b, Open3D's property order)."""
    from pyslam_amd.dense.ply_io import read_ply, read_ply_normals, write_ply_points

    s, frames = synthetic_frames("tiny_160x120_2cm", 0, 5)
    gpu, cpu = make_pair(0.02, 0.08, max_blocks=1 << 13)
    integrate_both(gpu, cpu, s, frames)
    pc = gpu.extract_point_cloud(normals=True)
    pb, qb = cpu.extract_point_cloud()
    nb = cpu.point_normals(pb)
    assert pc.has_normals() and pc.points.shape == pb.shape and len(pb) > 5000
    key_a, key_b = np.round(pc.points, 9), np.round(pb, 9)
    ia, ib = np.lexsort(key_a.T[::-1]), np.lexsort(key_b.T[::-1])
    np.testing.assert_allclose(pc.points[ia], pb[ib], rtol=0, atol=1e-9)
    np.testing.assert_allclose(pc.normals[ia], nb[ib], rtol=0, atol=1e-9)
    lens = np.linalg.norm(pc.normals, axis=1)
    assert ((np.abs(lens - 1.0) < 1e-9) | (lens == 0)).all()
    assert gpu.extract_point_cloud().normals is None  # not computed unless asked for
    path = str(tmp_path / "dense_map.ply")
    write_ply_points(path, pc.points, pc.colors, pc.normals)
    pts, cols, faces = read_ply(path)
    np.testing.assert_array_equal(pts, pc.points)
    np.testing.assert_array_equal(read_ply_normals(path), pc.normals)
    assert faces is None and cols.shape == (len(pts), 3)


def _synthetic_unit_states(seed, keys, voxel=0.01, trunc=0.04, R=16):
    """Voxel states [U, R, R, R] (x, y, z) that no depth image produces: a sphere and a tilted plane crossing unit borders along
    every axis, with 8 % of the voxels overwritten by values on the decision boundaries of the two extractions (+-0, +-0.98 and
    their float neighbours, +-1) and 10 % never observed (weight 0, tsdf 0 as in Open3D)."""
    rng = np.random.default_rng(seed)
    idx = np.stack(np.meshgrid(np.arange(R), np.arange(R), np.arange(R), indexing="ij"), -1).astype(np.float64)
    U = len(keys)
    tsdf = np.zeros((U, R, R, R), np.float32)
    centre, radius = np.array([-0.07, -0.10, -0.06]), 0.13
    n = np.array([0.3, -0.5, 0.81]) / np.linalg.norm([0.3, -0.5, 0.81])
    for k, key in enumerate(keys):
        p = (np.asarray(key, np.float64)[None, None, None, :] * R + idx + 0.5) * voxel
        d = np.minimum(np.linalg.norm(p - centre, axis=-1) - radius, p @ n + 0.02)
        tsdf[k] = np.clip(d / trunc, -1.0, 1.0).astype(np.float32)
    b = np.float32(0.98)
    special = np.array([-1.0, -b, np.nextafter(-b, np.float32(0)), np.nextafter(-b, np.float32(-2)), -0.5, -1e-3, -0.0, 0.0, 1e-3, 0.5,
                        np.nextafter(b, np.float32(0)), b, np.nextafter(b, np.float32(2)), 1.0], np.float32)
    m = rng.random(tsdf.shape) < 0.08
    tsdf[m] = rng.choice(special, int(m.sum()))
    weight = np.ones(tsdf.shape, np.float32)
    unobserved = rng.random(tsdf.shape) < 0.10
    weight[unobserved] = 0.0
    tsdf[unobserved] = 0.0
    colour = rng.integers(0, 256, tsdf.shape + (3,)).astype(np.float64)
    return tsdf, weight, colour


@pytest.mark.parametrize("seed", [1, 2])
def test_extraction_of_loaded_voxel_states_matches_oracle(seed):
    """Both extractions on voxel states handed to both sides directly (hv_tsdf_import_numerators / the oracle's load hook): a
    3 x 3 x 3 cluster of units with holes, negative unit indices and one isolated unit, values ON the decision boundaries
    (tsdf = +-0: no sign change; +-0.98: the point cloud's half-open range; unobserved voxels inside the surface; missing
    neighbour units on every side).  Mesh vertices / triangles / colours, points / colours and normals against the oracle."""
    import itertools

    rng = np.random.default_rng(100 + seed)
    R = 16
    cluster = [k for k in itertools.product(range(-2, 1), repeat=3) if rng.random() > 0.2]
    keys = np.array(cluster + [(4, 5, -3)], np.int32)
    tsdf, weight, colour = _synthetic_unit_states(seed, keys)
    gpu, cpu = make_pair(0.01, 0.04, max_blocks=256)
    cpu.load_units(keys, tsdf, weight, colour)
    word = lambda a: a.transpose(0, 3, 1, 2).reshape(len(keys), R ** 3)  # product voxel order: z * 256 + x * 16 + y
    payload = np.zeros((len(keys), R ** 3, 5), np.float32)
    payload[..., 0] = word(tsdf * weight)
    payload[..., 1] = word(weight)
    for c in range(3):
        payload[..., 2 + c] = word(colour[..., c].astype(np.float32) * weight)
    gpu.import_numerators(keys, payload)
    assert gpu.num_blocks() == len(keys)
    ka, ta, wa, _ = gpu.dump()
    kb, tb, wb, _ = cpu.dump()
    np.testing.assert_array_equal(ka, kb)
    np.testing.assert_array_equal(ta.view(np.uint32), tb.view(np.uint32))  # incl. the sign of -0
    np.testing.assert_array_equal(wa, wb)
    m = gpu.extract_triangle_mesh()
    vb, tb_, cb = cpu.extract_triangle_mesh()
    assert m.vertices.shape == vb.shape and m.triangles.shape == tb_.shape and len(tb_) > 2000
    va, ca, tra = canonical_mesh(m.vertices, m.triangles, m.vertex_colors)
    vb, cb, trb = canonical_mesh(vb, tb_, cb)
    np.testing.assert_allclose(va, vb, rtol=0, atol=1e-9)
    np.testing.assert_allclose(tra, trb, rtol=0, atol=1e-9)
    # (coincident vertices - an exact 0 on a voxel corner - carry that corner's colour on every edge: ties are harmless)
    np.testing.assert_allclose(ca, cb, rtol=0, atol=TOL)
    pc = gpu.extract_point_cloud(normals=True)
    pb, qb = cpu.extract_point_cloud()
    nb = cpu.point_normals(pb)
    assert pc.points.shape == pb.shape and len(pb) > 500
    ia, ib = np.lexsort(np.round(pc.points, 9).T[::-1]), np.lexsort(np.round(pb, 9).T[::-1])
    np.testing.assert_allclose(pc.points[ia], pb[ib], rtol=0, atol=1e-9)
    np.testing.assert_allclose(pc.colors[ia], qb[ib], rtol=0, atol=TOL)
    np.testing.assert_allclose(pc.normals[ia], nb[ib], rtol=0, atol=1e-9)


def test_device_resident_extraction_equals_host_extraction():
    """extract_triangle_mesh / extract_point_cloud(device=True): the same arrays as torch CUDA tensors (no PCIe copy)."""
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume

    s, frames = synthetic_frames("tiny_160x120_2cm", 0, 4)
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    vol = ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 12)
    for d, c, T in frames:
        vol.integrate(RGBDImage.create_from_color_and_depth(c, d, 1.0, 4.0, False), K, T)
    host = vol.extract_triangle_mesh()
    dev = vol.extract_triangle_mesh(device=True)
    assert dev.vertices.is_cuda and dev.triangles.is_cuda and len(host.vertices) > 100
    np.testing.assert_array_equal(dev.vertices.cpu().numpy(), host.vertices)
    np.testing.assert_array_equal(dev.vertex_colors.cpu().numpy(), host.vertex_colors)
    np.testing.assert_array_equal(dev.triangles.cpu().numpy(), host.triangles)
    ph = vol.extract_point_cloud(normals=True)
    pd = vol.extract_point_cloud(normals=True, device=True)
    np.testing.assert_array_equal(pd.points.cpu().numpy(), ph.points)
    np.testing.assert_array_equal(pd.colors.cpu().numpy(), ph.colors)
    np.testing.assert_array_equal(pd.normals.cpu().numpy(), ph.normals)


def test_float32_extraction_is_the_float64_result_rounded_once():
    """extract_triangle_mesh / extract_point_cloud(dtype=np.float32) (hv_tsdf_extract_mesh_f32 / _points_f32): Open3D's float64 rows
    cast to float32 - numpy's astype, bit for bit - with the same triangles, in any order of the calls (a result cached in the
    other type is computed again), on host arrays and on device tensors, and after further keyframes (incremental extraction)."""
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume

    s, frames = synthetic_frames("tiny_160x120_2cm", 0, 6)
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    vol = ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 12)
    for k, (d, c, T) in enumerate(frames):
        vol.integrate(RGBDImage.create_from_color_and_depth(c, d, 1.0, 4.0, False), K, T)
        if k < 3:
            continue
        m32 = vol.extract_triangle_mesh(dtype=np.float32)  # float32 first on a changed volume ...
        m64 = vol.extract_triangle_mesh()  # ... then float64 of the same contents
        again = vol.extract_triangle_mesh(dtype="float32")
        assert m32.vertices.dtype == np.float32 and m32.vertex_colors.dtype == np.float32 and len(m64.vertices) > 100
        np.testing.assert_array_equal(m32.vertices, m64.vertices.astype(np.float32))
        np.testing.assert_array_equal(m32.vertex_colors, m64.vertex_colors.astype(np.float32))
        np.testing.assert_array_equal(m32.triangles, m64.triangles)
        np.testing.assert_array_equal(again.vertices, m32.vertices)
        p64 = vol.extract_point_cloud(normals=True)
        p32 = vol.extract_point_cloud(normals=True, dtype=np.float32)  # the normals are taken at the float64 points
        assert p32.points.dtype == np.float32 and p32.normals.dtype == np.float64 and len(p64.points) > 100
        np.testing.assert_array_equal(p32.points, p64.points.astype(np.float32))
        np.testing.assert_array_equal(p32.colors, p64.colors.astype(np.float32))
        np.testing.assert_array_equal(p32.normals, p64.normals)
        np.testing.assert_array_equal(vol.extract_point_cloud().points, p64.points)  # (float64 again after a float32 result)
    dev = vol.extract_triangle_mesh(device=True, dtype=np.float32)
    assert dev.vertices.is_cuda and str(dev.vertices.dtype) == "torch.float32"
    np.testing.assert_array_equal(dev.vertices.cpu().numpy(), m32.vertices)
    np.testing.assert_array_equal(dev.vertex_colors.cpu().numpy(), m32.vertex_colors)
    pdev = vol.extract_point_cloud(device=True, dtype=np.float32)
    np.testing.assert_array_equal(pdev.points.cpu().numpy(), p32.points)
    with pytest.raises(TypeError):
        vol.extract_triangle_mesh(dtype=np.float16)
    empty = ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 10).extract_triangle_mesh(dtype=np.float32)
    assert empty.vertices.shape == (0, 3) and empty.vertices.dtype == np.float32 and empty.triangles.shape == (0, 3)


@pytest.mark.parametrize("path", ["online", "batch"])
def test_hip_volume_equals_the_closed_form_evaluator_on_full_frames(path):
    """the HIP volume held to tests/tsdf_closed_form.py - a float64, closed-form, whole-frame evaluator that
    shares no helper and no structure with oracle/tsdf_oracle.c - on three full 640x480 frames of an analytic scene (tilted plane
    + sphere) at the headline's 5 mm / sdf_trunc 0.04 m: unit set exact, the weight of every voxel whose decisions are not within
    rounding distance of a boundary exact (98.5 % of 13.3 M voxels), tsdf <= 6e-5, colours exact.  Both entry points: one
    hv_tsdf_integrate per frame, and the multi-frame sweep (hv_tsdf_integrate_batch, production fold form)."""
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume
    from tests import tsdf_closed_form as cf

    fr = cf.frames()
    ref = cf.evaluate(fr)
    K = PinholeCameraIntrinsic(cf.W, cf.H, *cf.K)
    vol = ScalableTSDFVolume(cf.VOXEL, cf.TRUNC, max_blocks=1 << 14)
    if path == "online":
        for d, c, T in fr:
            vol.integrate(RGBDImage.create_from_color_and_depth(c, d, 1.0, cf.DEPTH_TRUNC, False), K, T)
    else:
        vol.integrate_batch(np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]), K, np.stack([f[2] for f in fr]), depth_scale=1.0,
                            depth_trunc=cf.DEPTH_TRUNC)
    stats = cf.compare(vol.dump(), ref, f"hipvol {path}")
    assert stats["units"] > 3000 and stats["max_weight"] == 3 and stats["updated"] > 4_000_000 and stats["fragile_frac"] < 0.05
