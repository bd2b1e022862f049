"""GPU parity of the semantic block grids' full interface against the compiled reference
(oracle/_ref: VoxelBlockSemanticGrid / VoxelBlockSemanticProbabilisticGrid, unmodified sources):
probabilistic payload, carve, assign_object_ids_to_instance_ids + remap_instance_ids in pySLAM's per-frame
flow, get_object_segments with PCA boxes, segment operations."""
import numpy as np
import pytest

import oracle
from oracle.semantic import RefSemGrid2, ref_remap_instance_ids
from tests.semantic_helpers import CFG, DEPTH_MAX, DEPTH_MIN, frame_points, relabel, semantic_frame
from tests.test_semantic_oracle import srt, stream

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not oracle.ref_available(), reason="compiled reference not available")]

VOTE, PROB = 0, 1
VOTE2, PROB2 = 2, 3  # the "*2" payloads (voxel_data_semantic2.h): separate object / class counters; marginal label maps
EXACT = (VOTE, VOTE2)  # integer label state: compared bit for bit (the log-probability payloads: confidences within 2e-6)
ALL_KINDS = [VOTE, PROB, VOTE2, PROB2]


@pytest.fixture(autouse=True)
def restore_reference_statics():
    """The reference keeps the depth threshold / decay rate as process-wide statics of the payload types."""
    yield
    for kind in (VOTE, VOTE2):
        RefSemGrid2(kind, 0.05).set_depth_threshold(10.0)
    for kind in (PROB, PROB2):
        g = RefSemGrid2(kind, 0.05)
        g.set_depth_threshold(5.0)
        g.set_depth_decay_rate(0.07)


def gpu_grid(kind, voxel, **kw):
    from pyslam_amd import volumetric_semantic as vs

    cls = (vs.VoxelBlockSemanticGrid, vs.VoxelBlockSemanticProbabilisticGrid, vs.VoxelBlockSemanticGrid2, vs.VoxelBlockSemanticProbabilisticGrid2)[kind]
    return cls(voxel, 8, max_blocks=kw.get("max_blocks", 1 << 14), max_points=kw.get("max_points", 1 << 18))


def assert_state_equal(gpu, ref, kind, obj_map=None):
    kg, ig, pg, cg, confg = gpu.dump2()[:5]
    kr, ir, pr, cr, confr = ref.dump()
    np.testing.assert_array_equal(kg, kr)
    if obj_map:
        ir = ir.copy()
        ir[..., 1] = relabel(ir[..., 1], obj_map)
    np.testing.assert_array_equal(ig[..., :3], ir[..., :3])  # count, object id, class id
    np.testing.assert_array_equal(pg, pr)
    np.testing.assert_array_equal(cg, cr)
    if kind in EXACT:
        np.testing.assert_array_equal(ig[..., 3], ir[..., 3])
        np.testing.assert_array_equal(confg, confr)
    elif ig.size:
        np.testing.assert_allclose(confg, confr, rtol=0, atol=2e-6)
        assert np.abs(ig[..., 3] - ir[..., 3]).max() <= 1  # (int)(confidence * count)
        assert (confg == confr).mean() > 0.95


@pytest.mark.parametrize("pos_dtype", [np.float32, np.float64])
@pytest.mark.parametrize("use_inst,use_depth", [(True, True), (True, False), (False, True), (False, False)])
def test_probabilistic_stream(pos_dtype, use_inst, use_depth):
    gpu, ref = gpu_grid(PROB, 0.05), RefSemGrid2(PROB, 0.05)
    for g in (gpu, ref):
        g.set_depth_threshold(5.0)
        g.set_depth_decay_rate(0.07)
    for it in range(4):
        pts, cols, cls, inst, dep = stream(500 + it, 60000, pos_dtype)
        cls, inst = cls % 3, inst % 2  # <= 6 distinct (object, class) pairs per voxel: inside the 6 inline label slots
        c = cols if it != 1 else (cols / 255.0).astype(np.float32)
        for g in (gpu, ref):
            g.integrate(pts, c, cls, inst if use_inst else None, dep if use_depth else None)
    assert gpu.dropped_points() == 0 and gpu.label_overflows() == 0
    assert_state_equal(gpu, ref, PROB)
    for mc, mconf in ((1, 0.0), (2, 0.31), (3, 0.55)):
        v = gpu.get_voxels(mc, mconf)
        got = srt((v.points, v.colors, v.class_ids, v.object_ids, v.confidences))
        exp = srt(ref.get_voxels(mc, mconf))
        assert abs(len(got[0]) - len(exp[0])) <= 2  # confidences within 1 ulp of the threshold may flip
        if len(got[0]) == len(exp[0]):
            for a, b in zip(got[:4], exp[:4]):
                np.testing.assert_array_equal(a, b)
            np.testing.assert_allclose(got[4], exp[4], rtol=0, atol=2e-6)


def _many_label_stream(seed, n, n_cls, n_inst):
    pts, cols, _, _, dep = stream(seed, n)
    rng = np.random.default_rng(seed + 77)
    return pts, cols, rng.integers(0, n_cls, n).astype(np.int32), rng.integers(0, n_inst, n).astype(np.int32), dep


def test_probabilistic_label_maps_grow_past_the_inline_slots_like_the_reference():
    """Up to 63 distinct (object, class) pairs per voxel (the reference's std::map is unbounded; here 6 inline slots + chained
    10-pair nodes): every count, sum, arg-max label and confidence as the compiled reference's, nothing dropped.  A segment removal
    resets voxels whose maps had grown chains; the maps they grow afterwards reuse the chains and stay the reference's."""
    gpu, ref = gpu_grid(PROB, 0.2), RefSemGrid2(PROB, 0.2)
    for g in (gpu, ref):
        g.set_depth_threshold(5.0)
        g.set_depth_decay_rate(0.07)
    for it in range(3):
        pts, cols, cls, inst, dep = _many_label_stream(900 + it, 60000, 7, 9)
        for g in (gpu, ref):
            g.integrate(pts, cols, cls, inst, dep if it != 1 else None)
    assert gpu.dropped_points() == 0 and gpu.label_overflows() == 0
    nlab = gpu.dump2(max_labels=64)[5]
    assert nlab.max() > 26  # maps of three nodes and more
    assert_state_equal(gpu, ref, PROB)
    labels = gpu.dump2(max_labels=64)[6]
    v = np.unravel_index(np.argmax(nlab), nlab.shape)
    pairs = labels[v][: nlab[v]]
    assert len({tuple(p) for p in pairs}) == nlab[v] and (pairs >= 0).all()  # a map holds each pair once
    for g in (gpu, ref):
        g.remove_segment(3)
        g.merge_segments(1, 2)
    assert_state_equal(gpu, ref, PROB)
    pts, cols, cls, inst, dep = _many_label_stream(990, 60000, 7, 9)
    for g in (gpu, ref):
        g.integrate(pts, cols, cls, inst, dep)
    assert gpu.label_overflows() == 0
    assert_state_equal(gpu, ref, PROB)
    v = gpu.get_voxels(2, 0.05)
    got, exp = srt((v.points, v.colors, v.class_ids, v.object_ids, v.confidences)), srt(ref.get_voxels(2, 0.05))
    if len(got[0]) == len(exp[0]):  # (confidences within 1 ulp of the threshold may flip)
        for a, b in zip(got[:4], exp[:4]):
            np.testing.assert_array_equal(a, b)
    assert abs(len(got[0]) - len(exp[0])) <= 2


def test_probabilistic_noisy_label_stream_equals_the_reference():
    """The stream tests/test_semantic2_oracle.py::test_label_map_sizes_the_reference_builds measures: 30 labelled keyframes with 5 %
    of the pixels drawing a random (class, object) pair.  The reference's maps grow past 7 pairs in 1-2 % of the voxels (the pairs
    rounds 1-3 dropped); every voxel state and confidence is the compiled reference's."""
    from pyslam_amd.synthetic import SyntheticRGBD

    s = SyntheticRGBD(CFG, noise=True, invalid_frac=0.02)
    gpu, ref = gpu_grid(PROB, 0.02, max_blocks=1 << 15), RefSemGrid2(PROB, 0.02)
    rng = np.random.default_rng(3)
    for k, i in enumerate(range(0, 60, 2)):
        depth, rgb, T, cls, inst = semantic_frame(s, i, shuffle=k)
        flip = rng.random(cls.shape) < 0.05
        cls = np.where(flip, rng.integers(0, 40, cls.shape), cls).astype(np.int32)
        obj = np.where(flip, rng.integers(0, 30, cls.shape), inst).astype(np.int32)
        pts, cols, c, o, d = frame_points(depth, rgb, T, cls, obj, s.intrinsics, 4.0)
        for g in (gpu, ref):
            g.integrate(pts, cols, c, o, d)
    assert gpu.dropped_points() == 0 and gpu.label_overflows() == 0
    nlab = gpu.dump2()[5]
    occupied = nlab > 0
    assert nlab.max() > 7 and 0.005 < (nlab > 7).sum() / occupied.sum() < 0.05
    assert_state_equal(gpu, ref, PROB)


def test_probabilistic_label_overflow_is_counted(monkeypatch):
    """A label observation is dropped only past 254 pairs in one voxel or with the node pool exhausted - and then counted."""
    def one_voxel(n):
        g = gpu_grid(PROB, 0.1, max_blocks=1 << 8, max_points=1 << 12)
        g.integrate(np.zeros((n, 3), np.float32), np.zeros((n, 3), np.uint8), np.arange(n, dtype=np.int32), np.arange(n, dtype=np.int32))
        return g

    g = one_voxel(12)
    assert g.label_overflows() == 0 and g.dump2()[5].max() == 12
    v = g.get_voxels(1, 0.0)
    assert len(v.points) == 1 and v.object_ids[0] == 0  # first label keeps the arg-max (ties keep the incumbent)
    g = one_voxel(300)
    assert g.label_overflows() == 300 - 254 and g.dump2()[5].max() == 254
    monkeypatch.setenv("HV_PROB_NODE_CAP", "1")
    g = one_voxel(30)
    assert g.label_overflows() == 30 - 16 and g.dump2()[5].max() == 16  # 6 inline + the pool's only node


@pytest.mark.parametrize("kind", ALL_KINDS)
@pytest.mark.parametrize("do_carving", [False, True])
def test_pyslam_semantic_flow(kind, do_carving):
    """assign_object_ids_to_instance_ids -> remap_instance_ids -> integrate, 4 frames, per-frame instance ids
    permuted; new object ids may be permuted between the instances that need one in the same call."""
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import CameraFrustrum
    from pyslam_amd.volumetric_semantic import remap_instance_ids, set_next_object_id

    s = SyntheticRGBD(CFG, noise=True, invalid_frac=0.02)
    intr = s.intrinsics
    gpu, ref = gpu_grid(kind, CFG["voxel"]), RefSemGrid2(kind, CFG["voxel"])
    for g in (gpu, ref):
        g.set_depth_threshold(2.0)
        g.set_depth_decay_rate(0.07)
    set_next_object_id(1)
    ref.set_next_object_id(1)
    fr = CameraFrustrum(*intr, s.width, s.height, np.eye(4), depth_max=DEPTH_MAX, depth_min=DEPTH_MIN)
    obj_map = {}  # reference object id -> GPU object id
    for k, i in enumerate((0, 6, 12, 18)):
        depth, rgb, T, cls_img, inst_img = semantic_frame(s, i, shuffle=k)
        fr.set_T_cw(T)
        mg = gpu.assign_object_ids_to_instance_ids(fr, cls_img, inst_img, depth, depth_threshold=0.05, do_carving=do_carving,
                                                   min_vote_ratio=0.5, min_votes=3)
        mr = ref.assign_object_ids_to_instance_ids(fr.intr, s.width, s.height, T, fr.depth_max, fr.depth_min, cls_img, inst_img, depth,
                                                   0.05, do_carving, 0.5, 3)
        assert set(mg) == set(mr)
        new_g = sorted(v for v in mg.values() if v > 0 and v not in obj_map.values())
        new_r = sorted(v for v in mr.values() if v > 0 and v not in obj_map)
        assert len(new_g) == len(new_r)
        for inst in mr:  # pair up ids that are new in this call through the instance they were given to
            if mr[inst] > 0 and mr[inst] not in obj_map:
                obj_map[mr[inst]] = mg[inst]
        assert {k_: obj_map.get(v, v) for k_, v in mr.items()} == mg
        assert gpu._lib.hv_peek_next_object_id() == ref.peek_next_object_id()
        assert_state_equal(gpu, ref, kind, obj_map)
        og = remap_instance_ids(inst_img, mg, volume=gpu)
        orf = ref_remap_instance_ids(inst_img, mr)
        np.testing.assert_array_equal(og, relabel(orf, obj_map))
        pts, cols, cls, obj_g, depths = frame_points(depth, rgb, T, cls_img, og, intr, 4.0)
        obj_r = frame_points(depth, rgb, T, cls_img, orf, intr, 4.0)[3]
        gpu.integrate(pts, cols, cls, obj_g, depths)
        ref.integrate(pts, cols, cls, obj_r, depths)
        assert_state_equal(gpu, ref, kind, obj_map)
    assert gpu.label_overflows() == 0
    assert len(obj_map) >= 1  # objects get ids on their first re-observation (first sight: no votes -> -1)

    # get_object_segments: same objects, same point sets, same PCA boxes (as geometry)
    seg_g = gpu.get_object_segments(min_count=1, min_confidence=0.0)
    seg_r = ref.get_object_segments(1, 0.0)
    assert [o.object_id for o in seg_g.object_vector] == sorted(obj_map.get(o["object_id"], o["object_id"]) for o in seg_r)
    by_id = {obj_map.get(o["object_id"], o["object_id"]): o for o in seg_r}
    for og_ in seg_g.object_vector:
        o = by_id[og_.object_id]
        a = srt((og_.points, og_.colors))
        b = srt((o["points"], o["colors"]))
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
        if len(np.unique(cls_of(gpu, og_.object_id))) == 1:
            assert og_.class_id == o["class_id"]
        if kind in EXACT:
            assert (og_.confidence_min, og_.confidence_max) == (o["conf_min"], o["conf_max"])
        else:
            assert abs(og_.confidence_min - o["conf_min"]) < 2e-6 and abs(og_.confidence_max - o["conf_max"]) < 2e-6
        box = og_.oriented_bounding_box
        np.testing.assert_allclose(box.size, o["obb"][7:10], rtol=0, atol=1e-9)
        np.testing.assert_allclose(box.center, o["obb"][0:3], rtol=0, atol=1e-9)
        from pyslam_amd.volumetric_semantic import OrientedBoundingBox3D

        ref_box = OrientedBoundingBox3D(o["obb"][0:3], o["obb"][3:7], o["obb"][7:10])
        cg, cr = box.get_corners(), ref_box.get_corners()
        # same box as a point set (eigenvector signs are free)
        d = np.abs(cg[:, None, :] - cr[None, :, :]).sum(-1).min(1)
        assert d.max() < 1e-7

    # segment operations
    ids = sorted(o.object_id for o in seg_g.object_vector if o.object_id > 0)
    inv = {v: k_ for k_, v in obj_map.items()}
    if len(ids) >= 2:
        gpu.merge_segments(ids[0], ids[1])
        ref.merge_segments(inv.get(ids[0], ids[0]), inv.get(ids[1], ids[1]))
        assert_state_equal(gpu, ref, kind, obj_map)
    gpu.remove_segment(ids[0])
    ref.remove_segment(inv.get(ids[0], ids[0]))
    assert_state_equal(gpu, ref, kind, obj_map)
    cg_, og2 = gpu.get_ids()
    cr_, or2 = ref.get_ids()
    assert sorted(zip(cg_, og2)) == sorted(zip(cr_, relabel(or2, obj_map)))
    assert gpu.size() == len(cg_)
    gpu.remove_low_confidence_segments(1)
    ref.remove_low_confidence_segments(1)
    assert_state_equal(gpu, ref, kind, obj_map)


def cls_of(gpu, object_id):
    v = gpu.get_voxels(1, -1.0)
    return v.class_ids[v.object_ids == object_id]


def test_carve_and_reobserve_cycles_reuse_overflow_nodes():
    """a long session with carving on must not drain the overflow-node pool.  The same keyframe is fused with
    up to 24 distinct (object, class) pairs per voxel (> the 6 inline slots: chains of nodes), carved away completely (a depth image
    pushed back: every voxel in front of it is reset, voxel_grid_carving.h:47-79), and fused again - 50 times.  A reset voxel keeps
    its chain and grows its next map into it: the nodes handed out stop growing after the first cycle with the same labels and stay
    bounded when every tenth cycle draws new ones; nothing is dropped; the state is the compiled reference's, which frees and
    reallocates std::map nodes (voxel_data_semantic.h:249-672)."""
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import CameraFrustrum

    s = SyntheticRGBD(CFG, noise=False, invalid_frac=0.0)
    intr = s.intrinsics
    gpu, ref = gpu_grid(PROB, 0.05), RefSemGrid2(PROB, 0.05)
    depth, rgb, T, cls_img, inst_img = semantic_frame(s, 0)
    far = np.where(depth > 0, depth + 1.0, 0).astype(np.float32)
    fr = CameraFrustrum(*intr, s.width, s.height, T, depth_max=DEPTH_MAX, depth_min=DEPTH_MIN)

    def labels(seed):
        rng = np.random.default_rng(seed)
        return rng.integers(0, 6, cls_img.shape).astype(np.int32), rng.integers(0, 4, cls_img.shape).astype(np.int32)

    used = []
    for cycle in range(50):
        c_img, o_img = labels(5 if cycle % 10 else 100 + cycle)
        pts, cols, cls, obj, depths = frame_points(depth, rgb, T, c_img, o_img, intr, 4.0)
        for g in (gpu, ref):
            g.integrate(pts, cols, cls, obj, depths)
        used.append(gpu.prob_nodes_used())
        if cycle == 0:
            assert gpu.dump2(max_labels=32)[5].max() > 6 and used[0] > 0  # maps beyond the inline slots: chains exist
        if cycle in (0, 23, 49):
            assert_state_equal(gpu, ref, PROB)
        before = gpu.size()
        gpu.carve(fr, far, 0.01)
        ref.carve(fr.intr, s.width, s.height, T, fr.depth_max, fr.depth_min, far, 0.01)
        assert gpu.size() <= 0.02 * before  # (nearly) everything this keyframe put in is carved away again
    assert gpu.label_overflows() == 0 and gpu.dropped_points() == 0
    assert used[1:10] == [used[1]] * 9, used[:12]  # the same observation again: not one more node
    assert used[-1] <= 2 * used[0], (used[0], used[-1])  # new label draws lengthen some chains; 50 cycles do not multiply them
    assert_state_equal(gpu, ref, PROB)


@pytest.mark.parametrize("kind", ALL_KINDS)
def test_semantic_carve(kind):
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import CameraFrustrum

    s = SyntheticRGBD(CFG, noise=False, invalid_frac=0.02)
    intr = s.intrinsics
    gpu, ref = gpu_grid(kind, CFG["voxel"]), RefSemGrid2(kind, CFG["voxel"])
    depth, rgb, T, cls_img, inst_img = semantic_frame(s, 0)
    pts, cols, cls, obj, depths = frame_points(depth, rgb, T, cls_img, inst_img, intr, 4.0)
    for g in (gpu, ref):
        g.integrate(pts, cols, cls, obj, depths)
    # a later frame whose depth is pushed back: voxels now in front of the measured surface are carved
    depth2, _, T2, _, _ = semantic_frame(s, 9)
    depth2 = np.where(depth2 > 0, depth2 + 0.15, 0).astype(np.float32)
    fr = CameraFrustrum(*intr, s.width, s.height, T2, depth_max=DEPTH_MAX, depth_min=DEPTH_MIN)
    before = gpu.size()
    gpu.carve(fr, depth2, 0.01)
    ref.carve(fr.intr, s.width, s.height, T2, fr.depth_max, fr.depth_min, depth2, 0.01)
    assert gpu.size() < before
    assert_state_equal(gpu, ref, kind)
    gpu.remove_low_count_voxels(2)
    gpu.remove_low_confidence_voxels(0.75)
    v = gpu.get_voxels(1, -1.0)
    assert len(v.points) > 0 and (v.confidences >= 0.75).all()


def test_obb_pca_degenerate_cases():
    from pyslam_amd.volumetric_semantic import OrientedBoundingBox3D

    b = OrientedBoundingBox3D.compute_from_points(np.array([[1.0, 2.0, 3.0]]))
    assert np.allclose(b.center, [1, 2, 3]) and np.allclose(b.size, 0)
    b = OrientedBoundingBox3D.compute_from_points(np.array([[0.0, 0.0, 0.0], [2.0, 0.0, 0.0]]))
    assert np.allclose(b.center, [1, 0, 0]) and np.allclose(b.size, [2, 0, 0])
    rng = np.random.default_rng(3)
    P = rng.normal(size=(500, 3)) * np.array([3.0, 1.0, 0.2])
    b = OrientedBoundingBox3D.compute_from_points(P)
    assert b.size[0] > b.size[1] > b.size[2]
    C = b.get_corners()
    Rm = b.get_rotation_matrix()
    local = (P - b.center) @ Rm
    assert (np.abs(local) <= b.size / 2 + 1e-9).all()
    assert np.isclose(np.linalg.det(Rm), 1.0) and C.shape == (8, 3)


class _GpuKatGrid:
    def __init__(self, kind, voxel):
        self.g = gpu_grid(kind, voxel, max_blocks=1 << 8, max_points=1 << 12)

    def integrate(self, *a):
        self.g.integrate(*a)

    def get_voxels(self, mc, mconf):
        v = self.g.get_voxels(mc, mconf)
        return v.points, v.colors, v.class_ids, v.object_ids, v.confidences


def test_reference_kats_on_gpu_both_payloads():
    """cpp/test_volumetric_voxel_semantic.py replayed on the GPU grids."""
    from tests.semantic_kats import run_reference_kats

    run_reference_kats(_GpuKatGrid)


@pytest.mark.parametrize("kind,name", [(VOTE, "vote"), (PROB, "prob")])
def test_flow_matches_committed_golden(kind, name):
    """The pySLAM semantic flow on the GPU against the fixture generated from the compiled reference."""
    from pyslam_amd.volumetric_semantic import remap_instance_ids
    from tests.semantic_flow import FLOW_CFG, GpuAsSem2, run_flow
    from tests.test_semantic2_oracle import check_flow_against_golden

    g = gpu_grid(kind, FLOW_CFG["voxel"])
    r = run_flow(GpuAsSem2(g), lambda img, m: remap_instance_ids(img, m, volume=g), kind)
    check_flow_against_golden(r, name, 0.0 if kind == VOTE else 2e-6)


@pytest.mark.parametrize("kind", ALL_KINDS)
def test_semantic_queries_segments_by_class_and_integrate_segment(kind):
    """get_voxels_in_bb / get_voxels_in_camera_frustrum with semantics, get_class_segments, integrate_segment
    against the compiled reference."""
    from oracle.semantic import ref_get_class_segments, ref_get_voxels_in_bb, ref_get_voxels_in_frustum, ref_integrate_segment
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import BoundingBox3D, CameraFrustrum

    s = SyntheticRGBD(CFG, noise=True, invalid_frac=0.02)
    intr = s.intrinsics
    gpu, ref = gpu_grid(kind, CFG["voxel"]), RefSemGrid2(kind, CFG["voxel"])
    for i in (0, 6):
        depth, rgb, T, cls_img, inst_img = semantic_frame(s, i)
        pts, cols, cls, obj, depths = frame_points(depth, rgb, T, cls_img, inst_img, intr, 4.0)
        for g in (gpu, ref):
            g.integrate(pts, cols, cls, obj, depths)
    seg_pts = np.random.default_rng(4).uniform([2.0, 1.0, 0.5], [2.3, 1.3, 0.8], (5000, 3))
    seg_cols = np.full((5000, 3), 0.25, np.float32)
    gpu.integrate_segment(seg_pts, seg_cols, 77, 5)
    ref_integrate_segment(ref, seg_pts, seg_cols, 77, 5)
    gpu.integrate_segment(seg_pts, seg_cols, -1, 5)  # negative ids: ignored
    ref_integrate_segment(ref, seg_pts, seg_cols, -1, 5)
    assert_state_equal(gpu, ref, kind)
    tol = 0 if kind in EXACT else 2e-6

    def same(v, r):
        a = srt((v.points, v.colors, v.class_ids, v.object_ids, v.confidences))
        b = srt(r)
        assert len(a[0]) == len(b[0]) > 0
        for x, y in zip(a[:4], b[:4]):
            np.testing.assert_array_equal(x, y)
        np.testing.assert_allclose(a[4], b[4], rtol=0, atol=tol)

    bb = np.array([1.5, 0.5, 0.2, 4.0, 3.0, 1.8])
    same(gpu.get_voxels_in_bb(BoundingBox3D(bb[:3], bb[3:]), 2, 0.0, include_semantics=True), ref_get_voxels_in_bb(ref, bb, 2, 0.0))
    T = s[6][2]
    fr = CameraFrustrum(*intr, s.width, s.height, T, depth_max=3.0, depth_min=0.5)
    same(gpu.get_voxels_in_camera_frustrum(fr, 1, 0.0, include_semantics=True),
         ref_get_voxels_in_frustum(ref, fr.intr, s.width, s.height, T, fr.depth_max, fr.depth_min, 1, 0.0))
    assert len(gpu.get_voxels_in_bb(BoundingBox3D(bb[:3], bb[3:]), 2, 0.0).class_ids) == 0  # include_semantics=False
    ids, conf = ref_get_class_segments(ref, 1, 0.0)
    got = gpu.get_class_segments(1, 0.0)  # a ClassDataGroup (voxel_grid_data.h:131-139)
    assert list(got.class_ids) == [c.class_id for c in got.class_vector] and len(got) == len(got.class_vector)
    assert [(c.class_id, len(c.points)) for c in got] == [tuple(r) for r in ids.tolist()]
    np.testing.assert_allclose([[c.confidence_min, c.confidence_max] for c in got], conf, rtol=0, atol=tol)


@pytest.mark.parametrize("kind", ALL_KINDS)
def test_device_resident_keyframe_flow_equals_host_flow(kind):
    """filter_shadow_points -> assign_object_ids_to_instance_ids -> remap_instance_ids -> integrate_rgbd on torch CUDA tensors (one
    upload per image, what the semantic integrator does) against the same calls on numpy arrays: same id maps, id images and
    voxel states (object ids compared up to the permutation of ids created in the same call)."""
    import torch

    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import CameraFrustrum
    from pyslam_amd.volumetric_semantic import remap_instance_ids, set_next_object_id
    from tests.semantic_flow import canonical_ids

    s = SyntheticRGBD(CFG, noise=True, invalid_frac=0.02)
    results = []
    for device in (False, True):
        g = gpu_grid(kind, CFG["voxel"])
        g.set_depth_threshold(2.0)
        g.set_depth_decay_rate(0.07)
        fr = CameraFrustrum(*s.intrinsics, s.width, s.height, np.eye(4), depth_max=DEPTH_MAX, depth_min=DEPTH_MIN)
        set_next_object_id(1)
        maps, images = [], []
        for k, i in enumerate((0, 6, 12, 18)):
            depth, rgb, T, cls_img, inst_img = semantic_frame(s, i, shuffle=k)
            if device:
                depth, rgb, cls_img, inst_img = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (depth, rgb, cls_img, inst_img))
            d = g.filter_shadow_points(depth)
            assert bool(getattr(d, "is_cuda", False)) == device
            fr.set_T_cw(T)
            m = g.assign_object_ids_to_instance_ids(fr, cls_img, inst_img, d, depth_threshold=0.05, do_carving=(k == 2), min_vote_ratio=0.5,
                                                    min_votes=3)
            obj = remap_instance_ids(inst_img, m, volume=g)
            assert bool(getattr(obj, "is_cuda", False)) == device
            g.integrate_rgbd(d, rgb, *s.intrinsics, T, class_ids_image=cls_img, object_ids_image=obj, max_depth=4.0, use_depths=True)
            del depth, rgb, cls_img, d  # the launches may still be reading them: the allocator must not hand the blocks out yet
            maps.append(m)
            images.append(obj.cpu().numpy() if device else obj)
        results.append((maps, images, g.dump2()[:5]))
    (ma, ia, da), (mb, ib, db) = results
    assert any(v > 0 for m in ma for v in m.values())
    for x, y in zip(ma, mb):
        assert set(x) == set(y) and {k_ for k_ in x if x[k_] > 0} == {k_ for k_ in y if y[k_] > 0}
    for x, y in zip(ia, ib):
        np.testing.assert_array_equal(x > 0, y > 0)
        np.testing.assert_array_equal(np.where(x > 0, 0, x), np.where(y > 0, 0, y))
    (ka, inta, posa, cola, confa), (kb, intb, posb, colb, confb) = da, db
    np.testing.assert_array_equal(ka, kb)
    np.testing.assert_array_equal(inta[..., (0, 2, 3)], intb[..., (0, 2, 3)])
    np.testing.assert_array_equal(canonical_ids(inta[..., 1]), canonical_ids(intb[..., 1]))
    np.testing.assert_allclose(posa, posb, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(cola, colb, rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(confa, confb, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("kind", ALL_KINDS)
def test_fuse_keyframe_is_the_staged_calls_in_one(kind):
    """hv_semantic_fuse_keyframe (one call into the library per keyframe: what the integrator's device flow and the bench issue) against
    the staged calls it is made of - filter_shadow_points -> assign_object_ids_to_instance_ids -> remap_instance_ids -> integrate_rgbd -
    on the same keyframes, carving on in two of them, one keyframe without instance ids (the carve branch):
    the same voxel records (object ids up to the permutation of ids created in the same call)."""
    import torch

    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import CameraFrustrum
    from pyslam_amd.volumetric_semantic import remap_instance_ids, set_next_object_id
    from tests.semantic_flow import canonical_ids

    s = SyntheticRGBD(CFG, noise=True, invalid_frac=0.02)
    dumps = []
    for fused in (False, True):
        g = gpu_grid(kind, CFG["voxel"])
        g.set_depth_threshold(2.0)
        g.set_depth_decay_rate(0.07)
        fr = CameraFrustrum(*s.intrinsics, s.width, s.height, np.eye(4), depth_max=DEPTH_MAX, depth_min=DEPTH_MIN)
        set_next_object_id(1)
        for k, i in enumerate((0, 6, 12, 18, 24)):
            depth, rgb, T, cls_img, inst_img = semantic_frame(s, i, shuffle=k)
            depth, rgb, cls_img, inst_img = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (depth, rgb, cls_img, inst_img))
            carving = k == 2 or k == 3
            use_inst = k != 3  # keyframe 3: no instance ids -> the carve branch
            if fused:
                g.fuse_keyframe(fr, depth, rgb, cls_img, inst_img if use_inst else None, *s.intrinsics, T, filter_shadow_points=True,
                                use_instance_ids=use_inst, depth_threshold=0.05, do_carving=carving, min_vote_ratio=0.5, min_votes=3,
                                max_depth=4.0, use_depths=True)
            else:
                d = g.filter_shadow_points(depth)
                fr.set_T_cw(T)
                obj = None
                if use_inst:
                    m = g.assign_object_ids_to_instance_ids(fr, cls_img, inst_img, d, depth_threshold=0.05, do_carving=carving, min_vote_ratio=0.5,
                                                            min_votes=3)
                    obj = remap_instance_ids(inst_img, m, volume=g)
                elif carving:
                    g.carve(fr, d, 0.05)
                g.integrate_rgbd(d, rgb, *s.intrinsics, T, class_ids_image=cls_img, object_ids_image=obj, max_depth=4.0, use_depths=True)
            torch.cuda.synchronize()
        dumps.append(g.dump2()[:5])
    (ka, inta, posa, cola, confa), (kb, intb, posb, colb, confb) = dumps
    assert len(ka) > 100
    np.testing.assert_array_equal(ka, kb)
    np.testing.assert_array_equal(inta[..., (0, 2, 3)], intb[..., (0, 2, 3)])
    np.testing.assert_array_equal(canonical_ids(inta[..., 1]), canonical_ids(intb[..., 1]))
    np.testing.assert_array_equal(posa, posb)
    np.testing.assert_array_equal(cola, colb)
    np.testing.assert_array_equal(confa, confb)


@pytest.mark.parametrize("kind", ALL_KINDS)
@pytest.mark.parametrize("world", [2, 3])
def test_block_ownership_sharding_with_exchanged_votes_equals_the_single_grid(kind, world):
    """Semantic grids on `world` GPUs (here: `world` grids in one process play the ranks): hv_set_owner splits the blocks, every rank
    votes with its own voxels (assoc_vote), the pair lists are concatenated - what ShardedSemanticGrid all-gathers - and set on every
    rank, every rank decides.  All ranks must arrive at the single grid's map and object ids (the counter is process-wide here: the
    ranks' decisions are replayed from the same start value), and the union of their voxels must be the single grid's."""
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import CameraFrustrum
    from pyslam_amd.volumetric_semantic import get_next_object_id_peek, remap_instance_ids, set_next_object_id

    s = SyntheticRGBD(CFG, noise=True, invalid_frac=0.02)
    intr = s.intrinsics
    single = gpu_grid(kind, CFG["voxel"])
    ranks = [gpu_grid(kind, CFG["voxel"]) for _ in range(world)]
    for r, g in enumerate(ranks):
        g.set_owner(r, world)
    for g in [single] + ranks:
        g.set_depth_threshold(2.0)
        g.set_depth_decay_rate(0.07)
    fr = CameraFrustrum(*intr, s.width, s.height, np.eye(4), depth_max=DEPTH_MAX, depth_min=DEPTH_MIN)
    set_next_object_id(1)
    for k, i in enumerate((0, 6, 12, 18)):
        depth, rgb, T, cls_img, inst_img = semantic_frame(s, i, shuffle=k)
        fr.set_T_cw(T)
        start = get_next_object_id_peek()
        want = dict(single.assign_object_ids_to_instance_ids(fr, cls_img, inst_img, depth, depth_threshold=0.05, do_carving=(k == 2),
                                                             min_vote_ratio=0.5, min_votes=3))
        after = get_next_object_id_peek()
        for g in ranks:
            assert g.assoc_vote(fr, cls_img, inst_img, depth, 0.05, k == 2)
        lists = [g.assoc_pairs() for g in ranks]
        keys, counts = np.concatenate([x[0] for x in lists]), np.concatenate([x[1] for x in lists])
        maps = []
        for g in ranks:
            set_next_object_id(start)  # every rank's counter stands where the single grid's stood
            g.assoc_set_pairs(keys, counts)
            maps.append(dict(g.assoc_decide(0.5, 3)))
            assert get_next_object_id_peek() == after
        assert all(m == want for m in maps), (k, want, maps)
        obj_img = remap_instance_ids(inst_img, want, volume=single)
        pts, cols, cls, obj, depths = frame_points(depth, rgb, T, cls_img, obj_img, intr, 4.0)
        for g in [single] + ranks:
            g.integrate(pts, cols, cls, obj, depths)
    assert any(v > 0 for v in want.values())
    a = single.get_voxels(1, -1.0)
    rows = [g.get_voxels(1, -1.0) for g in ranks]
    assert sum(len(r.points) for r in rows) == len(a.points) and min(len(r.points) for r in rows) > 0.15 * len(a.points)
    got = srt(tuple(np.concatenate([getattr(r, f) for r in rows]) for f in ("points", "colors", "class_ids", "object_ids", "confidences")))
    exp = srt((a.points, a.colors, a.class_ids, a.object_ids, a.confidences))
    for x, y in zip(got, exp):
        np.testing.assert_array_equal(x, y)
    assert sum(g.num_blocks() for g in ranks) == single.num_blocks()


def test_remap_instance_ids_as_the_module_binds_it():
    """volumetric.remap_instance_ids (image_utils_module.h:49-94): ids absent from the map become -1, an EMPTY map hands the image back
    unchanged (the binding's early return; the C++ template behind it would set everything to -1) - host arrays and CUDA tensors, against
    the compiled reference through the same two early returns."""
    import torch

    from pyslam_amd.volumetric_semantic import remap_instance_ids

    rng = np.random.default_rng(8)
    img = rng.integers(-2, 9, (120, 160)).astype(np.int32)
    for m in ({}, {0: 0, 3: 41, 7: -1, 100: 5}):
        exp = ref_remap_instance_ids(img, m)
        np.testing.assert_array_equal(remap_instance_ids(img, m), exp)
        np.testing.assert_array_equal(remap_instance_ids(torch.from_numpy(img).cuda(), m).cpu().numpy(), exp)
    np.testing.assert_array_equal(ref_remap_instance_ids(img, {}), img)
    assert (ref_remap_instance_ids(img, {3: 41}) == np.where(img == 3, 41, -1)).all()
    # the narrower image types of the binding (int8 / uint8 / int16 / uint16): same lookup, results narrowed like the C++ assignment
    for dt in (np.int8, np.uint8, np.int16, np.uint16):
        small = rng.integers(0, 9, (40, 50)).astype(dt)
        m = {0: 0, 3: 300, 7: -1}
        exp = np.vectorize(lambda x: m.get(int(x), -1))(small).astype(np.int64).astype(dt)
        got = remap_instance_ids(small, m)
        assert got.dtype == dt
        np.testing.assert_array_equal(got, exp)
        assert remap_instance_ids(small, {}) is not None and (remap_instance_ids(small, {}) == small).all()
    with pytest.raises(RuntimeError, match="Unsupported instance id type"):
        remap_instance_ids(img.astype(np.float32), {1: 2})
