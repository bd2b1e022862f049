"""CPU: the C restatement of the full semantic block grids (oracle/semantic2_oracle.c: voting + probabilistic
payloads, association, carving, segments) against
(a) the reference's own known-answer tests for both payloads (cpp/test_volumetric_voxel_semantic.py);
(b) the committed golden fixtures generated from the compiled reference (tests/golden/semantic_flow_*.npz);
(c) the compiled reference itself (oracle/_ref), live, on random streams and on the pySLAM semantic flow."""
import os

import numpy as np
import pytest

import oracle
from oracle.semantic import (PortSemGrid2, RefSemGrid2, port_compute_obb_pca, port_remap_instance_ids,
                             ref_remap_instance_ids)
from tests.semantic_flow import FLOW_CFG, run_flow
from tests.semantic_kats import run_reference_kats
from tests.test_semantic_oracle import srt, stream

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
need_ref = pytest.mark.skipif(not oracle.ref_available(), reason="compiled reference not available")


@pytest.fixture(autouse=True)
def restore_statics():
    yield
    for mk in (PortSemGrid2,) + ((RefSemGrid2,) if oracle.ref_available() else ()):
        for kind in (0, 2):
            mk(kind, 0.05).set_depth_threshold(10.0)
        for kind in (1, 3):
            g = mk(kind, 0.05)
            g.set_depth_threshold(5.0)
            g.set_depth_decay_rate(0.07)


def test_reference_kats_on_the_port():
    run_reference_kats(lambda kind, voxel: PortSemGrid2(kind, voxel))


def test_hand_derived_answers_of_the_star2_payloads_on_the_port():
    """tests/semantic2_kats.py (derived by hand from voxel_data_semantic2.h; held on the compiled reference in
    tests/test_semantic2_payloads_reference.py) on the C restatement's kinds 2 and 3."""
    from tests.semantic2_kats import run_semantic2_kats

    run_semantic2_kats(lambda kind, voxel: PortSemGrid2(kind, voxel))


@need_ref
def test_reference_kats_on_the_compiled_reference():
    run_reference_kats(lambda kind, voxel: RefSemGrid2(kind, voxel))


def check_flow_against_golden(result, name, conf_atol):
    z = np.load(os.path.join(GOLD, f"semantic_flow_{name}.npz"), allow_pickle=False)
    for k in ("keys", "occ_block", "occ_voxel", "counts", "object_ids", "class_ids", "pos", "col", "seg_sizes"):
        np.testing.assert_array_equal(result[k], z[k], err_msg=k)
    assert [",".join(map(str, m)) for m in result["map_keys"]] == list(z["map_keys"])
    assert [",".join(map(str, m)) for m in result["map_valid"]] == list(z["map_valid"])
    assert result["next_object_id"] == int(z["next_object_id"])
    np.testing.assert_allclose(result["conf"], z["conf"], rtol=0, atol=conf_atol)
    assert np.abs(result["counters"] - z["counters"]).max() <= (0 if conf_atol == 0 else 1)
    np.testing.assert_allclose(result["seg_box_sizes"], z["seg_box_sizes"], rtol=0, atol=1e-8)


@pytest.mark.parametrize("kind,name", [(0, "vote"), (1, "prob")])
def test_port_flow_matches_golden(kind, name):
    r = run_flow(PortSemGrid2(kind, FLOW_CFG["voxel"]), port_remap_instance_ids, kind)
    check_flow_against_golden(r, name, 0.0)  # same libm as the fixture's generator: exact


@need_ref
@pytest.mark.parametrize("kind", [0, 1, 2, 3])  # 2, 3: the "*2" payloads of voxel_data_semantic2.h
@pytest.mark.parametrize("pos_dtype", [np.float32, np.float64])
@pytest.mark.parametrize("use_inst,use_depth", [(True, True), (True, False), (False, True), (False, False)])
def test_port_vs_reference_streams(kind, pos_dtype, use_inst, use_depth):
    a, b = PortSemGrid2(kind, 0.05), RefSemGrid2(kind, 0.05)
    for it in range(3):
        pts, cols, cls, inst, dep = stream(900 + it, 30000, pos_dtype)
        c = cols if it != 1 else (cols / 255.0).astype(np.float32)
        for g in (a, b):
            g.integrate(pts, c, cls, inst if use_inst else None, dep if use_depth else None)
    for x, y in zip(a.dump(), b.dump()):
        np.testing.assert_array_equal(x, y)
    for mc, mconf in ((1, 0.0), (2, 0.3), (3, 0.6)):
        for x, y in zip(srt(a.get_voxels(mc, mconf)), srt(b.get_voxels(mc, mconf))):
            np.testing.assert_array_equal(x, y)
    ca, oa = a.get_ids()
    cb, ob = b.get_ids()
    assert sorted(zip(ca, oa)) == sorted(zip(cb, ob))


@need_ref
@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_port_vs_reference_flow_and_segment_ops(kind):
    a, b = PortSemGrid2(kind, FLOW_CFG["voxel"]), RefSemGrid2(kind, FLOW_CFG["voxel"])
    ra = run_flow(a, port_remap_instance_ids, kind)
    rb = run_flow(b, ref_remap_instance_ids, kind)
    for k in ("keys", "occ_block", "occ_voxel", "counts", "object_ids", "class_ids", "counters", "conf", "pos", "col", "seg_sizes"):
        np.testing.assert_array_equal(ra[k], rb[k], err_msg=k)
    np.testing.assert_allclose(ra["seg_box_sizes"], rb["seg_box_sizes"], rtol=0, atol=1e-9)
    assert ra["next_object_id"] == rb["next_object_id"]
    # segment operations on the (id-canonicalised) state
    from tests.semantic_flow import canonical_ids

    def state(g):
        keys, ints, pos, col, conf = g.dump()
        return keys, ints[..., 0], canonical_ids(ints[..., 1]), ints[..., 2], pos

    ida = sorted(o["object_id"] for o in a.get_object_segments(1, 0.0) if o["object_id"] > 0)
    idb = sorted(o["object_id"] for o in b.get_object_segments(1, 0.0) if o["object_id"] > 0)
    assert len(ida) == len(idb) >= 1
    a.remove_low_confidence_segments(1)
    b.remove_low_confidence_segments(1)
    for x, y in zip(state(a), state(b)):
        np.testing.assert_array_equal(x, y)


@need_ref
def test_obb_pca_matches_reference_boxes():
    """compute_obb_pca on the object point sets of the flow: same centre / size as the compiled reference's
    OrientedBoundingBox3D::compute_from_points (axes up to sign)."""
    b = RefSemGrid2(0, FLOW_CFG["voxel"])
    run_flow(b, ref_remap_instance_ids, 0)
    segs = b.get_object_segments(0, 0.0)
    assert len(segs) >= 1
    for o in segs:
        obb = port_compute_obb_pca(o["points"])
        np.testing.assert_allclose(obb[0:3], o["obb"][0:3], rtol=0, atol=1e-9)
        np.testing.assert_allclose(obb[7:10], o["obb"][7:10], rtol=0, atol=1e-9)
    for n in (1, 2):
        pts = np.array([[0.5, 1.0, 2.0], [1.5, 1.0, 2.5]])[:n]
        obb = port_compute_obb_pca(pts)
        assert np.allclose(obb[0:3], pts.mean(0))


def test_label_map_sizes_the_reference_builds():
    """How many (object, class) pairs the reference's unbounded per-voxel std::map really holds (the port's map, bit-identical on
    every flow above): what the product's layout is sized by (hv_semantic.h: 6 pairs in the voxel record, 10-pair overflow nodes
    chained beyond them).  On the labelled synthetic stream the largest map has 4 pairs; uniform random label noise - the harshest
    model: every noisy pixel draws one of 40 x 30 pairs - pushes 1-2 % of the voxels past 7 pairs at a 5 % noise rate (rounds 1-3
    dropped those pairs; tests/test_gpu_semantic_ops.py::test_probabilistic_label_maps_grow_past_the_inline_slots_like_the_reference
    is the parity test of the chained maps)."""
    from pyslam_amd.synthetic import SyntheticRGBD
    from tests.semantic_helpers import CFG, frame_points, semantic_frame

    s = SyntheticRGBD(CFG, noise=True, invalid_frac=0.02)
    result = {}
    for p_noise in (0.0, 0.05):
        g = PortSemGrid2(1, 0.02)
        rng = np.random.default_rng(3)
        for k, i in enumerate(range(0, 60, 2)):
            depth, rgb, T, cls, inst = semantic_frame(s, i, shuffle=k)
            flip = rng.random(cls.shape) < p_noise
            cls = np.where(flip, rng.integers(0, 40, cls.shape), cls).astype(np.int32)
            obj = np.where(flip, rng.integers(0, 30, cls.shape), inst).astype(np.int32)
            pts, cols, c, o, d = frame_points(depth, rgb, T, cls, obj, s.intrinsics, 4.0)
            g.integrate(pts, cols, c, o, d)
        most, hist = g.label_histogram()
        result[p_noise] = (most, hist)
    most, hist = result[0.0]
    assert 1 <= most <= 4 and hist[5:].sum() == 0
    most, hist = result[0.05]
    over = hist[8:].sum() / hist.sum()
    assert most > 7 and 0.005 < over < 0.05  # 1-2 % of the occupied voxels need an overflow node


@need_ref
def test_remap_instance_ids_port_vs_reference_including_the_empty_map():
    rng = np.random.default_rng(8)
    img = rng.integers(-2, 9, (60, 80)).astype(np.int32)
    for m in ({}, {0: 0, 3: 41, 7: -1, 100: 5}):
        np.testing.assert_array_equal(port_remap_instance_ids(img, m), ref_remap_instance_ids(img, m))
    np.testing.assert_array_equal(ref_remap_instance_ids(img, {}), img)  # the binding's early return (image_utils_module.h:52-58)
