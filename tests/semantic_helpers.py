"""Shared scene / flow helpers of the semantic-grid tests: the per-frame flow of pySLAM's
VolumetricIntegratorVoxelSemanticGrid (pyslam/dense/volumetric_integrator_voxel_semantic_grid.py:340-461)
driven identically on the GPU grid and on an oracle grid."""
import numpy as np

from oracle import host_prep
from pyslam_amd.synthetic import SyntheticRGBD

CFG = dict(width=320, height=240, fx=262.5, fy=262.5, cx=159.5, cy=119.5, voxel=0.02)
DEPTH_MAX, DEPTH_MIN = 6.0, 0.1


def semantic_frame(s, i, shuffle=0):
    """depth f32, rgb u8, T_cw, class image i32, instance image i32 of synthetic frame i.  Classes = the scene's
    surface labels; instances: spheres / box get per-frame ids (permuted by `shuffle`), room faces are stuff (0);
    a band of pixels carries invalid (-1) labels."""
    depth, rgb, T = s[i]
    lab = s.labels(i).astype(np.int32)
    cls = lab.copy()
    inst = np.zeros_like(lab)
    things = [10, 11, 20]
    for k, t in enumerate(things):
        inst[lab == t] = 1 + (k + shuffle) % len(things)
    cls[:, :3] = -1
    inst[:3, :] = -1
    return depth, rgb, T, cls, inst


def frame_points(depth, rgb, T, cls_img, obj_img, intr, max_depth):
    """depth2pointcloud(..., semantic_image, object_ids_image) + world transform (…voxel_semantic_grid.py:402-436)."""
    pts_c, cols, valid = host_prep.depth2pointcloud(depth, rgb, *intr, max_depth)
    pw = host_prep.world_points(pts_c, T)
    depths = np.ascontiguousarray(pts_c[:, 2], dtype=np.float32)
    return (np.ascontiguousarray(pw, dtype=np.float64), np.ascontiguousarray(cols, dtype=np.float32),
            np.ascontiguousarray(cls_img[valid], dtype=np.int32), np.ascontiguousarray(obj_img[valid], dtype=np.int32), depths)


def relabel(values, mapping):
    out = np.array(values, copy=True)
    for a, b in mapping.items():
        out[np.asarray(values) == a] = b
    return out


def sorted_rows(v):
    """(points, colours, class ids, object ids, confidences) sorted by position (the reference's row order is its hash-map order)."""
    i = np.lexsort(np.asarray(v[0]).T[::-1])
    return tuple(np.asarray(a)[i] for a in v)


def compare_keyframe_flow(kind, config, voxel, frame_ids, max_blocks, max_points, depth_threshold=0.03, do_carving=False,
                          frustum_depth=(8.0, 0.01), full_dump=False, device_images=True, gpu=None):
    """pySLAM's per-keyframe semantic flow (volumetric_integrator_voxel_semantic_grid.py:326-461: shadow filter ->
    assign_object_ids_to_instance_ids -> remap_instance_ids -> depth2pointcloud with labels + world transform -> integrate) on the HIP
    grid — device-resident images, the fused integrate_rgbd: what the integrator and tools/bench_semantic.py run — against the COMPILED
    REFERENCE (oracle.semantic.RefSemGrid2 + the numpy host prep pinned to the reference's depth.py) on the same keyframes, at any
    configuration.  Asserts after every keyframe: filtered depth bit-equal, id maps equal (new object ids up to the permutation
    between instances that need one in the same call, voxel_semantic_data_association.h:284-361), remapped id images equal; after
    the last: same block count, every occupied voxel's averaged position / colour / class / object id equal, confidences exact
    (voting payloads) or <= 2e-6 (log-probability payloads), the count distribution through get_voxels(min_count = 2, 3, 5), segments per object; with
    full_dump also every voxel record (counts, sums, counters).  -> dict of sizes (for the bench line)."""
    import torch

    from oracle import host_prep as hp
    from oracle.semantic import RefSemGrid2, ref_remap_instance_ids
    from pyslam_amd.volumetric import CameraFrustrum
    from pyslam_amd import volumetric_semantic as vs
    from pyslam_amd.volumetric_semantic import remap_instance_ids, set_next_object_id

    s = SyntheticRGBD(config)
    intr = s.intrinsics
    intr32 = np.array(intr, np.float32)
    if gpu is None:
        # kind 0 voting, 1 probabilistic, 2 / 3 the "*2" payloads of voxel_data_semantic2.h (RefSemGrid2's kinds)
        gpu = (vs.VoxelBlockSemanticGrid, vs.VoxelBlockSemanticProbabilisticGrid, vs.VoxelBlockSemanticGrid2,
               vs.VoxelBlockSemanticProbabilisticGrid2)[kind](voxel, 8, max_blocks=max_blocks, max_points=max_points)
    ref = RefSemGrid2(kind, voxel, 8)
    for g in (gpu, ref):
        g.set_depth_threshold(5.0)
        g.set_depth_decay_rate(0.07)
    set_next_object_id(1)
    ref.set_next_object_id(1)
    fr = CameraFrustrum(*intr, s.width, s.height, np.eye(4), depth_max=frustum_depth[0], depth_min=frustum_depth[1])
    obj_map = {}  # reference object id -> GPU object id
    n_new = 0
    ref_seconds = []  # the reference side of every keyframe on this host's clock (bench: cpu_reference)
    import time

    for k, i in enumerate(frame_ids):
        depth, rgb, T, cls_img, inst_img = semantic_frame(s, i, shuffle=k)
        fr.set_T_cw(T)
        if device_images:
            dev = torch.device("cuda", gpu._cfg.device)
            d_g = gpu.filter_shadow_points(torch.from_numpy(depth).to(dev))
            c_g, cl_g, in_g = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (rgb, cls_img, inst_img))
        else:
            d_g, c_g, cl_g, in_g = gpu.filter_shadow_points(depth), rgb, cls_img, inst_img
        t0 = time.perf_counter()
        d_r = hp.filter_shadow_points(depth)
        t_ref = time.perf_counter() - t0
        np.testing.assert_array_equal(d_g.cpu().numpy() if device_images else d_g, d_r)
        mg = gpu.assign_object_ids_to_instance_ids(fr, cl_g, in_g, d_g, depth_threshold=depth_threshold, do_carving=do_carving,
                                                   min_vote_ratio=0.5, min_votes=3)
        t0 = time.perf_counter()
        mr = ref.assign_object_ids_to_instance_ids(intr32, s.width, s.height, T, fr.depth_max, fr.depth_min, cls_img, inst_img, d_r,
                                                   depth_threshold, do_carving, 0.5, 3)
        t_ref += time.perf_counter() - t0
        assert set(mg) == set(mr), (k, mg, mr)
        new_g = sorted(v for v in mg.values() if v > 0 and v not in obj_map.values())
        new_r = sorted(v for v in mr.values() if v > 0 and v not in obj_map)
        assert len(new_g) == len(new_r), (k, mg, mr)
        for inst in mr:
            if mr[inst] > 0 and mr[inst] not in obj_map:
                obj_map[mr[inst]] = mg[inst]
                n_new += 1
        assert {k_: obj_map.get(v, v) for k_, v in mr.items()} == mg, (k, mg, mr, obj_map)
        assert gpu._lib.hv_peek_next_object_id() == ref.peek_next_object_id()
        og = remap_instance_ids(in_g, mg, volume=gpu)
        t0 = time.perf_counter()
        orf = ref_remap_instance_ids(inst_img, mr)
        t_ref += time.perf_counter() - t0
        np.testing.assert_array_equal(og.cpu().numpy() if device_images else og, relabel(orf, obj_map))
        gpu.integrate_rgbd(d_g, c_g, *intr, T, class_ids_image=cl_g, object_ids_image=og, max_depth=4.0, use_depths=True)
        t0 = time.perf_counter()
        pts, cols, cls, ob, depths = frame_points(d_r, rgb, T, cls_img, orf, intr, 4.0)
        ref.integrate(pts.astype(np.float32), cols, cls, ob, depths)
        ref_seconds.append(t_ref + time.perf_counter() - t0)
    assert gpu.dropped_points() == 0 and gpu.label_overflows() == 0
    assert gpu.num_blocks() == ref.num_blocks()
    tol = 0.0 if kind in (0, 2) else 2e-6  # integer label state: exact; log-probability payloads: confidences within 2e-6
    vg = gpu.get_voxels(1, -1.0)
    a = sorted_rows((vg.points, vg.colors, vg.class_ids, vg.object_ids, vg.confidences))
    b = sorted_rows(ref.get_voxels(1, -1.0))
    assert len(a[0]) == len(b[0]) > 0
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_array_equal(a[3], relabel(b[3], obj_map))
    np.testing.assert_allclose(a[4], b[4], rtol=0, atol=tol)
    for mc in (2, 3, 5):  # the count distribution (rows carry sum / count, not the count)
        assert len(gpu.get_voxels(mc, -1.0).points) == len(ref.get_voxels(mc, -1.0)[0])
    seg_g = gpu.get_object_segments(min_count=1, min_confidence=0.0)
    seg_r = ref.get_object_segments(1, 0.0)
    assert [(o.object_id, len(o.points)) for o in seg_g.object_vector] == sorted(
        (obj_map.get(o["object_id"], o["object_id"]), len(o["points"])) for o in seg_r)
    if full_dump:
        kg, ig, pg, cg, confg = gpu.dump2()[:5]
        kr, ir, pr, cr, confr = ref.dump()
        np.testing.assert_array_equal(kg, kr)
        ir = ir.copy()
        ir[..., 1] = relabel(ir[..., 1], obj_map)
        np.testing.assert_array_equal(ig[..., :3], ir[..., :3])
        np.testing.assert_array_equal(pg, pr)
        np.testing.assert_array_equal(cg, cr)
        if kind in (0, 2):
            np.testing.assert_array_equal(ig[..., 3], ir[..., 3])
            np.testing.assert_array_equal(confg, confr)
        else:
            np.testing.assert_allclose(confg, confr, rtol=0, atol=tol)
    return dict(keyframes=len(frame_ids), blocks=int(gpu.num_blocks()), occupied_voxels=int(len(a[0])), objects=len(seg_r),
                new_object_ids=n_new, map_sizes=[len(obj_map)], conf_max_abs_diff=float(np.abs(a[4] - b[4]).max()),
                label_overflows=int(gpu.label_overflows()), ref_seconds=ref_seconds)
