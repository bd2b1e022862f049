"""The deterministic semantic scenario shared by tools/make_golden.py (run on the compiled reference), the CPU
port test and the GPU golden test: 4 labelled keyframes through pySLAM's per-frame semantic flow
(assign_object_ids_to_instance_ids -> remap_instance_ids -> integrate, with carving), then segments.
`grid` is any object with the _Sem2Base interface (oracle.semantic) — see GpuAsSem2 for the GPU adapter."""
import numpy as np

from pyslam_amd.synthetic import SyntheticRGBD
from tests.semantic_helpers import DEPTH_MAX, DEPTH_MIN, frame_points, semantic_frame

FLOW_CFG = dict(width=160, height=120, fx=131.25, fy=131.25, cx=79.5, cy=59.5, voxel=0.04)
FRAMES = (0, 6, 12, 18)


def run_flow(grid, remap, kind):
    """-> dict of results.  Object ids are canonicalised (relabelled in order of first appearance in the
    key-sorted dump) so that permutations of ids created in the same call do not matter."""
    s = SyntheticRGBD(FLOW_CFG, noise=True, invalid_frac=0.02)
    intr = np.array(s.intrinsics, np.float32)
    grid.set_depth_threshold(2.0)
    grid.set_depth_decay_rate(0.07)
    grid.set_next_object_id(1)
    maps = []
    for k, i in enumerate(FRAMES):
        depth, rgb, T, cls_img, inst_img = semantic_frame(s, i, shuffle=k)
        m = grid.assign_object_ids_to_instance_ids(intr, s.width, s.height, T, DEPTH_MAX, DEPTH_MIN, cls_img, inst_img, depth, 0.05, True,
                                                   0.5, 3)
        maps.append(m)
        obj_img = remap(inst_img, m)
        pts, cols, cls, obj, depths = frame_points(depth, rgb, T, cls_img, obj_img, s.intrinsics, 4.0)
        grid.integrate(pts, cols, cls, obj, depths)
    keys, ints, pos, col, conf = grid.dump()
    canon = canonical_ids(ints[..., 1])
    occ = ints[..., 0] > 0
    segs = grid.get_object_segments(1, 0.0)
    return dict(
        kind=kind, keys=keys, occ_block=np.nonzero(occ)[0].astype(np.int32), occ_voxel=np.nonzero(occ)[1].astype(np.int16),
        counts=ints[..., 0][occ], object_ids=canon[occ], class_ids=ints[..., 2][occ], counters=ints[..., 3][occ], conf=conf[occ],
        pos=pos[occ], col=col[occ], map_keys=np.array([sorted(m) for m in maps], dtype=object),
        map_valid=np.array([[int(m[k_] >= 0) for k_ in sorted(m)] for m in maps], dtype=object),
        next_object_id=grid.peek_next_object_id(), seg_sizes=np.array(sorted(len(o["points"]) for o in segs), np.int64),
        seg_box_sizes=np.array(sorted(tuple(np.round(o["obb"][7:10], 9)) for o in segs), np.float64).reshape(-1, 3))


def canonical_ids(obj):
    """Relabel positive object ids by order of first appearance (ids <= 0 keep their meaning)."""
    flat = obj.reshape(-1)
    out = flat.copy()
    first = {}
    for v in flat[flat > 0]:
        if int(v) not in first:
            first[int(v)] = len(first) + 1
    for a, b in first.items():
        out[flat == a] = b
    return out.reshape(obj.shape)


class GpuAsSem2:
    """Adapter giving a pyslam_amd semantic grid the oracle-style interface used by run_flow."""

    def __init__(self, grid):
        self.g = grid

    def set_depth_threshold(self, t):
        self.g.set_depth_threshold(t)

    def set_depth_decay_rate(self, r):
        self.g.set_depth_decay_rate(r)

    def set_next_object_id(self, v):
        from pyslam_amd.volumetric_semantic import set_next_object_id

        set_next_object_id(v)

    def peek_next_object_id(self):
        from pyslam_amd.volumetric_semantic import get_next_object_id_peek

        return get_next_object_id_peek()

    def assign_object_ids_to_instance_ids(self, intr, width, height, T, dmax, dmin, cls_img, inst_img, depth, thr, carve, ratio, votes):
        from pyslam_amd.volumetric import CameraFrustrum

        fr = CameraFrustrum(*intr, width, height, T, depth_max=dmax, depth_min=dmin)
        return self.g.assign_object_ids_to_instance_ids(fr, cls_img, inst_img, depth, thr, carve, ratio, votes)

    def integrate(self, *a):
        self.g.integrate(*a)

    def dump(self):
        keys, ints, pos, col, conf = self.g.dump2()[:5]
        return keys, ints, pos, col, conf

    def get_object_segments(self, min_count, min_conf):
        grp = self.g.get_object_segments(min_count, min_conf)
        return [dict(points=o.points, obb=np.concatenate([o.oriented_bounding_box.center, o.oriented_bounding_box.orientation,
                                                          o.oriented_bounding_box.size])) for o in grp.object_vector]
