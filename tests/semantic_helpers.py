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
