"""`volumetric.VoxelBlockSemanticGrid` (voting payload) on the GPU: mirror of the semantic block grid
binding (cpp/volumetric/volumetric_grid_module.h:939-1033; payload voxel_data_semantic.h:106-202).

Provided: integrate(points, colors, class_ids, instance_ids, depths), get_voxels(min_count,
min_confidence) with class_ids / object_ids / confidences, set_depth_threshold, clear/reset, num_blocks.
Not provided (SURVEY 8a V17/V18 remainder): segment operations, instance->object association, the
probabilistic payload."""
import ctypes

import numpy as np

from . import _lib as L
from .volumetric import VoxelGridData, _Volume


class VoxelBlockSemanticGrid(_Volume):
    def __init__(self, voxel_size, block_size=8, device=0, max_blocks=None, max_points=None):
        voxel_size = float(np.float32(voxel_size))
        super().__init__(L.HV_MODE_VOXEL_SEMANTIC_GRID, voxel_size, 0.0, block_size, 1, device, max_blocks, max_points)
        self.voxel_size, self.block_size = voxel_size, int(block_size)

    def set_depth_threshold(self, depth_threshold):
        L.check(self._lib.hv_set_depth_threshold(self._h, float(depth_threshold)))

    def integrate(self, points, colors=None, class_ids=None, instance_ids=None, depths=None):
        pts = np.asarray(points)
        if pts.ndim != 2 or pts.shape[1] != 3:
            raise RuntimeError("points must be a contiguous Nx3 array")
        pdt = 1 if pts.dtype == np.float64 else 0
        pts = np.ascontiguousarray(pts, dtype=np.float64 if pdt else np.float32)
        n = pts.shape[0]
        if n == 0:
            return
        kind, cols = L.HV_COLOR_NONE, None
        if colors is not None:
            cols = np.ascontiguousarray(colors)
            if cols.ndim != 2 or cols.shape[1] != 3:
                raise RuntimeError("colors must be a contiguous Nx3 array")
            if cols.shape[0] != n:
                raise RuntimeError("points and colors must have the same size")
            if cols.dtype == np.uint8:
                kind = L.HV_COLOR_U8
            elif cols.dtype == np.float32:
                kind = L.HV_COLOR_F32
            else:
                raise RuntimeError(f"Colors must be uint8 or float32, got dtype with {cols.dtype}")

        def ids(a, name):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.int32)
            if a.ndim != 1 or a.shape[0] != n:
                raise RuntimeError(f"points and {name} must have the same size")
            return a

        cls, inst = ids(class_ids, "class_ids"), ids(instance_ids, "instance_ids")
        dep = None
        if depths is not None:
            dep = np.ascontiguousarray(depths, dtype=np.float32)
            if dep.ndim != 1 or dep.shape[0] != n:
                raise RuntimeError("points and depths must have the same size")
        if inst is not None and cls is None:
            raise RuntimeError("instance_ids but no class_ids is not supported")
        L.check(self._lib.hv_integrate_points_semantic(self._h, L.ptr(pts), pdt, n, L.ptr(cols), kind, L.ptr(cls), L.ptr(inst),
                                                       L.ptr(dep), L.HV_HOST))

    def get_voxels(self, min_count=1, min_confidence=0.0):
        n = ctypes.c_int64()
        L.check(self._lib.hv_get_voxels_semantic(self._h, int(min_count), float(min_confidence), None, None, None, None, None, 0,
                                                 ctypes.byref(n)))
        m = n.value
        out = VoxelGridData(np.zeros((m, 3), np.float64), np.zeros((m, 3), np.float32))
        out.class_ids, out.object_ids = np.zeros(m, np.int32), np.zeros(m, np.int32)
        out.confidences = np.zeros(m, np.float32)
        if m:
            L.check(self._lib.hv_get_voxels_semantic(self._h, int(min_count), float(min_confidence), L.ptr(out.points),
                                                     L.ptr(out.colors), L.ptr(out.class_ids), L.ptr(out.object_ids),
                                                     L.ptr(out.confidences), m, ctypes.byref(n)))
        return out

    def clear(self):
        L.check(self._lib.hv_reset(self._h))

    reset = clear

    def empty(self):
        return self.num_blocks() == 0

    def get_block_size(self):
        return self.block_size

    def dump(self):
        nb, nv = self.num_blocks(), self.block_size ** 3
        keys = np.zeros((nb, 3), np.int32)
        ints = np.zeros((nb, nv, 4), np.int32)
        pos = np.zeros((nb, nv, 3), np.float64)
        col = np.zeros((nb, nv, 3), np.float32)
        n = ctypes.c_int64()
        L.check(self._lib.hv_dump_blocks_semantic(self._h, L.ptr(keys), L.ptr(ints), L.ptr(pos), L.ptr(col), ctypes.byref(n)))
        return keys, ints, pos, col
