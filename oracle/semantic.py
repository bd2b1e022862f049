"""TEST INFRASTRUCTURE ONLY — ctypes front-ends of the semantic (voting) voxel-grid oracles:
PortSemGrid (oracle/semantic_oracle.c) and RefSemGrid (compiled reference, ref_sgrid_*)."""
import ctypes
import os

import numpy as np

from . import _ptr, port_lib, ref_lib

_c = ctypes
_vp, _i64, _i32, _f32 = _c.c_void_p, _c.c_int64, _c.c_int, _c.c_float
_bound = set()


def _bind(lib, prefix):
    if (id(lib), prefix) in _bound:
        return
    getattr(lib, prefix + "create").restype = _vp
    getattr(lib, prefix + "create").argtypes = [_f32, _i32]
    getattr(lib, prefix + "destroy").argtypes = [_vp]
    getattr(lib, prefix + "clear").argtypes = [_vp]
    getattr(lib, prefix + "num_blocks").restype = _i64
    getattr(lib, prefix + "num_blocks").argtypes = [_vp]
    getattr(lib, prefix + "set_depth_threshold").argtypes = [_f32]
    getattr(lib, prefix + "integrate").argtypes = [_vp, _vp, _i32, _i64, _vp, _i32, _vp, _vp, _vp]
    getattr(lib, prefix + "dump").restype = _i64
    getattr(lib, prefix + "dump").argtypes = [_vp, _vp, _vp, _vp, _vp]
    getattr(lib, prefix + "get_voxels").restype = _i64
    getattr(lib, prefix + "get_voxels").argtypes = [_vp, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _i64]
    _bound.add((id(lib), prefix))


class _SemBase:
    def __init__(self, lib, prefix, voxel_size, block_size):
        _bind(lib, prefix)
        self._lib, self._p = lib, prefix
        self.block_size = int(block_size)
        self._h = getattr(lib, prefix + "create")(float(np.float32(voxel_size)), self.block_size)

    def __del__(self):
        if getattr(self, "_h", None):
            getattr(self._lib, self._p + "destroy")(self._h)
            self._h = None

    def set_depth_threshold(self, t):
        getattr(self._lib, self._p + "set_depth_threshold")(float(t))

    def integrate(self, points, colors, class_ids=None, instance_ids=None, depths=None):
        points = np.ascontiguousarray(points)
        pos_kind = 1 if points.dtype == np.float64 else 0
        if pos_kind == 0:
            points = np.ascontiguousarray(points, dtype=np.float32)
        colors = np.ascontiguousarray(colors)
        color_kind = 1 if colors.dtype == np.uint8 else 2
        if color_kind == 2:
            colors = np.ascontiguousarray(colors, dtype=np.float32)
        cls = None if class_ids is None else np.ascontiguousarray(class_ids, dtype=np.int32)
        inst = None if instance_ids is None else np.ascontiguousarray(instance_ids, dtype=np.int32)
        dep = None if depths is None else np.ascontiguousarray(depths, dtype=np.float32)
        getattr(self._lib, self._p + "integrate")(self._h, _ptr(points), pos_kind, points.shape[0], _ptr(colors), color_kind,
                                                  _ptr(cls), _ptr(inst), _ptr(dep))

    def num_blocks(self):
        return getattr(self._lib, self._p + "num_blocks")(self._h)

    def clear(self):
        getattr(self._lib, self._p + "clear")(self._h)

    def dump(self):
        """-> keys [B,3], ints [B,bs^3,4] {count, object_id, class_id, counter}, pos_sums [B,bs^3,3] f64, col_sums f32."""
        nb, nv = self.num_blocks(), self.block_size ** 3
        keys = np.zeros((nb, 3), np.int32)
        ints = np.zeros((nb, nv, 4), np.int32)
        pos = np.zeros((nb, nv, 3), np.float64)
        col = np.zeros((nb, nv, 3), np.float32)
        getattr(self._lib, self._p + "dump")(self._h, _ptr(keys), _ptr(ints), _ptr(pos), _ptr(col))
        return keys, ints, pos, col

    def get_voxels(self, min_count=1, min_confidence=0.0):
        fn = getattr(self._lib, self._p + "get_voxels")
        n = fn(self._h, int(min_count), float(min_confidence), None, None, None, None, None, 0)
        pts = np.zeros((n, 3), np.float64)
        cols = np.zeros((n, 3), np.float32)
        cls = np.zeros(n, np.int32)
        obj = np.zeros(n, np.int32)
        conf = np.zeros(n, np.float32)
        if n:
            fn(self._h, int(min_count), float(min_confidence), _ptr(pts), _ptr(cols), _ptr(cls), _ptr(obj), _ptr(conf), n)
        return pts, cols, cls, obj, conf


class PortSemGrid(_SemBase):
    def __init__(self, voxel_size, block_size=8):
        super().__init__(port_lib(), "so_", voxel_size, block_size)


class RefSemGrid(_SemBase):
    def __init__(self, voxel_size, block_size=8):
        super().__init__(ref_lib(), "ref_sgrid_", voxel_size, block_size)
