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


# ---------------------------------------------------------------------------------------------------
# Full semantic block grids of the compiled reference (oracle/ref_shim_semantic.cpp): kind 0 =
# VoxelBlockSemanticGrid (voting), kind 1 = VoxelBlockSemanticProbabilisticGrid.
# ---------------------------------------------------------------------------------------------------
_f64 = _c.c_double
_bound2 = set()


def _bind2(lib, p):
    if (id(lib), p) in _bound2:
        return
    g = lambda n: getattr(lib, p + n)  # noqa: E731
    g("create").restype = _vp
    g("create").argtypes = [_i32, _f64, _i32]
    g("destroy").argtypes = [_vp]
    g("clear").argtypes = [_vp]
    g("num_blocks").restype = _i64
    g("num_blocks").argtypes = [_vp]
    g("set_depth_threshold").argtypes = [_vp, _f32]
    g("set_depth_decay_rate").argtypes = [_vp, _f32]
    g("peek_next_object_id").restype = _i32
    g("peek_next_object_id").argtypes = []
    g("set_next_object_id").argtypes = [_i32]
    g("integrate").argtypes = [_vp, _vp, _i32, _i64, _vp, _i32, _vp, _vp, _vp]
    g("dump").restype = _i64
    g("dump").argtypes = [_vp, _vp, _vp, _vp, _vp, _vp]
    g("get_voxels").restype = _i64
    g("get_voxels").argtypes = [_vp, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _i64]
    g("assign_object_ids").restype = _i64
    g("assign_object_ids").argtypes = [_vp, _vp, _i32, _i32, _vp, _f32, _f32, _vp, _vp, _vp, _f32, _i32, _f32, _i32, _vp, _vp, _i64]
    g("carve").argtypes = [_vp, _vp, _i32, _i32, _vp, _f32, _f32, _vp, _f32]
    g("get_object_segments").restype = _i64
    g("get_object_segments").argtypes = [_vp, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _c.POINTER(_i64)]
    g("merge_segments").argtypes = [_vp, _i32, _i32]
    g("remove_segment").argtypes = [_vp, _i32]
    g("remove_low_confidence_segments").argtypes = [_vp, _i32]
    g("get_ids").restype = _i64
    g("get_ids").argtypes = [_vp, _vp, _vp, _i64]
    _bound2.add((id(lib), p))


class _Sem2Base:
    """Shared front-end of the full semantic grids (compiled reference or C restatement)."""

    def __init__(self, lib, prefix, kind, voxel_size, block_size):
        _bind2(lib, prefix)
        self._lib, self._p, self.kind = lib, prefix, int(kind)
        self.block_size = int(block_size)
        self._h = getattr(lib, prefix + "create")(self.kind, float(voxel_size), self.block_size)

    def _f(self, name):
        return getattr(self._lib, self._p + name)

    def __del__(self):
        if getattr(self, "_h", None):
            self._f("destroy")(self._h)
            self._h = None

    def set_depth_threshold(self, t):
        self._f("set_depth_threshold")(self._h, float(t))

    def set_depth_decay_rate(self, r):
        self._f("set_depth_decay_rate")(self._h, float(r))

    def peek_next_object_id(self):
        return self._f("peek_next_object_id")()

    def set_next_object_id(self, v):
        self._f("set_next_object_id")(int(v))

    def clear(self):
        self._f("clear")(self._h)

    def num_blocks(self):
        return self._f("num_blocks")(self._h)

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
        self._f("integrate")(self._h, _ptr(points), pos_kind, points.shape[0], _ptr(colors), color_kind, _ptr(cls), _ptr(inst), _ptr(dep))

    def dump(self):
        """-> keys [B,3], ints [B,bs^3,4] {count, object_id, class_id, counter}, pos_sums f64, col_sums f32, conf [B,bs^3] f32."""
        nb, nv = self.num_blocks(), self.block_size ** 3
        keys = np.zeros((nb, 3), np.int32)
        ints = np.zeros((nb, nv, 4), np.int32)
        conf = np.zeros((nb, nv), np.float32)
        pos = np.zeros((nb, nv, 3), np.float64)
        col = np.zeros((nb, nv, 3), np.float32)
        self._f("dump")(self._h, _ptr(keys), _ptr(ints), _ptr(conf), _ptr(pos), _ptr(col))
        return keys, ints, pos, col, conf

    def get_voxels(self, min_count=1, min_confidence=0.0):
        fn = self._f("get_voxels")
        n = fn(self._h, int(min_count), float(min_confidence), None, None, None, None, None, 0)
        pts = np.zeros((n, 3), np.float64)
        cols = np.zeros((n, 3), np.float32)
        cls = np.zeros(n, np.int32)
        obj = np.zeros(n, np.int32)
        conf = np.zeros(n, np.float32)
        if n:
            fn(self._h, int(min_count), float(min_confidence), _ptr(pts), _ptr(cols), _ptr(cls), _ptr(obj), _ptr(conf), n)
        return pts, cols, cls, obj, conf

    def assign_object_ids_to_instance_ids(self, intr, width, height, T_cw, depth_max, depth_min, class_img, inst_img, depth=None,
                                          depth_threshold=0.1, do_carving=False, min_vote_ratio=0.5, min_votes=3):
        intr = np.ascontiguousarray(intr, dtype=np.float32)
        T = np.ascontiguousarray(T_cw, dtype=np.float64)
        cls = np.ascontiguousarray(class_img, dtype=np.int32)
        inst = np.ascontiguousarray(inst_img, dtype=np.int32)
        dep = None if depth is None else np.ascontiguousarray(depth, dtype=np.float32)
        cap = 1 << 16
        mi, mo = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        n = self._f("assign_object_ids")(self._h, _ptr(intr), width, height, _ptr(T), depth_max, depth_min, _ptr(cls), _ptr(inst), _ptr(dep),
                                         float(depth_threshold), int(bool(do_carving)), float(min_vote_ratio), int(min_votes), _ptr(mi),
                                         _ptr(mo), cap)
        return {int(k): int(v) for k, v in zip(mi[:n], mo[:n])}

    def carve(self, intr, width, height, T_cw, depth_max, depth_min, depth, depth_threshold):
        intr = np.ascontiguousarray(intr, dtype=np.float32)
        T = np.ascontiguousarray(T_cw, dtype=np.float64)
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        self._f("carve")(self._h, _ptr(intr), width, height, _ptr(T), depth_max, depth_min, _ptr(depth), depth_threshold)

    def get_object_segments(self, min_count=1, min_confidence=0.0):
        """-> list of dicts (ascending object id): object_id, class_id, points, colors, conf_min, conf_max, obb[10]."""
        fn = self._f("get_object_segments")
        npts = _i64(0)
        no = fn(self._h, int(min_count), float(min_confidence), None, None, None, None, None, 0, 0, _c.byref(npts))
        ids = np.zeros((no, 3), np.int32)
        conf = np.zeros((no, 2), np.float32)
        obb = np.zeros((no, 10), np.float64)
        pts = np.zeros((npts.value, 3), np.float64)
        cols = np.zeros((npts.value, 3), np.float32)
        if no:
            fn(self._h, int(min_count), float(min_confidence), _ptr(ids), _ptr(conf), _ptr(obb), _ptr(pts), _ptr(cols), no, npts.value,
               _c.byref(npts))
        out, at = [], 0
        for o in range(no):
            k = int(ids[o, 2])
            out.append(dict(object_id=int(ids[o, 0]), class_id=int(ids[o, 1]), points=pts[at:at + k], colors=cols[at:at + k],
                            conf_min=float(conf[o, 0]), conf_max=float(conf[o, 1]), obb=obb[o].copy()))
            at += k
        return out

    def merge_segments(self, id1, id2):
        self._f("merge_segments")(self._h, int(id1), int(id2))

    def remove_segment(self, object_id):
        self._f("remove_segment")(self._h, int(object_id))

    def remove_low_confidence_segments(self, min_confidence):
        self._f("remove_low_confidence_segments")(self._h, int(min_confidence))

    def get_ids(self):
        fn = self._f("get_ids")
        n = fn(self._h, None, None, 0)
        cls, obj = np.zeros(n, np.int32), np.zeros(n, np.int32)
        if n:
            fn(self._h, _ptr(cls), _ptr(obj), n)
        return cls, obj


class RefSemGrid2(_Sem2Base):
    """The compiled reference's VoxelBlockSemanticGrid (kind 0) / VoxelBlockSemanticProbabilisticGrid (kind 1) /
    VoxelBlockSemanticGrid2 (kind 2: separate object and class counters) / VoxelBlockSemanticProbabilisticGrid2 (kind 3: marginal
    label maps) - voxel_block_semantic_grid.h:118-123."""

    def __init__(self, kind, voxel_size, block_size=8):
        super().__init__(ref_lib(), "ref_sem2_", kind, voxel_size, block_size)

    def dump_marginals(self):
        """-> (object confidence, class confidence) [B,bs^3] f32 in dump()'s order: get_object_confidence / get_class_confidence of the
        two *2 payloads (voxel_data_semantic2.h:60-76, 528-560); -1 everywhere for kinds 0 and 1."""
        fn = self._lib.ref_sem2_dump_marginals
        fn.restype = _i64
        fn.argtypes = [_vp, _vp, _vp]
        nb, nv = self.num_blocks(), self.block_size ** 3
        oc, cc = np.zeros((nb, nv), np.float32), np.zeros((nb, nv), np.float32)
        fn(self._h, _ptr(oc), _ptr(cc))
        return oc, cc


class PortSemGrid2(_Sem2Base):
    """oracle/semantic2_oracle.c: the C restatement of the same two grids."""

    def __init__(self, kind, voxel_size, block_size=8):
        super().__init__(port_lib(), "so2_", kind, voxel_size, block_size)

    def label_histogram(self, cap=33):
        """-> (largest label map of an occupied voxel, hist[k] = occupied voxels holding k (object, class) pairs)."""
        fn = self._lib.so2_label_histogram
        fn.restype = _i32
        fn.argtypes = [_vp, _vp, _i32]
        hist = np.zeros(cap, np.int64)
        most = fn(self._h, _ptr(hist), cap)
        return int(most), hist


def port_remap_instance_ids(inst_img, mapping):
    lib = port_lib()
    lib.so2_remap_instance_ids.argtypes = [_vp, _i32, _i32, _vp, _vp, _i64, _vp]
    img = np.ascontiguousarray(inst_img, dtype=np.int32)
    keys = np.fromiter(mapping.keys(), np.int32, len(mapping))
    vals = np.fromiter(mapping.values(), np.int32, len(mapping))
    out = np.empty_like(img)
    lib.so2_remap_instance_ids(_ptr(img), img.shape[0], img.shape[1], _ptr(keys), _ptr(vals), len(keys), _ptr(out))
    return out


def port_compute_obb_pca(points):
    lib = port_lib()
    lib.so2_compute_obb_pca.argtypes = [_vp, _i64, _vp]
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    obb = np.zeros(10, np.float64)
    lib.so2_compute_obb_pca(_ptr(pts), pts.shape[0], _ptr(obb))
    return obb


def ref_remap_instance_ids(inst_img, mapping):
    lib = ref_lib()
    lib.ref_remap_instance_ids.argtypes = [_vp, _i32, _i32, _vp, _vp, _i64, _vp]
    img = np.ascontiguousarray(inst_img, dtype=np.int32)
    keys = np.fromiter(mapping.keys(), np.int32, len(mapping))
    vals = np.fromiter(mapping.values(), np.int32, len(mapping))
    out = np.empty_like(img)
    lib.ref_remap_instance_ids(_ptr(img), img.shape[0], img.shape[1], _ptr(keys), _ptr(vals), len(keys), _ptr(out))
    return out


def _bind_queries(lib):
    lib.ref_sem2_get_voxels_in_bb.restype = _i64
    lib.ref_sem2_get_voxels_in_bb.argtypes = [_vp, _vp, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _i64]
    lib.ref_sem2_get_voxels_in_frustum.restype = _i64
    lib.ref_sem2_get_voxels_in_frustum.argtypes = [_vp, _vp, _i32, _i32, _vp, _f32, _f32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _i64]
    lib.ref_sem2_integrate_segment.argtypes = [_vp, _vp, _i64, _vp, _i32, _i32]
    lib.ref_sem2_get_class_segments.restype = _i64
    lib.ref_sem2_get_class_segments.argtypes = [_vp, _i32, _f32, _vp, _vp, _i64]


def _rows(fn):
    n = fn(None, None, None, None, None, 0)
    pts, cols = np.zeros((n, 3), np.float64), np.zeros((n, 3), np.float32)
    cls, obj, conf = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.float32)
    if n:
        fn(_ptr(pts), _ptr(cols), _ptr(cls), _ptr(obj), _ptr(conf), n)
    return pts, cols, cls, obj, conf


def ref_get_voxels_in_bb(grid, bb, min_count=1, min_confidence=0.0):
    """RefSemGrid2.get_voxels_in_bb<IncludeSemantics=true>."""
    lib = ref_lib()
    _bind_queries(lib)
    bb = np.ascontiguousarray(bb, dtype=np.float64)
    return _rows(lambda *a: lib.ref_sem2_get_voxels_in_bb(grid._h, _ptr(bb), int(min_count), float(min_confidence), *a))


def ref_get_voxels_in_frustum(grid, intr, width, height, T_cw, depth_max, depth_min, min_count=1, min_confidence=0.0):
    lib = ref_lib()
    _bind_queries(lib)
    intr = np.ascontiguousarray(intr, dtype=np.float32)
    T = np.ascontiguousarray(T_cw, dtype=np.float64)
    return _rows(lambda *a: lib.ref_sem2_get_voxels_in_frustum(grid._h, _ptr(intr), width, height, _ptr(T), depth_max, depth_min,
                                                               int(min_count), float(min_confidence), *a))


def ref_integrate_segment(grid, points, colors, object_id, class_id):
    lib = ref_lib()
    _bind_queries(lib)
    pts = np.ascontiguousarray(points, dtype=np.float64)
    cols = np.ascontiguousarray(colors, dtype=np.float32)
    lib.ref_sem2_integrate_segment(grid._h, _ptr(pts), pts.shape[0], _ptr(cols), int(object_id), int(class_id))


def ref_get_class_segments(grid, min_count=1, min_confidence=0.0):
    """-> ids [C,2] {class_id, n_points}, conf [C,2] {min, max}, ascending class id."""
    lib = ref_lib()
    _bind_queries(lib)
    n = lib.ref_sem2_get_class_segments(grid._h, int(min_count), float(min_confidence), None, None, 0)
    ids, conf = np.zeros((n, 2), np.int32), np.zeros((n, 2), np.float32)
    if n:
        lib.ref_sem2_get_class_segments(grid._h, int(min_count), float(min_confidence), _ptr(ids), _ptr(conf), n)
    return ids, conf
