"""TEST INFRASTRUCTURE ONLY — ctypes front-ends of the CPU oracles.

Import rules (enforced by tests/test_abi.py::test_product_never_touches_the_oracle): only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package; the
product package ``pyslam_amd`` never does.

* :class:`PortGrid`   — oracle/voxel_oracle.c, restatement of cpp/volumetric VoxelBlockGrid (pinned
  against the compiled reference).
* :class:`RefGrid`    — oracle/_ref/libref_volumetric.so, the unmodified reference sources
  (available wherever the prebuilt .so travelled or /root/reference is present to build it).
* :class:`PortTsdf`   — oracle/tsdf_oracle.c, restatement of Open3D ScalableTSDFVolume semantics
  (PARITY UNPINNED: Open3D is an absent third-party dependency of the reference).
* :mod:`oracle.host_prep` — numpy restatement of pyslam/utilities/depth.py + geometry.inv_T.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PORT_SO = os.path.join(_HERE, "liboracle_port.so")
_REF_SO = os.path.join(_HERE, "_ref", "libref_volumetric.so")

_c = ctypes
_vp, _i64, _i32, _f32, _f64 = _c.c_void_p, _c.c_int64, _c.c_int, _c.c_float, _c.c_double


def build(want_ref=True):
    """Compile the C restatement (always) and the reference shim (if /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "port"])
    if want_ref:
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


def _load_port():
    if not os.path.exists(_PORT_SO):
        build(want_ref=False)
    L = _c.CDLL(_PORT_SO)
    L.vo_create.restype = _vp
    L.vo_create.argtypes = [_f32, _i32]
    L.vo_destroy.argtypes = [_vp]
    L.vo_clear.argtypes = [_vp]
    L.vo_integrate.argtypes = [_vp, _vp, _i64, _vp, _i32]
    L.vo_integrate_omp.restype = None
    L.vo_integrate_omp.argtypes = [_vp, _vp, _i64, _vp, _i32, _i32]
    for name in ("vo_num_blocks", "vo_size"):
        getattr(L, name).restype = _i64
        getattr(L, name).argtypes = [_vp]
    L.vo_block_size.argtypes = [_vp]
    L.vo_empty.argtypes = [_vp]
    L.vo_remove_low_count.argtypes = [_vp, _i32]
    L.vo_dump.restype = _i64
    L.vo_dump.argtypes = [_vp, _vp, _vp, _vp, _vp]
    L.vo_get_voxels.restype = _i64
    L.vo_get_voxels.argtypes = [_vp, _i32, _f32, _vp, _vp, _i64]
    L.vo_get_voxels_in_bb.restype = _i64
    L.vo_get_voxels_in_bb.argtypes = [_vp, _vp, _i32, _f32, _vp, _vp, _i64]
    L.vo_get_voxels_in_frustum.restype = _i64
    L.vo_get_voxels_in_frustum.argtypes = [_vp, _vp, _i32, _i32, _vp, _f32, _f32, _i32, _f32, _vp, _vp, _i64]
    L.vo_carve.argtypes = [_vp, _vp, _i32, _i32, _vp, _f32, _f32, _vp, _f32]
    L.vo_frustum_contains_pt.argtypes = [_vp, _i32, _i32, _vp, _f32, _f32, _vp, _vp]
    L.vo_frustum_bbox.argtypes = [_vp, _i32, _i32, _vp, _f32, _f32, _vp]
    L.vo_keys.argtypes = [_f32, _i32, _vp, _i64, _vp, _vp, _vp, _vp]
    # tsdf
    L.to_create.restype = _vp
    L.to_create.argtypes = [_f64, _f64, _i32, _i32]
    L.to_destroy.argtypes = [_vp]
    L.to_reset.argtypes = [_vp]
    L.to_set_threads.argtypes = [_vp, _i32]
    L.to_integrate.argtypes = [_vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _f64, _f64]
    for name in ("to_num_units", "to_num_touched", "to_last_updated"):
        getattr(L, name).restype = _i64
        getattr(L, name).argtypes = [_vp]
    L.to_batch_begin.argtypes = [_vp]
    L.to_batch_end.argtypes = [_vp, _vp, _vp]
    L.to_touched_keys.restype = _i64
    L.to_touched_keys.argtypes = [_vp, _vp]
    L.to_dump.restype = _i64
    L.to_dump.argtypes = [_vp, _vp, _vp, _vp, _vp]
    L.to_extract_mesh.restype = _i64
    L.to_extract_mesh.argtypes = [_vp, _vp, _vp, _i64, _vp, _i64, _vp]
    L.to_extract_points.restype = _i64
    L.to_extract_points.argtypes = [_vp, _vp, _vp, _i64]
    L.to_point_normals.restype = None
    L.to_point_normals.argtypes = [_vp, _vp, _i64, _vp]
    L.to_invert4x4.argtypes = [_vp, _vp]
    L.to_load_unit.restype = None
    L.to_load_unit.argtypes = [_vp, _vp, _vp, _vp, _vp]
    return L


def _load_ref():
    if not os.path.exists(_REF_SO):
        if os.path.isdir("/root/reference/cpp/volumetric"):
            build(want_ref=True)
        else:
            raise FileNotFoundError(
                f"{_REF_SO} missing and /root/reference absent; build it in the dev container"
            )
    L = _c.CDLL(_REF_SO)
    L.ref_grid_create.restype = _vp
    L.ref_grid_create.argtypes = [_f32, _i32]
    L.ref_grid_destroy.argtypes = [_vp]
    L.ref_grid_clear.argtypes = [_vp]
    L.ref_grid_integrate.argtypes = [_vp, _vp, _i64, _vp, _i32]
    L.ref_grid_integrate_f64.argtypes = [_vp, _vp, _i64, _vp, _i32]
    for name in ("ref_grid_num_blocks", "ref_grid_size", "ref_grid_total_voxel_count"):
        getattr(L, name).restype = _i64
        getattr(L, name).argtypes = [_vp]
    L.ref_grid_block_size.argtypes = [_vp]
    L.ref_grid_empty.argtypes = [_vp]
    L.ref_grid_remove_low_count.argtypes = [_vp, _i32]
    L.ref_grid_dump.restype = _i64
    L.ref_grid_dump.argtypes = [_vp, _vp, _vp, _vp, _vp]
    L.ref_grid_get_voxels.restype = _i64
    L.ref_grid_get_voxels.argtypes = [_vp, _i32, _f32, _vp, _vp, _i64]
    L.ref_grid_get_voxels_in_bb.restype = _i64
    L.ref_grid_get_voxels_in_bb.argtypes = [_vp, _vp, _i32, _f32, _vp, _vp, _i64]
    L.ref_grid_get_voxels_in_frustum.restype = _i64
    L.ref_grid_get_voxels_in_frustum.argtypes = [_vp, _vp, _i32, _i32, _vp, _f32, _f32, _i32, _f32, _vp, _vp, _i64]
    L.ref_grid_carve.argtypes = [_vp, _vp, _i32, _i32, _vp, _f32, _f32, _vp, _f32]
    L.ref_frustum_contains.argtypes = [_vp, _i32, _i32, _vp, _f32, _f32, _vp, _vp]
    L.ref_frustum_bbox.argtypes = [_vp, _i32, _i32, _vp, _f32, _f32, _vp]
    L.ref_keys.argtypes = [_f32, _i32, _vp, _i64, _vp, _vp, _vp, _vp]
    return L


_port = None
_ref = None


def port_lib():
    global _port
    if _port is None:
        _port = _load_port()
    return _port


def use_native_port():
    """bench.py's cpu_baseline leg: rebuild the C restatement with -O3 -march=native FOR THE HOST IT RUNS ON (the
    travelling liboracle_port.so is built without -march so that it loads anywhere) into a per-host temp dir and use
    it from now on.  Same sources, same -ffp-contract=off: results are unchanged, only the baseline gets faster.
    Falls back silently to the travelling library when no compiler is around."""
    global _port
    import hashlib
    import tempfile

    srcs = [os.path.join(_HERE, n) for n in ("voxel_oracle.c", "tsdf_oracle.c", "semantic_oracle.c", "semantic2_oracle.c")]
    mc_dir = os.path.join(os.path.dirname(_HERE), "pyslam_amd", "csrc")
    h = hashlib.sha256()
    for path in srcs + [os.path.join(mc_dir, "mc_tables.h")]:
        with open(path, "rb") as f:
            h.update(f.read())
    out_dir = os.path.join(tempfile.gettempdir(), f"pyslam_amd_oracle_native_{h.hexdigest()[:16]}")
    so = os.path.join(out_dir, "liboracle_port_native.so")
    try:
        if not os.path.exists(so):
            os.makedirs(out_dir, exist_ok=True)
            tmp = f"{so}.{os.getpid()}.tmp"
            subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-fopenmp",
                                   "-I" + mc_dir, "-o", tmp] + srcs + ["-lm"], stderr=subprocess.DEVNULL)
            os.replace(tmp, so)
        global _PORT_SO
        prev = _PORT_SO
        _PORT_SO = so
        try:
            _port = _load_port()
        finally:
            _PORT_SO = prev
        return True
    except (OSError, subprocess.CalledProcessError):
        return False


def ref_lib():
    global _ref
    if _ref is None:
        _ref = _load_ref()
    return _ref


def ref_available():
    return os.path.exists(_REF_SO) or os.path.isdir("/root/reference/cpp/volumetric")


def _color_kind(colors):
    if colors is None:
        return 0, None
    if colors.dtype == np.uint8:
        return 1, np.ascontiguousarray(colors)
    return 2, np.ascontiguousarray(colors, dtype=np.float32)


class _GridBase:
    """Shared ctypes plumbing: subclasses give the symbol prefix and library."""

    _names = None  # dict of method -> symbol
    _lib = None

    def __init__(self, voxel_size, block_size=8):
        self.voxel_size = float(np.float32(voxel_size))
        self.block_size = int(block_size)
        self._h = getattr(self._lib, self._names["create"])(self.voxel_size, self.block_size)

    def __del__(self):
        if getattr(self, "_h", None):
            getattr(self._lib, self._names["destroy"])(self._h)
            self._h = None

    def integrate(self, points, colors=None):
        """float64 points take the binding's double overload where the library has it (the compiled reference); everything
        else goes through float32, as pybind11's no-convert pass dispatches."""
        kind, cols = _color_kind(colors)
        if np.asarray(points).dtype == np.float64 and "integrate_f64" in self._names:
            points = np.ascontiguousarray(points, dtype=np.float64)
            getattr(self._lib, self._names["integrate_f64"])(self._h, _ptr(points), points.shape[0], _ptr(cols), kind)
            return
        points = np.ascontiguousarray(points, dtype=np.float32)
        getattr(self._lib, self._names["integrate"])(self._h, _ptr(points), points.shape[0], _ptr(cols), kind)

    def num_blocks(self):
        return getattr(self._lib, self._names["num_blocks"])(self._h)

    def size(self):
        return getattr(self._lib, self._names["size"])(self._h)

    def empty(self):
        return bool(getattr(self._lib, self._names["empty"])(self._h))

    def clear(self):
        getattr(self._lib, self._names["clear"])(self._h)

    reset = clear

    def remove_low_count_voxels(self, min_count):
        getattr(self._lib, self._names["remove_low_count"])(self._h, int(min_count))

    def dump(self):
        """-> keys [B,3] i32, hashes [B] u64, counts [B,bs^3] i32, sums [B,bs^3,6] f32 (key-sorted)."""
        nb = self.num_blocks()
        nv = self.block_size ** 3
        keys = np.zeros((nb, 3), np.int32)
        hashes = np.zeros(nb, np.uint64)
        counts = np.zeros((nb, nv), np.int32)
        sums = np.zeros((nb, nv, 6), np.float32)
        getattr(self._lib, self._names["dump"])(self._h, _ptr(keys), _ptr(hashes), _ptr(counts), _ptr(sums))
        return keys, hashes, counts, sums

    def _rows(self, fn, *args):
        n = fn(*args, None, None, 0)
        pts = np.zeros((n, 3), np.float32)
        cols = np.zeros((n, 3), np.float32)
        if n:
            fn(*args, _ptr(pts), _ptr(cols), n)
        return pts, cols

    def get_voxels(self, min_count=1, min_confidence=0.0):
        return self._rows(getattr(self._lib, self._names["get_voxels"]), self._h, int(min_count), float(min_confidence))

    def get_voxels_in_bb(self, bb, min_count=1, min_confidence=0.0):
        bb = np.ascontiguousarray(bb, dtype=np.float64)
        fn = getattr(self._lib, self._names["get_voxels_in_bb"])
        return self._rows(lambda *a: fn(self._h, _ptr(bb), int(min_count), float(min_confidence), *a))

    def get_voxels_in_camera_frustrum(self, intr, width, height, T_cw, depth_max, depth_min, min_count=1, min_confidence=0.0):
        intr = np.ascontiguousarray(intr, dtype=np.float32)
        T = np.ascontiguousarray(T_cw, dtype=np.float64)
        fn = getattr(self._lib, self._names["get_voxels_in_frustum"])
        return self._rows(
            lambda *a: fn(self._h, _ptr(intr), width, height, _ptr(T), depth_max, depth_min, int(min_count), float(min_confidence), *a)
        )

    def carve(self, intr, width, height, T_cw, depth_max, depth_min, depth, depth_threshold):
        intr = np.ascontiguousarray(intr, dtype=np.float32)
        T = np.ascontiguousarray(T_cw, dtype=np.float64)
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        getattr(self._lib, self._names["carve"])(self._h, _ptr(intr), width, height, _ptr(T), depth_max, depth_min, _ptr(depth), depth_threshold)


class PortGrid(_GridBase):
    def __init__(self, voxel_size, block_size=8):
        self._lib = port_lib()
        self._names = {
            k: "vo_" + k
            for k in (
                "create destroy integrate num_blocks size empty clear remove_low_count dump "
                "get_voxels get_voxels_in_bb get_voxels_in_frustum carve"
            ).split()
        }
        super().__init__(voxel_size, block_size)

    def integrate_parallel(self, points, colors=None, threads=8):
        """The reference's TBB branch (integrate_raw_preorder_no_block_mutex, voxel_block_grid.hpp:292-456: thread-local
        grouping by block, sequential merge, blocks in parallel) restated with OpenMP - same result as integrate(), bit for bit."""
        kind, cols = _color_kind(colors)
        points = np.ascontiguousarray(points, dtype=np.float32)
        self._lib.vo_integrate_omp(self._h, _ptr(points), points.shape[0], _ptr(cols), kind, int(threads))


class RefGrid(_GridBase):
    def __init__(self, voxel_size, block_size=8):
        self._lib = ref_lib()
        self._names = {
            k: "ref_grid_" + k
            for k in (
                "create destroy integrate integrate_f64 num_blocks size empty clear remove_low_count dump "
                "get_voxels get_voxels_in_bb get_voxels_in_frustum carve"
            ).split()
        }
        super().__init__(voxel_size, block_size)


def keys(points, voxel_size, block_size=8, which="port"):
    """-> voxel_keys, block_keys, local_keys [N,3] i32 and BlockKeyHash [N] u64."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    n = points.shape[0]
    vk = np.zeros((n, 3), np.int32)
    bk = np.zeros((n, 3), np.int32)
    lk = np.zeros((n, 3), np.int32)
    h = np.zeros(n, np.uint64)
    fn = port_lib().vo_keys if which == "port" else ref_lib().ref_keys
    fn(float(np.float32(voxel_size)), int(block_size), _ptr(points), n, _ptr(vk), _ptr(bk), _ptr(lk), _ptr(h))
    return vk, bk, lk, h


def frustum_contains(intr, width, height, T_cw, depth_max, depth_min, p_w, which="port"):
    intr = np.ascontiguousarray(intr, dtype=np.float32)
    T = np.ascontiguousarray(T_cw, dtype=np.float64)
    p = np.ascontiguousarray(p_w, dtype=np.float32)
    out = np.zeros(3, np.float32)
    fn = port_lib().vo_frustum_contains_pt if which == "port" else ref_lib().ref_frustum_contains
    ok = fn(_ptr(intr), width, height, _ptr(T), depth_max, depth_min, _ptr(p), _ptr(out))
    return bool(ok), out


def ref_frustum_surface(intr, width, height, T_cw, depth_max, depth_min, points, orientation=None, translation=None):
    """The compiled reference's CameraFrustrum (oracle/ref_shim.cpp: ref_frustum_surface) -> dict: corners [8,3], obb [10], K [3,3],
    R_cw [3,3], t_cw [3], orientation_cw wxyz [4], and for the float64 points [N,3]: in_bbox, in_obb, inside, uvd [N,3] f32."""
    lib = ref_lib()
    fn = lib.ref_frustum_surface
    fn.restype = None
    fn.argtypes = [_vp, _i32, _i32, _vp, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]
    intr = np.ascontiguousarray(intr, dtype=np.float32)
    T = np.ascontiguousarray(np.eye(4) if T_cw is None else T_cw, dtype=np.float64)
    q = None if orientation is None else np.ascontiguousarray(orientation, dtype=np.float64)
    t = None if translation is None else np.ascontiguousarray(translation, dtype=np.float64)
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    n = len(pts)
    out = dict(corners=np.zeros((8, 3)), obb=np.zeros(10), K=np.zeros((3, 3)), R_cw=np.zeros((3, 3)), t_cw=np.zeros(3), orientation_cw=np.zeros(4),
               in_bbox=np.zeros(n, np.uint8), in_obb=np.zeros(n, np.uint8), inside=np.zeros(n, np.uint8), uvd=np.zeros((n, 3), np.float32))
    fn(_ptr(intr), int(width), int(height), _ptr(T), _ptr(q), _ptr(t), float(depth_max), float(depth_min), _ptr(out["corners"]), _ptr(out["obb"]),
       _ptr(out["K"]), _ptr(out["R_cw"]), _ptr(out["t_cw"]), _ptr(out["orientation_cw"]), _ptr(pts), n, _ptr(out["in_bbox"]), _ptr(out["in_obb"]),
       _ptr(out["inside"]), _ptr(out["uvd"]))
    return out


def frustum_bbox(intr, width, height, T_cw, depth_max, depth_min, which="port"):
    intr = np.ascontiguousarray(intr, dtype=np.float32)
    T = np.ascontiguousarray(T_cw, dtype=np.float64)
    bb = np.zeros(6, np.float64)
    fn = port_lib().vo_frustum_bbox if which == "port" else ref_lib().ref_frustum_bbox
    fn(_ptr(intr), width, height, _ptr(T), depth_max, depth_min, _ptr(bb))
    return bb


class PortTsdf:
    """Open3D-semantics TSDF volume (restated; see oracle/tsdf_oracle.c header)."""

    def __init__(self, voxel_length, sdf_trunc, unit_resolution=16, depth_sampling_stride=4, threads=1):
        self._lib = port_lib()
        self.voxel_length = float(voxel_length)
        self.sdf_trunc = float(sdf_trunc)
        self.res = int(unit_resolution)
        self._h = self._lib.to_create(self.voxel_length, self.sdf_trunc, self.res, int(depth_sampling_stride))
        self._lib.to_set_threads(self._h, int(threads))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.to_destroy(self._h)
            self._h = None

    def reset(self):
        self._lib.to_reset(self._h)

    def integrate(self, depth, rgb, intr, T_cw, depth_scale=1.0, depth_trunc=4.0):
        if depth.dtype == np.uint16:
            kind, depth = 1, np.ascontiguousarray(depth)
        else:
            kind, depth = 0, np.ascontiguousarray(depth, dtype=np.float32)
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        H, W = depth.shape
        intr = np.ascontiguousarray(intr, dtype=np.float64)
        T = np.ascontiguousarray(T_cw, dtype=np.float64)
        self._lib.to_integrate(self._h, _ptr(depth), kind, _ptr(rgb), H, W, _ptr(intr), _ptr(T), float(depth_scale), float(depth_trunc))

    def num_units(self):
        return self._lib.to_num_units(self._h)

    def num_touched(self):
        return self._lib.to_num_touched(self._h)

    def last_updated(self):
        return self._lib.to_last_updated(self._h)

    def batch_begin(self):
        """Accounting for bench.py: count the distinct units touched / voxels updated by the next integrate() calls."""
        self._lib.to_batch_begin(self._h)

    def batch_end(self):
        """-> (distinct units touched, distinct voxels updated) since batch_begin(); switches the accounting off."""
        u, v = _c.c_int64(0), _c.c_int64(0)
        self._lib.to_batch_end(self._h, _c.byref(u), _c.byref(v))
        return u.value, v.value

    def touched_keys(self):
        n = self._lib.to_num_touched(self._h)
        keys = np.zeros((n, 3), np.int32)
        self._lib.to_touched_keys(self._h, _ptr(keys))
        return keys

    def dump(self):
        """-> keys [U,3], tsdf [U,R^3] f32, weight [U,R^3] f32, color [U,R^3,3] f64 (0..255)."""
        nu = self.num_units()
        nv = self.res ** 3
        keys = np.zeros((nu, 3), np.int32)
        tsdf = np.zeros((nu, nv), np.float32)
        weight = np.zeros((nu, nv), np.float32)
        color = np.zeros((nu, nv, 3), np.float64)
        self._lib.to_dump(self._h, _ptr(keys), _ptr(tsdf), _ptr(weight), _ptr(color))
        return keys, tsdf, weight, color

    def load_units(self, keys, tsdf, weight, color):
        """Test hook: units `keys` [U,3] get the voxel states tsdf / weight [U,R,R,R] (x, y, z) and color [U,R,R,R,3] (0..255)."""
        keys = np.ascontiguousarray(keys, dtype=np.int32)
        for k in range(len(keys)):
            t = np.ascontiguousarray(tsdf[k], dtype=np.float32).reshape(-1)
            w = np.ascontiguousarray(weight[k], dtype=np.float32).reshape(-1)
            c = np.ascontiguousarray(color[k], dtype=np.float64).reshape(-1)
            assert t.size == self.res ** 3 and w.size == t.size and c.size == 3 * t.size
            self._lib.to_load_unit(self._h, _ptr(keys[k]), _ptr(t), _ptr(w), _ptr(c))

    def extract_triangle_mesh(self):
        nt = _c.c_int64(0)
        nv = self._lib.to_extract_mesh(self._h, None, None, 0, None, 0, _c.byref(nt))
        verts = np.zeros((nv, 3), np.float64)
        cols = np.zeros((nv, 3), np.float64)
        tris = np.zeros((nt.value, 3), np.int32)
        self._lib.to_extract_mesh(self._h, _ptr(verts), _ptr(cols), nv, _ptr(tris), nt.value, _c.byref(nt))
        return verts, tris, cols

    def extract_point_cloud(self):
        n = self._lib.to_extract_points(self._h, None, None, 0)
        pts = np.zeros((n, 3), np.float64)
        cols = np.zeros((n, 3), np.float64)
        self._lib.to_extract_points(self._h, _ptr(pts), _ptr(cols), n)
        return pts, cols

    def point_normals(self, points):
        """ScalableTSDFVolume::GetNormalAt for each point [N,3] f64 (the normals o3d's extract_point_cloud() carries)."""
        points = np.ascontiguousarray(points, dtype=np.float64)
        normals = np.zeros_like(points)
        self._lib.to_point_normals(self._h, _ptr(points), len(points), _ptr(normals))
        return normals


def invert4x4(T):
    T = np.ascontiguousarray(T, dtype=np.float64)
    out = np.zeros((4, 4), np.float64)
    port_lib().to_invert4x4(_ptr(T), _ptr(out))
    return out


class RefVoxelGrid:
    """The compiled reference's direct voxel hash, volumetric::VoxelGrid (cpp/volumetric/voxel_grid.h)."""

    def __init__(self, voxel_size):
        L = ref_lib()
        L.ref_vgrid_create.restype = _vp
        L.ref_vgrid_create.argtypes = [_f64]
        L.ref_vgrid_destroy.argtypes = [_vp]
        L.ref_vgrid_size.restype = _i64
        L.ref_vgrid_size.argtypes = [_vp]
        L.ref_vgrid_integrate.argtypes = [_vp, _vp, _i64, _vp, _i32]
        L.ref_vgrid_get_voxels.restype = _i64
        L.ref_vgrid_get_voxels.argtypes = [_vp, _i32, _f32, _vp, _vp, _i64]
        self._lib = L
        self._h = L.ref_vgrid_create(float(voxel_size))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.ref_vgrid_destroy(self._h)
            self._h = None

    def integrate(self, points, colors=None):
        points = np.ascontiguousarray(points, dtype=np.float32)
        kind, cols = _color_kind(colors)
        self._lib.ref_vgrid_integrate(self._h, _ptr(points), points.shape[0], _ptr(cols), kind)

    def size(self):
        return self._lib.ref_vgrid_size(self._h)

    def get_voxels(self, min_count=1, min_confidence=0.0):
        n = self._lib.ref_vgrid_get_voxels(self._h, int(min_count), float(min_confidence), None, None, 0)
        pts = np.zeros((n, 3), np.float32)
        cols = np.zeros((n, 3), np.float32)
        if n:
            self._lib.ref_vgrid_get_voxels(self._h, int(min_count), float(min_confidence), _ptr(pts), _ptr(cols), n)
        return pts, cols
