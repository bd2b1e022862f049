"""Host-side mirror of the two volume objects pySLAM's dense integrators drive.

* :class:`VoxelBlockGrid`, :class:`CameraFrustrum`, :class:`BoundingBox3D`, :class:`VoxelGridData`
  mirror the ``volumetric`` pybind11 module (reference: cpp/volumetric/volumetric_grid_module.h:
  732-935, camera_frustrum_module.h:41-130) — same method names, argument meaning, defaults and
  error messages — over the HIP library's VOXEL_GRID mode.
* :class:`ScalableTSDFVolume`, :class:`PinholeCameraIntrinsic`, :class:`TriangleMesh`,
  :class:`PointCloud` mirror the slice of ``open3d`` that pyslam/dense/volumetric_integrator_tsdf.py
  uses (:104-119, 215-223, 239-267) over the library's TSDF mode.

All compute happens in libpyslam_hipvol.so on the GPU; nothing here falls back to the CPU.
Arrays may be numpy (host) or torch CUDA tensors (zero-copy, already resident in HBM).
"""
import ctypes

import numpy as np

from . import _lib as L


def _result_array(shape, dtype):
    """Host array a large device result is copied into.  Page-locked when torch can provide it (hipHostMalloc behind torch's
    caching host allocator: the block is reused from one output tick to the next) - the D2H copy of a 270 MB mesh then runs
    as one DMA at PCIe speed instead of through the runtime's pageable staging; plain numpy otherwise or with
    PYSLAM_AMD_PINNED_RESULTS=0.  Either way the caller gets an ordinary writable numpy array."""
    import os

    n = int(np.prod(shape))
    if n * np.dtype(dtype).itemsize >= (1 << 20) and os.environ.get("PYSLAM_AMD_PINNED_RESULTS", "1") != "0":
        try:
            import torch

            if torch.cuda.is_available():
                return torch.empty(tuple(shape), dtype=getattr(torch, np.dtype(dtype).name), pin_memory=True).numpy()
        except Exception:  # no torch / no pinned memory left: the pageable copy is still correct
            pass
    return np.empty(shape, dtype)


def _as_f64_4x4(T):
    T = np.ascontiguousarray(np.asarray(T, dtype=np.float64))
    if T.shape != (4, 4):
        raise RuntimeError("T_cw must be a 4x4 matrix")
    return T


class _Volume:
    """Owns one hv_volume handle."""

    def __init__(self, mode, voxel_size, sdf_trunc, block_size, stride, device, max_blocks, max_points):
        lib = L.load()
        cfg = L.HvConfig()
        lib.hv_default_config(mode, ctypes.byref(cfg))
        cfg.device = int(device)
        cfg.voxel_size = float(voxel_size)
        cfg.sdf_trunc = float(sdf_trunc)
        cfg.block_size = int(block_size)
        cfg.depth_sampling_stride = int(stride)
        if max_blocks is not None:
            cfg.max_blocks = int(max_blocks)
        if max_points is not None:
            cfg.max_points = int(max_points)
        self._lib = lib
        self._cfg = cfg
        handle = ctypes.c_void_p()
        L.check(lib.hv_create(ctypes.byref(cfg), ctypes.byref(handle)))
        self._h = handle

    def close(self):
        if getattr(self, "_h", None):
            self._lib.hv_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- shared introspection -------------------------------------------------------------------
    def num_blocks(self):
        n = ctypes.c_int64()
        L.check(self._lib.hv_num_blocks(self._h, ctypes.byref(n)))
        return n.value

    def max_blocks(self):
        n = ctypes.c_int64()
        L.check(self._lib.hv_max_blocks(self._h, ctypes.byref(n)))
        return n.value

    def reserve_blocks(self, new_max_blocks):
        """Grow the block pool / hash, keeping the contents (also happens automatically when more than half full)."""
        L.check(self._lib.hv_reserve_blocks(self._h, int(new_max_blocks)))

    def synchronize(self):
        L.check(self._lib.hv_synchronize(self._h))

    def set_stream(self, stream_handle):
        """Adopt a caller-owned hipStream_t (e.g. ``torch.cuda.Stream().cuda_stream``)."""
        L.check(self._lib.hv_set_stream(self._h, ctypes.c_void_p(int(stream_handle))))

    def register_host_memory(self, address, nbytes):
        """Page-lock [address, address + nbytes) (hv_host_register): host frames inside it are DMA'd in place."""
        L.check(self._lib.hv_host_register(ctypes.c_void_p(int(address)), int(nbytes)))

    def unregister_host_memory(self, address):
        L.check(self._lib.hv_host_unregister(ctypes.c_void_p(int(address))))

    # -- torch CUDA tensors handed to / returned by the C ABI ------------------------------------------
    def _torch_stream(self, device):
        """The volume's hipStream_t as a torch stream (for event ordering against torch's streams and the caching allocator)."""
        import torch

        h = int(self._lib.hv_get_stream(self._h) or 0)
        if getattr(self, "_ts_key", None) != (h, device):
            self._ts = torch.cuda.ExternalStream(h, device=device) if h else torch.cuda.default_stream(device)
            self._ts_key = (h, device)
        return self._ts

    def adopt_torch_stream(self):
        """Run this volume's launches on a torch stream of its own device and hand that stream back: inside
        ``with torch.cuda.stream(s)`` torch ops (uploads, allocations) and the volume's kernels are ordered by the one queue, and
        _torch_in / _torch_out need no cross-stream event waits (eight per semantic keyframe otherwise)."""
        import torch

        if getattr(self, "_adopted", None) is None:
            dev = torch.device("cuda", int(self._cfg.device))
            self._adopted = torch.cuda.Stream(dev)
            self.set_stream(self._adopted.cuda_stream)
        return self._adopted

    def _torch_in(self, *tensors):
        """Before a launch that reads / writes torch CUDA tensors on the volume's stream: that stream waits for what torch's
        current stream has queued (the producers).  No host synchronisation.  -> the volume's torch stream, or None when no
        tensor is on a GPU.  Always paired with _torch_out() after the launch."""
        import torch

        for t in tensors:
            if t is not None and getattr(t, "is_cuda", False):
                cur = torch.cuda.current_stream(t.device)
                if int(cur.cuda_stream) == int(self._lib.hv_get_stream(self._h) or 0) and int(cur.cuda_stream) != 0:
                    return None  # the volume runs on torch's current stream (adopt_torch_stream): already ordered
                ts = self._torch_stream(t.device)
                ts.wait_stream(cur)
                return ts
        return None

    def _torch_out(self, ts, device):
        """After such a launch: torch's current stream waits for the volume's.  Torch ops on the results are ordered after
        the launch, and so is everything the caching allocator may later place in a block the caller drops (a block freed
        on torch's stream is only reused by work queued on that stream), so tensors may be released right after the call.
        (record_stream() on the volume's stream would say the same to the allocator, but leaves it holding events on a stream
        that hv_destroy() may already have destroyed when the tensor is finally freed.)"""
        import torch

        if ts is not None:
            torch.cuda.current_stream(device).wait_stream(ts)

    def _carve(self, camera_frustrum, depth_image, depth_threshold):
        """carve(camera_frustrum, depth f32 HxW, threshold) of every grid type (voxel_grid_carving.h:47-79).  Host array or
        torch CUDA tensor (used in place, ordered against torch's stream)."""
        f = camera_frustrum
        if hasattr(depth_image, "data_ptr"):
            depth = depth_image.contiguous().float()
        else:
            depth = np.ascontiguousarray(depth_image, dtype=np.float32)
        if depth.ndim != 2 or depth.shape[0] * depth.shape[1] == 0 or depth.shape[0] != f.height or depth.shape[1] != f.width:
            return  # "Depth image is empty" / check_image_size(): the reference prints a message and returns
        loc = L.location(depth)
        ts = self._torch_in(depth) if loc == L.HV_DEVICE else None
        L.check(self._lib.hv_carve(self._h, L.ptr(f.intr), f.width, f.height, L.ptr(f.T_cw), f.depth_max, f.depth_min, L.ptr(depth),
                                   float(depth_threshold), loc))
        if ts is not None:
            self._torch_out(ts, depth.device)

    def dropped_points(self):
        n = ctypes.c_int64()
        L.check(self._lib.hv_dropped_points(self._h, ctypes.byref(n)))
        return n.value

    def profile_enable(self, on=True):
        L.check(self._lib.hv_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        ms, launches, units = ctypes.c_double(), ctypes.c_int64(), ctypes.c_int64()
        L.check(self._lib.hv_profile_read(self._h, ctypes.byref(ms), ctypes.byref(launches), ctypes.byref(units)))
        return ms.value, launches.value, units.value

    def profile_launches(self):
        """Durations (ms) of the bracketed launches since profile_enable / the last profile_read, in issue order."""
        n = ctypes.c_int64()
        L.check(self._lib.hv_profile_read_launches(self._h, None, 0, ctypes.byref(n)))
        out = np.zeros(n.value, np.float32)
        if n.value:
            L.check(self._lib.hv_profile_read_launches(self._h, L.ptr(out), n.value, ctypes.byref(n)))
        return out

    def filter_shadow_points(self, depth, delta_x=2, delta_y=2, fill_value=-1.0, stream=None):
        """pyslam.utilities.depth.filter_shadow_points(depth, delta_depth=None, ...) on the GPU.
        stream: a torch.cuda.Stream - the launches of a CUDA tensor's filter go to THAT stream (which must be torch's current one:
        the result is allocated on it) instead of the volume's; the filter reads nothing of the volume, so it may run beside the
        volume's kernels (device_pipeline.KeyframeUploader: the next keyframe's depth is filtered while this one is fused)."""
        if stream is not None:
            import torch

            d = depth.contiguous().float()
            out = torch.empty_like(d)
            L.check(self._lib.hv_filter_shadow_points_on_stream(self._h, L.ptr(d), int(d.shape[0]), int(d.shape[1]), int(delta_x),
                                                                int(delta_y), float(fill_value), L.ptr(out), int(stream.cuda_stream)))
            return out
        if hasattr(depth, "data_ptr"):
            import torch

            d = depth.contiguous().float()
            out = torch.empty_like(d)
        else:
            d = np.ascontiguousarray(depth, dtype=np.float32)
            out = np.empty_like(d)
        H, W = int(d.shape[0]), int(d.shape[1])
        ts = self._torch_in(d, out) if L.location(d) == L.HV_DEVICE else None
        L.check(self._lib.hv_filter_shadow_points(self._h, L.ptr(d), H, W, int(delta_x), int(delta_y), float(fill_value),
                                                  L.ptr(out), L.location(d)))
        if ts is not None:
            self._torch_out(ts, d.device)
        return out

    def remap(self, img, map_x, map_y, linear=False):
        """cv2.remap(img, map_x, map_y, INTER_LINEAR if linear else INTER_NEAREST) on the GPU.  Host arrays, or a
        torch CUDA tensor (the maps are then uploaded once and cached): the image never leaves the device."""
        if hasattr(img, "data_ptr") and img.is_cuda:
            import torch

            src = img.contiguous()
            kind = {torch.uint8: 0, torch.float32: 1, torch.int32: 2}.get(src.dtype)
            if kind is None:
                raise RuntimeError(f"remap: unsupported image dtype {src.dtype}")
            key = (id(map_x), id(map_y))
            if getattr(self, "_dev_maps_key", None) != key:
                self._dev_maps = (torch.from_numpy(np.ascontiguousarray(map_x, dtype=np.float32)).to(src.device),
                                  torch.from_numpy(np.ascontiguousarray(map_y, dtype=np.float32)).to(src.device))
                self._dev_maps_key = key
            mx, my = self._dev_maps
            H, W = int(src.shape[0]), int(src.shape[1])
            C = 1 if src.dim() == 2 else int(src.shape[2])
            out = torch.empty_like(src)
            torch.cuda.current_stream(src.device).synchronize()  # producers ran on torch's stream, hv_remap on the volume's
            L.check(self._lib.hv_remap(self._h, L.ptr(src), kind, C, H, W, L.ptr(mx), L.ptr(my), 1 if linear else 0, L.ptr(out),
                                       L.HV_DEVICE))
            self.synchronize()
            return out
        img = np.ascontiguousarray(img)
        kind = {np.dtype(np.uint8): 0, np.dtype(np.float32): 1, np.dtype(np.int32): 2}.get(img.dtype)
        if kind is None:
            raise RuntimeError(f"remap: unsupported image dtype {img.dtype}")
        H, W = img.shape[:2]
        C = 1 if img.ndim == 2 else img.shape[2]
        mx = np.ascontiguousarray(map_x, dtype=np.float32)
        my = np.ascontiguousarray(map_y, dtype=np.float32)
        out = np.empty_like(img)
        L.check(self._lib.hv_remap(self._h, L.ptr(img), kind, C, H, W, L.ptr(mx), L.ptr(my), 1 if linear else 0, L.ptr(out), L.HV_HOST))
        return out

    def bytes_per_block(self):
        n = ctypes.c_int64()
        L.check(self._lib.hv_bytes_per_block(self._h, ctypes.byref(n)))
        return n.value


# ================================================================================================
# `volumetric` module mirror
# ================================================================================================
class BoundingBox3D:
    """Axis-aligned box: cpp/volumetric/bounding_boxes_3d.h:39-80, bounding_boxes_3d.cpp:174-247, bindings
    bounding_boxes_module.h:49-93.  BoundingBox3D(), BoundingBox3D(min_point, max_point) or the six scalars."""

    def __init__(self, *args):
        if len(args) == 0:
            vals = (0.0,) * 6
        elif len(args) == 2:
            vals = tuple(float(x) for x in args[0]) + tuple(float(x) for x in args[1])
        elif len(args) == 6:
            vals = tuple(float(x) for x in args)
        else:
            raise TypeError("BoundingBox3D(), BoundingBox3D(min_point, max_point) or BoundingBox3D(min_x, min_y, min_z, max_x, max_y, max_z)")
        self.min_x, self.min_y, self.min_z, self.max_x, self.max_y, self.max_z = vals

    def as_array(self):
        return np.array([self.min_x, self.min_y, self.min_z, self.max_x, self.max_y, self.max_z], np.float64)

    def get_min_point(self):
        return np.array([self.min_x, self.min_y, self.min_z], np.float64)

    def get_max_point(self):
        return np.array([self.max_x, self.max_y, self.max_z], np.float64)

    def get_center(self):
        return (self.get_min_point() + self.get_max_point()) / 2.0

    def get_size(self):
        return self.get_max_point() - self.get_min_point()

    def get_volume(self):
        sx, sy, sz = self.get_size()
        return float(sx * sy * sz)

    def get_surface_area(self):
        sx, sy, sz = self.get_size()
        return float(2.0 * (sx * sy + sx * sz + sy * sz))

    def get_diagonal_length(self):
        sx, sy, sz = self.get_size()
        return float(np.sqrt(sx * sx + sy * sy + sz * sz))

    def contains(self, points):
        """One point [3] -> bool; several [N,3] -> list of bool (the binding's two overloads).  Closed box."""
        p = np.asarray(points, np.float64)
        m = np.all((p >= self.get_min_point()) & (p <= self.get_max_point()), axis=-1)
        return bool(m) if p.ndim == 1 else [bool(x) for x in m]

    def intersects(self, other):
        return bool(np.all((self.get_min_point() <= other.get_max_point()) & (self.get_max_point() >= other.get_min_point())))

    @staticmethod
    def compute_from_points(points):
        p = np.asarray(points, np.float64).reshape(-1, 3)
        if len(p) == 0:
            return BoundingBox3D()
        return BoundingBox3D(p.min(axis=0), p.max(axis=0))


class ImagePoint:
    """``volumetric.ImagePoint`` (camera_frustrum.h:31-35, camera_frustrum_module.h:41-47): pixel coordinates and depth as float32."""

    def __init__(self, u=0.0, v=0.0, depth=0.0):
        self.u, self.v, self.depth = float(np.float32(u)), float(np.float32(v)), float(np.float32(depth))

    def __repr__(self):
        return f"ImagePoint(u={self.u}, v={self.v}, depth={self.depth})"


class Quaterniond:
    """``volumetric.Quaterniond`` (eigen_module.h:36-96): Eigen::Quaterniond as the module exposes it - Quaterniond() identity,
    Quaterniond(w, x, y, z), Quaterniond([w, x, y, z]); w() x() y() z(), coeffs() -> [w, x, y, z], normalized(), normalize(),
    conjugate(), inverse(), toRotationMatrix(); picklable."""

    def __init__(self, *args):
        if len(args) == 0:
            c = (1.0, 0.0, 0.0, 0.0)
        elif len(args) == 4:
            c = args
        elif len(args) == 1 and np.size(args[0]) == 4:
            c = np.asarray(args[0], np.float64).reshape(4)
        else:
            raise RuntimeError("Quaternion array must have exactly 4 elements [w, x, y, z]")
        self._c = np.array([float(v) for v in c], np.float64)  # w, x, y, z

    def w(self):
        return float(self._c[0])

    def x(self):
        return float(self._c[1])

    def y(self):
        return float(self._c[2])

    def z(self):
        return float(self._c[3])

    def coeffs(self):
        return self._c.copy()

    def normalized(self):
        return Quaterniond(self._c / np.sqrt(np.sum(self._c * self._c)))

    def normalize(self):
        self._c = self._c / np.sqrt(np.sum(self._c * self._c))

    def conjugate(self):
        return Quaterniond(self._c[0], -self._c[1], -self._c[2], -self._c[3])

    def inverse(self):
        n2 = float(np.sum(self._c * self._c))
        return Quaterniond(self.conjugate()._c / n2) if n2 > 0.0 else Quaterniond(0.0, 0.0, 0.0, 0.0)

    def toRotationMatrix(self):
        w, x, y, z = self._c  # (Eigen does not normalise here)
        tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
        twx, twy, twz = tx * w, ty * w, tz * w
        txx, txy, txz = tx * x, ty * x, tz * x
        tyy, tyz, tzz = ty * y, tz * y, tz * z
        return np.array([[1.0 - (tyy + tzz), txy - twz, txz + twy], [txy + twz, 1.0 - (txx + tzz), tyz - twx], [txz - twy, tyz + twx, 1.0 - (txx + tyy)]])

    def __repr__(self):
        return "Quaterniond(w=%f, x=%f, y=%f, z=%f)" % tuple(self._c)

    def __reduce__(self):
        return (Quaterniond, tuple(float(v) for v in self._c))


def _quat_from_matrix(m):
    """Eigen::Quaterniond(Matrix3d) (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl<Other, 3, 3>) -> (w, x, y, z)."""
    t = m[0, 0] + m[1, 1] + m[2, 2]
    q = np.zeros(4, np.float64)  # x, y, z, w
    if t > 0.0:
        t = np.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0], q[1], q[2] = (m[2, 1] - m[1, 2]) * t, (m[0, 2] - m[2, 0]) * t, (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (m[k, j] - m[j, k]) * t
        q[j] = (m[j, i] + m[i, j]) * t
        q[k] = (m[k, i] + m[i, k]) * t
    return np.array([q[3], q[0], q[1], q[2]])


def _matrix_from_quat(wxyz):
    """Eigen::Quaterniond::normalized().toRotationMatrix()."""
    w, x, y, z = np.asarray(wxyz, np.float64) / np.sqrt(np.sum(np.asarray(wxyz, np.float64) ** 2))
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1.0 - (tyy + tzz), txy - twz, txz + twy], [txy + twz, 1.0 - (txx + tzz), tyz - twx], [txz - twy, tyz + twx, 1.0 - (txx + tyy)]])


def _mat3_vec(R, v):
    """R v in Eigen's coefficient order ((r0 v0 + r1 v1) + r2 v2 per row): the same float64 bits as the reference's products."""
    return R[:, 0] * v[0] + R[:, 1] * v[1] + R[:, 2] * v[2]


class CameraFrustrum:
    """``volumetric.CameraFrustrum`` (cpp/volumetric/camera_frustrum.h:37-130, camera_frustrum.cpp, bindings camera_frustrum_module.h:50-130).
    The binding's three constructors: ``CameraFrustrum(fx, fy, cx, cy, width, height, T_cw, depth_max, depth_min)``,
    ``CameraFrustrum(K, width, height, T_cw, depth_max, depth_min)`` and ``CameraFrustrum(fx, fy, cx, cy, width, height, orientation,
    translation, depth_max, depth_min)`` (orientation: quaternion (w, x, y, z) or an object with .w() .. .z()); keywords as there.
    Intrinsics and depth limits are float32 members, the pose float64.  Corners, boxes and the point tests are the host-side
    geometry of the reference class; the grids take the frustum as a query (hv_query.h)."""

    def __init__(self, *args, **kw):
        names9 = ("fx", "fy", "cx", "cy", "width", "height", "T_cw", "depth_max", "depth_min")
        names6 = ("K", "width", "height", "T_cw", "depth_max", "depth_min")
        names10 = ("fx", "fy", "cx", "cy", "width", "height", "orientation", "translation", "depth_max", "depth_min")
        if "K" in kw or (args and np.ndim(args[0]) == 2):
            a = dict(zip(names6, args), **kw)
            K = np.asarray(a["K"], np.float64)
            a.update(fx=K[0, 0], fy=K[1, 1], cx=K[0, 2], cy=K[1, 2])
        elif "orientation" in kw or len(args) == 10:
            a = dict(zip(names10, args), **kw)
        else:
            a = dict(zip(names9, args), **kw)
        self.intr = np.array([a["fx"], a["fy"], a["cx"], a["cy"]], dtype=np.float32)
        self.width = int(a["width"])
        self.height = int(a["height"])
        self.depth_max = float(np.float32(a.get("depth_max", 10.0)))
        self.depth_min = float(np.float32(a.get("depth_min", 1e-2)))
        self._cache = None
        if "orientation" in a:
            self.set_T_cw(a["orientation"], a["translation"])
        else:
            self.set_T_cw(np.eye(4) if a.get("T_cw") is None else a["T_cw"])

    # ---- setters (each drops the cached corners / boxes, camera_frustrum.cpp:64-121) ----
    def set_T_cw(self, T_cw, translation=None):
        """set_T_cw(T_cw 4x4) or set_T_cw(orientation, translation)."""
        if translation is not None:
            q = T_cw
            wxyz = np.array([q.w(), q.x(), q.y(), q.z()], np.float64) if hasattr(q, "w") and callable(q.w) else np.asarray(q, np.float64)
            T = np.eye(4)
            T[:3, :3] = _matrix_from_quat(wxyz)
            T[:3, 3] = np.asarray(translation, np.float64)
            T_cw = T
        self.T_cw = _as_f64_4x4(T_cw)
        self._cache = None

    def set_width(self, width):
        self.width = int(width)
        self._cache = None

    def set_height(self, height):
        self.height = int(height)
        self._cache = None

    def set_depth_max(self, depth_max):
        self.depth_max = float(np.float32(depth_max))
        self._cache = None

    def set_depth_min(self, depth_min):
        self.depth_min = float(np.float32(depth_min))
        self._cache = None

    def set_intrinsics(self, *args, **kw):
        """set_intrinsics(K) or set_intrinsics(fx, fy, cx, cy)."""
        if "K" in kw or len(args) == 1:
            K = np.asarray(kw.get("K", args[0] if args else None), np.float64)
            vals = (K[0, 0], K[1, 1], K[0, 2], K[1, 2])
        else:
            vals = tuple(dict(zip(("fx", "fy", "cx", "cy"), args), **kw)[k] for k in ("fx", "fy", "cx", "cy"))
        self.intr = np.array(vals, dtype=np.float32)
        self._cache = None

    # ---- getters ----
    def get_width(self):
        return self.width

    def get_height(self):
        return self.height

    def get_fx(self):
        return float(self.intr[0])

    def get_fy(self):
        return float(self.intr[1])

    def get_cx(self):
        return float(self.intr[2])

    def get_cy(self):
        return float(self.intr[3])

    def get_K(self):
        fx, fy, cx, cy = (float(x) for x in self.intr)
        return np.array([[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]])

    def get_T_cw(self):
        T = np.eye(4)
        T[:3, :] = self.T_cw[:3, :]
        return T

    def get_R_cw(self):
        return self.T_cw[:3, :3].copy()

    def get_t_cw(self):
        return self.T_cw[:3, 3].copy()

    def get_orientation_cw(self):
        """-> Quaterniond of R_cw (Eigen::Quaterniond(R_cw_), camera_frustrum.cpp:152-154)."""
        return Quaterniond(_quat_from_matrix(self.T_cw[:3, :3]))

    def is_cache_valid(self):
        return self._cache is not None

    def _update_cache(self):
        if self._cache is not None:
            return self._cache
        fx, fy, cx, cy = (float(x) for x in self.intr)
        R_cw, t_cw = self.T_cw[:3, :3], self.T_cw[:3, 3]
        R_wc = R_cw.T
        t_wc = _mat3_vec(-R_wc, t_cw)
        corners = []  # camera_frustrum.cpp:209-245: near and far point of top-left, top-right, bottom-right, bottom-left
        for u, v in ((0.0, 0.0), (float(self.width), 0.0), (float(self.width), float(self.height)), (0.0, float(self.height))):
            xn, yn = (u - cx) / fx, (v - cy) / fy
            for d in (self.depth_min, self.depth_max):
                corners.append(_mat3_vec(R_wc, np.array([xn * d, yn * d, d])) + t_wc)
        corners = np.array(corners)
        bbox = BoundingBox3D(corners.min(axis=0), corners.max(axis=0))  # :247-264
        cam = np.array([_mat3_vec(R_cw, c) + t_cw for c in corners])  # :266-301: the box of the corners in the camera frame
        lo, hi = cam.min(axis=0), cam.max(axis=0)
        center_w = _mat3_vec(R_wc, (lo + hi) / 2.0) + t_wc
        self._cache = (corners, bbox, (center_w, _quat_from_matrix(R_wc), hi - lo))
        return self._cache

    def get_corners(self):
        return [c.copy() for c in self._update_cache()[0]]

    def get_bbox(self):
        return self._update_cache()[1]

    def get_obb(self):
        from .volumetric_semantic import OrientedBoundingBox3D

        return OrientedBoundingBox3D(*self._update_cache()[2])

    def is_in_bbox(self, point_w):
        return self.get_bbox().contains(np.asarray(point_w, np.float64))

    def is_in_obb(self, point_w):
        return self.get_obb().contains(np.asarray(point_w, np.float64))

    def contains(self, point_w):
        """-> (inside, ImagePoint): camera_frustrum.cpp:175-196 (the depth test first - ImagePoint(-1, -1, -1) when it fails -, then
        the projection in float64 narrowed to float32 pixel coordinates, then the image bounds)."""
        p = np.asarray(point_w, np.float64)
        pc = _mat3_vec(self.T_cw[:3, :3], p) + self.T_cw[:3, 3]
        depth = np.float32(pc[2])
        if not (depth >= np.float32(self.depth_min) and depth <= np.float32(self.depth_max)):
            return False, ImagePoint(-1.0, -1.0, -1.0)
        fx, fy, cx, cy = (float(x) for x in self.intr)
        with np.errstate(divide="ignore", invalid="ignore"):
            u = np.float32(fx * (pc[0] / pc[2]) + cx)
            v = np.float32(fy * (pc[1] / pc[2]) + cy)
        inside = bool(u >= np.float32(0.0) and u < np.float32(self.width) and v >= np.float32(0.0) and v < np.float32(self.height))
        return inside, ImagePoint(u, v, depth)


class VoxelGridData:
    """cpp/volumetric/voxel_grid_data.h:36-50: .points/.colors (+ empty semantic fields); default-constructible like the binding's."""

    def __init__(self, points=None, colors=None):
        self.points = np.zeros((0, 3), np.float32) if points is None else points
        self.colors = np.zeros((0, 3), np.float32) if colors is None else colors
        self.class_ids = np.zeros((0,), np.int32)
        self.object_ids = np.zeros((0,), np.int32)
        self.confidences = np.zeros((0,), np.float32)


class VoxelData:
    """``volumetric.VoxelData`` (volumetric_grid_module.h:943-947; cpp/volumetric/voxel_data.h:118-139): the value type of one voxel as
    the module hands it to Python - ``count`` (read / write), ``get_position()``, ``get_color()`` = sum / count in float32 (0 / 0 = nan for
    an empty voxel: the release build has no zero-count check, voxel_data.h:31-37).  A host-side value class as in the reference; the
    grids keep their voxels in HBM (HvVoxel) and return rows, not objects."""

    _pos_dtype = np.float32

    def __init__(self):
        self.count = 0
        self.position_sum = np.zeros(3, self._pos_dtype)
        self.color_sum = np.zeros(3, np.float32)

    def get_position(self):
        with np.errstate(invalid="ignore", divide="ignore"):
            return list(self.position_sum / self._pos_dtype(self.count))

    def get_color(self):
        with np.errstate(invalid="ignore", divide="ignore"):
            return list(self.color_sum / np.float32(self.count))


class VoxelBlockGrid(_Volume):
    """``volumetric.VoxelBlockGrid(voxel_size, block_size=8)`` on the GPU.

    integrate() results (count, position_sum, color_sum per voxel) are bit-identical to the
    reference's sequential accumulation; row order of get_voxels() differs (the reference's is its
    unordered_map iteration order), so compare outputs as sets.
    """

    def __init__(self, voxel_size, block_size=8, device=0, max_blocks=None, max_points=None):
        voxel_size = float(np.float32(voxel_size))  # pybind narrows to float (py::init<float,int>)
        super().__init__(L.HV_MODE_VOXEL_GRID, voxel_size, 0.0, block_size, 1, device, max_blocks, max_points)
        self.voxel_size = voxel_size
        self.block_size = int(block_size)

    # -- integrate -------------------------------------------------------------------------------
    def integrate(self, points, colors=None):
        """points: [N,3] float32 or float64 (the binding's two overloads, volumetric_grid_module.h:738-749: float64 points
        are keyed in double, anything else goes through float32); colors: [N,3] uint8|float32|None."""
        is_torch = hasattr(points, "data_ptr")
        if is_torch:
            if points.dim() != 2 or points.shape[1] != 3:
                raise RuntimeError("points must be a contiguous Nx3 array")
            wide = str(points.dtype) == "torch.float64"
            pts = points.contiguous() if wide else points.contiguous().float()
            n = pts.shape[0]
        else:
            pts = np.asarray(points)
            if pts.ndim != 2 or pts.shape[1] != 3:
                raise RuntimeError("points must be a contiguous Nx3 array")
            wide = pts.dtype == np.float64
            pts = np.ascontiguousarray(pts, dtype=np.float64 if wide else np.float32)
            n = pts.shape[0]
        if n == 0:
            return
        kind, cols = L.HV_COLOR_NONE, None
        if colors is not None:
            if hasattr(colors, "data_ptr"):
                cols = colors.contiguous()
                shape, dt = tuple(cols.shape), str(cols.dtype)
                is_u8, is_f32 = dt == "torch.uint8", dt == "torch.float32"
            else:
                cols = np.ascontiguousarray(colors)
                shape, dt = cols.shape, str(cols.dtype)
                is_u8, is_f32 = cols.dtype == np.uint8, cols.dtype == np.float32
            if len(shape) != 2 or shape[1] != 3:
                raise RuntimeError("colors must be a contiguous Nx3 array")
            if shape[0] != n:
                raise RuntimeError("points and colors must have the same size")
            if is_u8:
                kind = L.HV_COLOR_U8
            elif is_f32:
                kind = L.HV_COLOR_F32
            else:
                raise RuntimeError(f"Colors must be uint8 or float32, got dtype with {dt}")
            if L.location(cols) != L.location(pts):
                raise RuntimeError("points and colors must live on the same device")
        fn = self._lib.hv_integrate_points_f64 if wide else self._lib.hv_integrate_points
        L.check(fn(self._h, L.ptr(pts), n, L.ptr(cols), kind, L.location(pts)))

    def integrate_rgbd(self, depth, rgb, fx, fy, cx, cy, T_cw, max_depth=np.inf, min_depth=0.0, depth_scale=1.0):
        """Fused depth2pointcloud + world transform + integrate for one posed RGB-D frame
        (pyslam/utilities/depth.py:45-85, volumetric_integrator_voxel_grid.py:251-300)."""
        dkind = L.HV_DEPTH_U16 if str(depth.dtype) in ("uint16", "torch.uint16") else L.HV_DEPTH_F32
        if hasattr(depth, "data_ptr"):  # torch: the kernels read packed f32 / u16 depth and packed u8 colour
            if str(depth.dtype) not in ("torch.float32", "torch.uint16") or not depth.is_contiguous() or not rgb.is_contiguous() \
                    or str(rgb.dtype) != "torch.uint8":
                raise RuntimeError("integrate_rgbd: device inputs must be contiguous float32|uint16 depth and uint8 colour")
        else:  # host arrays: float64 / strided inputs are converted, not silently misread
            depth = np.ascontiguousarray(depth, dtype=np.uint16 if dkind == L.HV_DEPTH_U16 else np.float32)
            rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        H, W = int(depth.shape[0]), int(depth.shape[1])
        if tuple(rgb.shape) != (H, W, 3):
            raise RuntimeError(f"integrate_rgbd: colour image {tuple(rgb.shape)} does not match depth {(H, W)}")
        intr = np.array([fx, fy, cx, cy], dtype=np.float64)
        T = _as_f64_4x4(T_cw)
        maxd = float(min(max_depth, 3.0e38))
        L.check(
            self._lib.hv_integrate_rgbd_points(
                self._h, L.ptr(depth), dkind, float(depth_scale), L.ptr(rgb), H, W, L.ptr(intr), L.ptr(T),
                float(min_depth), maxd, L.location(depth)
            )
        )

    def integrate_rgbd_batch(self, depth, rgb, fx, fy, cx, cy, T_cw, max_depth=np.inf, min_depth=0.0, depth_scale=1.0):
        """Replay F posed frames ([F,H,W] depth, [F,H,W,3] rgb, [F,4,4] T_cw): bit-identical to F integrate_rgbd()
        calls, with one device sort per max_points / (H*W) frames (create the grid with a large max_points)."""
        dkind = L.HV_DEPTH_U16 if str(depth.dtype) in ("uint16", "torch.uint16") else L.HV_DEPTH_F32
        F, H, W = (int(x) for x in depth.shape)
        if not hasattr(depth, "data_ptr"):
            depth = np.ascontiguousarray(depth, dtype=np.uint16 if dkind == L.HV_DEPTH_U16 else np.float32)
            rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        intr = np.array([fx, fy, cx, cy], dtype=np.float64)
        T = np.ascontiguousarray(np.asarray(T_cw, dtype=np.float64).reshape(F, 16))
        L.check(self._lib.hv_integrate_rgbd_points_batch(self._h, L.ptr(depth), dkind, float(depth_scale), L.ptr(rgb), F, H, W,
                                                         L.ptr(intr), L.ptr(T), float(min_depth), float(min(max_depth, 3.0e38)),
                                                         L.location(depth)))

    # -- queries ---------------------------------------------------------------------------------
    def _collect(self, call):
        n = ctypes.c_int64()
        L.check(call(None, None, 0, ctypes.byref(n)))
        pts = np.zeros((n.value, 3), np.float32)
        cols = np.zeros((n.value, 3), np.float32)
        if n.value:
            L.check(call(L.ptr(pts), L.ptr(cols), n.value, ctypes.byref(n)))
        return VoxelGridData(pts, cols)

    def set_owner(self, rank, world_size):
        """Multi-GPU block ownership: fuse only the blocks with hash(block key) % world_size == rank (no collective while fusing)."""
        L.check(self._lib.hv_set_owner(self._h, int(rank), int(world_size)))

    def get_voxels(self, min_count=1, min_confidence=0.0):
        return self._collect(
            lambda p, c, cap, n: self._lib.hv_get_voxels(self._h, int(min_count), float(min_confidence), p, c, cap, n, L.HV_HOST)
        )

    def get_points(self):
        return self.get_voxels(1, 0.0).points

    def get_colors(self):
        return self.get_voxels(1, 0.0).colors

    def get_voxels_in_bb(self, bbox, min_count=1, min_confidence=0.0, include_semantics=False):
        bb = bbox.as_array() if isinstance(bbox, BoundingBox3D) else np.ascontiguousarray(bbox, dtype=np.float64)
        return self._collect(
            lambda p, c, cap, n: self._lib.hv_get_voxels_in_bb(
                self._h, L.ptr(bb), int(min_count), float(min_confidence), p, c, cap, n, L.HV_HOST
            )
        )

    def get_voxels_in_camera_frustrum(self, camera_frustrum, min_count=1, min_confidence=0.0, include_semantics=False):
        f = camera_frustrum
        return self._collect(
            lambda p, c, cap, n: self._lib.hv_get_voxels_in_frustum(
                self._h, L.ptr(f.intr), f.width, f.height, L.ptr(f.T_cw), f.depth_max, f.depth_min,
                int(min_count), float(min_confidence), p, c, cap, n, L.HV_HOST
            )
        )

    def carve(self, camera_frustrum, depth_image, depth_threshold=1e-2):
        self._carve(camera_frustrum, depth_image, depth_threshold)

    def remove_low_count_voxels(self, min_count):
        L.check(self._lib.hv_remove_low_count_voxels(self._h, int(min_count)))

    def remove_low_confidence_voxels(self, min_confidence):
        return  # no-op for non-semantic voxels (voxel_block_grid.hpp:650-676)

    def clear(self):
        L.check(self._lib.hv_reset(self._h))

    reset = clear

    def size(self):
        n = ctypes.c_int64()
        L.check(self._lib.hv_size(self._h, ctypes.byref(n)))
        return n.value

    get_total_voxel_count = size

    def empty(self):
        return self.num_blocks() == 0

    def get_block_size(self):
        return self.block_size

    # -- parity/debug ----------------------------------------------------------------------------
    def dump(self):
        """-> keys [B,3] i32, hashes [B] u64, counts [B,bs^3] i32, sums [B,bs^3,6] f32, key-sorted."""
        nb = self.num_blocks()
        nv = self.block_size ** 3
        keys = np.zeros((nb, 3), np.int32)
        hashes = np.zeros(nb, np.uint64)
        counts = np.zeros((nb, nv), np.int32)
        sums = np.zeros((nb, nv, 6), np.float32)
        n = ctypes.c_int64()
        L.check(self._lib.hv_dump_blocks(self._h, L.ptr(keys), L.ptr(hashes), L.ptr(counts), L.ptr(sums), ctypes.byref(n)))
        return keys, hashes, counts, sums

    def keys_from_points(self, points):
        pts = np.ascontiguousarray(points, dtype=np.float32)
        n = pts.shape[0]
        vk = np.zeros((n, 3), np.int32)
        bk = np.zeros((n, 3), np.int32)
        lk = np.zeros((n, 3), np.int32)
        h = np.zeros(n, np.uint64)
        L.check(self._lib.hv_keys_from_points(self._h, L.ptr(pts), n, L.ptr(vk), L.ptr(bk), L.ptr(lk), L.ptr(h)))
        return vk, bk, lk, h


class VoxelGrid(VoxelBlockGrid):
    """``volumetric.VoxelGrid(voxel_size)`` — the reference's *direct* voxel hash (cpp/volumetric/voxel_grid.h:83-245;
    selected only with kVolumetricIntegrationUseVoxelBlocks=False).  Its observable results (per-voxel sums in
    point-index order, get_voxels / queries / carve) are those of the block grid, so on the GPU it is the block hash
    behind the direct grid's constructor.  Two reference quirks of this non-default path: uint8 colours are dropped
    by its scalar branch (HasColors = is_same<Tc, float>, voxel_grid.hpp:493-498) — mirrored here; with float32
    points + float32 colours an AVX2/SSE build accumulates batches of 4 in double (voxel_grid_simd.hpp) — not
    mirrored (the non-SIMD build, which is what the oracle compiles, matches bit for bit)."""

    def __init__(self, voxel_size=0.05, device=0, max_blocks=None, max_points=None):
        super().__init__(voxel_size, 8, device=device, max_blocks=max_blocks, max_points=max_points)

    def integrate(self, points, colors=None):
        if colors is not None and getattr(colors, "dtype", None) == np.uint8:
            colors = None
        return super().integrate(points, colors)


class TBBUtils:
    """`volumetric.TBBUtils` exists only so callers' thread-cap call keeps working; the GPU path has
    no CPU worker threads (cpp/volumetric/tbb_utils.h:29-58)."""

    _max_threads = None  # what the caller set last (nothing on the GPU path depends on it)

    @staticmethod
    def set_max_threads(num_threads):
        """-> the number of threads set (tbb_utils.h:37-48: a value <= 0 asks for the default, all hardware threads)."""
        import os

        TBBUtils._max_threads = int(num_threads) if int(num_threads) > 0 else (os.cpu_count() or 1)
        return TBBUtils._max_threads

    @staticmethod
    def get_max_threads():
        """tbb_utils.h:51-53: the cap in force (the host's hardware threads until one is set)."""
        import os

        return TBBUtils._max_threads if TBBUtils._max_threads is not None else (os.cpu_count() or 1)


# ================================================================================================
# open3d slice mirror (TSDF)
# ================================================================================================
class PinholeCameraIntrinsic:
    """o3d.camera.PinholeCameraIntrinsic(width, height, fx, fy, cx, cy)."""

    def __init__(self, width, height, fx, fy, cx, cy):
        self.width = int(width)
        self.height = int(height)
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)

    def as_array(self):
        return np.array([self.fx, self.fy, self.cx, self.cy], dtype=np.float64)


class RGBDImage:
    """o3d.geometry.RGBDImage.create_from_color_and_depth(color, depth, depth_scale, depth_trunc,
    convert_rgb_to_intensity=False): the scale/trunc conversion itself runs on the GPU inside
    integrate(); this object only carries the operands."""

    def __init__(self, color, depth, depth_scale=1000.0, depth_trunc=3.0):
        self.color = color
        self.depth = depth
        self.depth_scale = float(depth_scale)
        self.depth_trunc = float(depth_trunc)

    @staticmethod
    def create_from_color_and_depth(color, depth, depth_scale=1000.0, depth_trunc=3.0, convert_rgb_to_intensity=True):
        if convert_rgb_to_intensity:
            raise RuntimeError("[ScalableTSDFVolume::Integrate] Unsupported image format.")
        return RGBDImage(color, depth, depth_scale, depth_trunc)


class TriangleMesh:
    def __init__(self, vertices, triangles, vertex_colors):
        self.vertices = vertices
        self.triangles = triangles
        self.vertex_colors = vertex_colors
        self.vertex_normals = np.zeros((0, 3), np.float64)  # compute_vertex_normals() is not called by pySLAM


class PointCloud:
    def __init__(self, points, colors, normals=None):
        self.points = points
        self.colors = colors
        self.normals = normals  # [N,3] f64 when asked for (Open3D's point cloud always carries them), else None

    def has_normals(self):
        return self.normals is not None and len(self.normals) == len(self.points)


class ScalableTSDFVolume(_Volume):
    """o3d.pipelines.integration.ScalableTSDFVolume(voxel_length, sdf_trunc, color_type=RGB8) on
    the GPU (volume_unit_resolution=16, depth_sampling_stride=4 as in Open3D)."""

    def __init__(self, voxel_length, sdf_trunc, color_type=None, volume_unit_resolution=16,
                 depth_sampling_stride=4, device=0, max_blocks=None, max_points=None):
        super().__init__(L.HV_MODE_TSDF, voxel_length, sdf_trunc, volume_unit_resolution, depth_sampling_stride,
                         device, max_blocks, max_points)
        self.voxel_length = float(voxel_length)
        self.sdf_trunc = float(sdf_trunc)
        self.res = int(volume_unit_resolution)

    def reset(self):
        L.check(self._lib.hv_reset(self._h))

    def integrate(self, image, intrinsic, extrinsic):
        """image: RGBDImage (color HxWx3 uint8 RGB, depth HxW float32|uint16); extrinsic = T_cw."""
        depth, color = image.depth, image.color
        dkind = L.HV_DEPTH_U16 if str(depth.dtype) in ("uint16", "torch.uint16") else L.HV_DEPTH_F32
        if not hasattr(depth, "data_ptr"):
            depth = np.ascontiguousarray(depth, dtype=np.uint16 if dkind == L.HV_DEPTH_U16 else np.float32)
            color = np.ascontiguousarray(color, dtype=np.uint8)
        H, W = int(depth.shape[0]), int(depth.shape[1])
        if tuple(color.shape) != (H, W, 3) or intrinsic.width != W or intrinsic.height != H:
            raise RuntimeError("[ScalableTSDFVolume::Integrate] Unsupported image format.")
        if L.location(depth) != L.location(color):
            raise RuntimeError("depth and color must live on the same device")
        intr = intrinsic.as_array()
        T = _as_f64_4x4(extrinsic)
        # the kernels gather from the caller's planes asynchronously on the volume's stream: keep device inputs
        # alive until the next call (by then the stream has consumed them or they are still referenced here)
        self._inflight = (getattr(self, "_inflight_prev", None), depth, color)
        self._inflight_prev = (depth, color)
        L.check(
            self._lib.hv_tsdf_integrate(
                self._h, L.ptr(depth), dkind, L.ptr(color), H, W, L.ptr(intr), L.ptr(T), image.depth_scale,
                image.depth_trunc, L.location(depth)
            )
        )

    def integrate_batch(self, depth, color, intrinsic, extrinsics, depth_scale=1.0, depth_trunc=4.0):
        """Replay F posed frames ([F,H,W] depth, [F,H,W,3] colour, [F,4,4] T_cw); same result as F
        integrate() calls (the rebuild() use case, volumetric_integrator_base.py:1242-1318)."""
        dkind = L.HV_DEPTH_U16 if str(depth.dtype) in ("uint16", "torch.uint16") else L.HV_DEPTH_F32
        F, H, W = (int(s) for s in depth.shape)
        T = np.ascontiguousarray(np.asarray(extrinsics, dtype=np.float64).reshape(F, 16))
        intr = intrinsic.as_array()
        self._inflight = (getattr(self, "_inflight_prev", None), depth, color)
        self._inflight_prev = (depth, color)
        L.check(
            self._lib.hv_tsdf_integrate_batch(
                self._h, L.ptr(depth), dkind, L.ptr(color), F, H, W, L.ptr(intr), L.ptr(T), float(depth_scale),
                float(depth_trunc), L.location(depth)
            )
        )

    def integrate_frames(self, depths, colors, intrinsic, extrinsics, depth_scale=1.0, depth_trunc=4.0):
        """integrate_batch for HOST frames held one numpy array per frame (what the integrator worker has after draining
        its queue): no np.stack - the library copies every frame straight into page-locked staging slots and sends them
        over PCIe on a copy stream while the previous batch is swept.  Same result as len(depths) integrate() calls."""
        F = len(depths)
        if F == 0:
            return
        dkind = L.HV_DEPTH_U16 if str(depths[0].dtype) == "uint16" else L.HV_DEPTH_F32
        dtype = np.uint16 if dkind == L.HV_DEPTH_U16 else np.float32
        depths = [np.ascontiguousarray(d, dtype=dtype) for d in depths]
        colors = [np.ascontiguousarray(c, dtype=np.uint8) for c in colors]
        H, W = (int(x) for x in depths[0].shape)
        for d, c in zip(depths, colors):
            if d.shape != (H, W) or c.shape != (H, W, 3):
                raise RuntimeError("[ScalableTSDFVolume::Integrate] Unsupported image format.")
        T = np.ascontiguousarray(np.asarray(extrinsics, dtype=np.float64).reshape(F, 16))
        intr = intrinsic.as_array()
        dp = (ctypes.c_void_p * F)(*[d.ctypes.data for d in depths])
        cp = (ctypes.c_void_p * F)(*[c.ctypes.data for c in colors])
        L.check(self._lib.hv_tsdf_integrate_frames(self._h, dp, dkind, cp, F, H, W, L.ptr(intr), L.ptr(T), float(depth_scale),
                                                   float(depth_trunc)))

    def set_color_order(self, bgr=False):
        """Colour frames handed to integrate* are R, G, B (Open3D's order, default) or B, G, R (OpenCV's: pySLAM's keyframe.img)."""
        L.check(self._lib.hv_tsdf_set_color_order(self._h, 1 if bgr else 0))

    def set_tile(self, u0, v0, u1, v1):
        """Restrict fusion to the image tile [u0,u1) x [v0,v1) (multi-GPU sharding); zeros = whole image."""
        L.check(self._lib.hv_tsdf_set_tile(self._h, int(u0), int(v0), int(u1), int(v1)))

    def set_rectify_maps(self, map_x, map_y):
        """Undistort / rectify on the device: every frame handed to integrate / integrate_batch / integrate_frames afterwards goes
        through the maps first (colour bilinear, depth nearest - the reference's per-keyframe cv2.remap pair,
        volumetric_integrator_base.py:1017-1043), one launch per batch; the caller passes the RECTIFIED intrinsics.  None clears."""
        if map_x is None or map_y is None:
            L.check(self._lib.hv_tsdf_set_rectify_maps(self._h, None, None, 0, 0, L.HV_HOST))
            return
        mx, my = np.ascontiguousarray(map_x, dtype=np.float32), np.ascontiguousarray(map_y, dtype=np.float32)
        assert mx.ndim == 2 and mx.shape == my.shape
        L.check(self._lib.hv_tsdf_set_rectify_maps(self._h, L.ptr(mx), L.ptr(my), int(mx.shape[0]), int(mx.shape[1]), L.HV_HOST))

    def set_owner(self, rank, world_size):
        """Fuse/store only the units owned by `rank` of `world_size` (multi-GPU unit-ownership sharding)."""
        L.check(self._lib.hv_tsdf_set_owner(self._h, int(rank), int(world_size)))

    @staticmethod
    def _out_dtype(dtype):
        dt = np.dtype(np.float64 if dtype is None else dtype)
        if dt not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise TypeError(f"extraction dtype must be float64 (Open3D's) or float32, got {dt}")
        return dt

    def extract_triangle_mesh(self, device=False, dtype=None):
        """o3d's extract_triangle_mesh().  device=True: vertices / vertex_colors / triangles are torch CUDA tensors on the volume's GPU
        (nothing crosses PCIe: for consumers that render or post-process on the GPU); default: host arrays like Open3D's.
        dtype=np.float32: vertices / vertex_colors as float32 - Open3D's float64 values rounded once on the device (what
        Parameters.kDenseMappingDtypeVertices / Colors name and pySLAM's viewer casts to, config_parameters.py:290-291): a third fewer
        bytes per output tick.  Default float64 = Open3D's arrays."""
        dt = self._out_dtype(dtype)
        fn = self._lib.hv_tsdf_extract_mesh if dt == np.float64 else self._lib.hv_tsdf_extract_mesh_f32
        nv, nt = ctypes.c_int64(), ctypes.c_int64()
        L.check(fn(self._h, None, None, 0, None, 0, ctypes.byref(nv), ctypes.byref(nt)))
        if device:
            import torch

            dev = torch.device("cuda", int(self._cfg.device))
            tdt = torch.float64 if dt == np.float64 else torch.float32
            verts = torch.empty((nv.value, 3), dtype=tdt, device=dev)
            cols = torch.empty((nv.value, 3), dtype=tdt, device=dev)
            tris = torch.empty((nt.value, 3), dtype=torch.int32, device=dev)
            torch.cuda.current_stream(dev).synchronize()  # (the allocator may hand out blocks with work pending on torch's stream)
        else:
            verts = _result_array((nv.value, 3), dt)
            cols = _result_array((nv.value, 3), dt)
            tris = _result_array((nt.value, 3), np.int32)
        if nv.value or nt.value:
            L.check(fn(self._h, L.ptr(verts), L.ptr(cols), nv.value, L.ptr(tris), nt.value, ctypes.byref(nv), ctypes.byref(nt)))
        return TriangleMesh(verts, tris, cols)

    def extract_point_cloud(self, normals=False, device=False, dtype=None):
        """o3d's extract_point_cloud().  normals=True also computes the per-point normals Open3D attaches (GetNormalAt: the
        gradient of the trilinearly interpolated tsdf) - pySLAM's viewer path does not read them, its save path writes them.
        device=True: torch CUDA tensors on the volume's GPU instead of host arrays.  dtype=np.float32: points / colors as float32
        (see extract_triangle_mesh; the normals stay float64, taken at the float64 points)."""
        dt = self._out_dtype(dtype)
        fn = self._lib.hv_tsdf_extract_points if dt == np.float64 else self._lib.hv_tsdf_extract_points_f32
        n = ctypes.c_int64()
        L.check(fn(self._h, None, None, 0, ctypes.byref(n)))
        if device:
            import torch

            dev = torch.device("cuda", int(self._cfg.device))
            tdt = torch.float64 if dt == np.float64 else torch.float32
            pts = torch.empty((n.value, 3), dtype=tdt, device=dev)
            cols = torch.empty((n.value, 3), dtype=tdt, device=dev)
            torch.cuda.current_stream(dev).synchronize()
        else:
            pts = _result_array((n.value, 3), dt)
            cols = _result_array((n.value, 3), dt)
        nrm = None
        if n.value:
            L.check(fn(self._h, L.ptr(pts), L.ptr(cols), n.value, ctypes.byref(n)))
        if normals:
            nrm = torch.zeros((n.value, 3), dtype=torch.float64, device=dev) if device else np.zeros((n.value, 3), np.float64)
            if n.value:
                L.check(self._lib.hv_tsdf_extract_point_normals(self._h, L.ptr(nrm), n.value, ctypes.byref(n)))
        return PointCloud(pts, cols, nrm)

    # -- parity/debug + multi-GPU ------------------------------------------------------------------
    def dump(self):
        """-> keys [U,3], tsdf [U,R^3] f32, weight [U,R^3] f32, color [U,R^3,3] f64 (0..255), key-sorted,
        voxel order x*R^2 + y*R + z (Open3D IndexOf)."""
        nu = self.num_blocks()
        nv = self.res ** 3
        keys = np.zeros((nu, 3), np.int32)
        tsdf = np.zeros((nu, nv), np.float32)
        weight = np.zeros((nu, nv), np.float32)
        color = np.zeros((nu, nv, 3), np.float64)
        n = ctypes.c_int64()
        L.check(self._lib.hv_tsdf_dump(self._h, L.ptr(keys), L.ptr(tsdf), L.ptr(weight), L.ptr(color), ctypes.byref(n)))
        return keys, tsdf, weight, color

    def touched_keys(self):
        n = ctypes.c_int64()
        L.check(self._lib.hv_tsdf_touched(self._h, None, 0, ctypes.byref(n)))
        keys = np.zeros((n.value, 3), np.int32)
        if n.value:
            L.check(self._lib.hv_tsdf_touched(self._h, L.ptr(keys), n.value, ctypes.byref(n)))
        return keys

    def unit_keys(self):
        n = ctypes.c_int64()
        L.check(self._lib.hv_tsdf_unit_keys(self._h, None, 0, ctypes.byref(n)))
        keys = np.zeros((n.value, 3), np.int32)
        if n.value:
            L.check(self._lib.hv_tsdf_unit_keys(self._h, L.ptr(keys), n.value, ctypes.byref(n)))
        return keys

    def dirty_keys(self):
        """Units this GPU stamped (i.e. may have updated) since the last mark_merged(), sorted [K,3] int32."""
        n = ctypes.c_int64()
        L.check(self._lib.hv_tsdf_dirty_keys(self._h, None, 0, ctypes.byref(n)))
        keys = np.zeros((n.value, 3), np.int32)
        if n.value:
            L.check(self._lib.hv_tsdf_dirty_keys(self._h, L.ptr(keys), n.value, ctypes.byref(n)))
        return keys

    def mark_merged(self):
        L.check(self._lib.hv_tsdf_mark_merged(self._h))

    # -- the halo merge with lists and plan in device memory (hv_halo.hip; distributed.ShardedTSDF._merge_halo_device) ------------
    def halo_lists_device(self, dirty, held):
        """This volume's dirty and held unit keys (packed 64-bit words) into the torch CUDA int64 tensors `dirty` / `held`
        (each at least num_blocks() long).  -> (n_dirty, n_held)."""
        nd, nh = ctypes.c_int64(), ctypes.c_int64()
        L.check(self._lib.hv_merge_halo_lists_device(self._h, L.ptr(dirty), int(dirty.numel()), L.ptr(held), int(held.numel()),
                                                     ctypes.byref(nd), ctypes.byref(nh)))
        return nd.value, nh.value

    def halo_plan_device(self, dirty_all, dirty_counts, held_all, held_counts, world_size, rank, all_dirty_kept=False):
        """The gathered lists ([world, stride] CUDA int64 tensors; counts: host int64 [world]) -> the merge plan, left in the volume.
        -> number of shared units."""
        n = ctypes.c_int64()
        dc, hc = np.ascontiguousarray(dirty_counts, dtype=np.int64), np.ascontiguousarray(held_counts, dtype=np.int64)
        L.check(self._lib.hv_merge_halo_plan_device(self._h, L.ptr(dirty_all), L.ptr(dc), int(dirty_all.shape[1]), L.ptr(held_all), L.ptr(hc),
                                                    int(held_all.shape[1]), int(world_size), int(rank), int(bool(all_dirty_kept)), ctypes.byref(n)))
        return n.value

    def halo_plan_fetch(self):
        """-> (shared_keys [K,3] int32, action [K] uint8) of the stored plan (host arrays; inspection and tests)."""
        n = ctypes.c_int64()
        L.check(self._lib.hv_merge_halo_plan_fetch(self._h, None, None, 0, ctypes.byref(n)))
        keys, action = np.zeros((n.value, 3), np.int32), np.zeros(n.value, np.uint8)
        if n.value:
            L.check(self._lib.hv_merge_halo_plan_fetch(self._h, L.ptr(keys), L.ptr(action), n.value, ctypes.byref(n)))
        return keys, action

    def halo_pack_planned(self, first, count, payload):
        L.check(self._lib.hv_merge_halo_pack_planned(self._h, int(first), int(count), L.ptr(payload)))

    def halo_unpack_planned(self, first, count, payload):
        L.check(self._lib.hv_merge_halo_unpack_planned(self._h, int(first), int(count), L.ptr(payload)))

    def halo_unpack(self, keys, payload, action):
        """hv_merge_halo_unpack: action[k] 0 = not held here, 1 = keep (state := payload), 2 = zero the unit."""
        keys = np.ascontiguousarray(keys, dtype=np.int32)
        action = np.ascontiguousarray(action, dtype=np.uint8)
        L.check(self._lib.hv_merge_halo_unpack(self._h, L.ptr(keys), keys.shape[0], L.ptr(payload), L.ptr(action), L.location(payload)))

    def export_numerators(self, keys, out=None):
        """keys [K,3] int32 -> payload [K, R^3, 5] float32 {sum tsdf*w, w, sum r, sum g, sum b}."""
        keys = np.ascontiguousarray(keys, dtype=np.int32)
        k = keys.shape[0]
        if out is None:
            out = np.zeros((k, self.res ** 3, 5), np.float32)
        L.check(self._lib.hv_tsdf_export_numerators(self._h, L.ptr(keys), k, L.ptr(out), L.location(out)))
        return out

    def import_numerators(self, keys, payload):
        keys = np.ascontiguousarray(keys, dtype=np.int32)
        L.check(self._lib.hv_tsdf_import_numerators(self._h, L.ptr(keys), keys.shape[0], L.ptr(payload), L.location(payload)))
