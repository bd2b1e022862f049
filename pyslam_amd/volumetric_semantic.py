"""The semantic block grids of pySLAM's ``volumetric`` module on the GPU (bindings:
cpp/volumetric/volumetric_grid_module.h:939-1033; class cpp/volumetric/voxel_block_semantic_grid.h:57-121):

* ``VoxelBlockSemanticGrid``              — voting payload (voxel_data_semantic.h:106-202)
* ``VoxelBlockSemanticProbabilisticGrid`` — log-probability payload (voxel_data_semantic.h:249-672)

Methods: integrate(points, colors, class_ids, instance_ids, depths), get_voxels(min_count, min_confidence),
carve, assign_object_ids_to_instance_ids, get_object_segments, merge_segments, remove_segment,
remove_low_confidence_segments, remove_low_count_voxels, remove_low_confidence_voxels, get_ids,
set_depth_threshold, set_depth_decay_rate, clear/reset, size, num_blocks; module-level
``remap_instance_ids`` (image_utils.h:69-163) and the ``ObjectData`` / ``ObjectDataGroup`` /
``OrientedBoundingBox3D`` result types (voxel_grid_data.h:58-99, bounding_boxes_3d.h:82-131).
Also: get_voxels_in_bb / get_voxels_in_camera_frustrum (include_semantics), integrate_segment, get_class_segments.
The module's "*2" payloads (voxel_data_semantic2.h) are the *Grid2 classes at the end."""
import ctypes
import weakref
from collections.abc import Mapping

import numpy as np

from . import _lib as L
from .volumetric import VoxelData, VoxelGridData, _Volume


class OBBComputationMethod:
    """bounding_boxes_3d.h:28-31."""

    PCA = 0
    CONVEX_HULL_MINIMAL = 1


def _sat_intersects(ca, ha, Ra, cb, hb, Rb):
    """Separating-axis test of two boxes {centre, half extents, axes as columns}: bounding_boxes_3d.cpp:60-170 (the 3 + 3
    face axes and the 9 edge-edge axes of Gottschalk et al., the same 1e-9 padding of |R|)."""
    R = Ra.T @ Rb
    A = np.abs(R) + 1e-9
    tw = cb - ca
    t = Ra.T @ tw
    for i in range(3):
        if abs(t[i]) > ha[i] + hb @ A[i]:
            return False
    for j in range(3):
        if abs(tw @ Rb[:, j]) > ha @ A[:, j] + hb[j]:
            return False
    for i in range(3):
        i1, i2 = (i + 1) % 3, (i + 2) % 3
        for j in range(3):
            j1, j2 = (j + 1) % 3, (j + 2) % 3
            ra = ha[i1] * A[i2, j] + ha[i2] * A[i1, j]
            rb = hb[j1] * A[i, j2] + hb[j2] * A[i, j1]
            if abs(t[i2] * R[i1, j] - t[i1] * R[i2, j]) > ra + rb:
                return False
    return True


class OrientedBoundingBox3D:
    """bounding_boxes_3d.h:82-131, bounding_boxes_3d.cpp:253-345, bindings bounding_boxes_module.h:97-160: center (3,),
    orientation quaternion (w, x, y, z) object -> world, size (3,)."""

    def __init__(self, center=(0.0, 0.0, 0.0), orientation=(1.0, 0.0, 0.0, 0.0), size=(0.0, 0.0, 0.0)):
        self.center = np.asarray(center, np.float64).copy()
        self.orientation = np.asarray(orientation, np.float64).copy()
        self.size = np.asarray(size, np.float64).copy()

    def get_rotation_matrix(self):
        w, x, y, z = self.orientation / np.linalg.norm(self.orientation)
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    def get_matrix(self):
        """object -> world."""
        M = np.eye(4)
        M[:3, :3] = self.get_rotation_matrix()
        M[:3, 3] = self.center
        return M

    def get_inverse_matrix(self):
        """world -> object."""
        M = np.eye(4)
        Rt = self.get_rotation_matrix().T
        M[:3, :3] = Rt
        M[:3, 3] = -Rt @ self.center
        return M

    def get_volume(self):
        return float(np.prod(self.size))

    def get_surface_area(self):
        sx, sy, sz = self.size
        return float(2.0 * (sx * sy + sx * sz + sy * sz))

    def get_diagonal_length(self):
        return float(np.sqrt(np.sum(self.size * self.size)))

    def get_corners(self):
        """bounding_boxes_3d.cpp:287-318 (same corner order)."""
        R, h = self.get_rotation_matrix(), self.size / 2.0
        signs = [(1, 1, -1), (-1, 1, -1), (-1, -1, -1), (1, -1, -1), (1, 1, 1), (-1, 1, 1), (-1, -1, 1), (1, -1, 1)]
        return np.array([self.center + R @ (h * np.array(s, np.float64)) for s in signs])

    def contains(self, points):
        """One point [3] -> bool; several [N,3] -> list of bool.  Closed box with the reference's 1e-10 slack
        (bounding_boxes_3d.cpp:320-337)."""
        p = np.asarray(points, np.float64)
        q = (p - self.center) @ self.get_rotation_matrix()  # rows: R^T (p - c)
        h = self.size / 2.0 + 1e-10
        m = np.all((q >= -h) & (q <= h), axis=-1)
        return bool(m) if p.ndim == 1 else [bool(x) for x in m]

    def intersects(self, other):
        """Against another OrientedBoundingBox3D or a BoundingBox3D (bounding_boxes_3d.cpp:339-345)."""
        if isinstance(other, OrientedBoundingBox3D):
            cb, hb, Rb = other.center, other.size / 2.0, other.get_rotation_matrix()
        else:
            cb, hb, Rb = other.get_center(), other.get_size() / 2.0, np.eye(3)
        return _sat_intersects(self.center, self.size / 2.0, self.get_rotation_matrix(), cb, hb, Rb)

    @staticmethod
    def compute_from_points(points, method=OBBComputationMethod.PCA):
        """OrientedBoundingBox3D::compute_from_points(points, PCA), bounding_boxes_3d.cpp:373-553.  The convex-hull variant
        needs Qhull in the reference too (QHULL_FOUND) and is not provided here."""
        if method != OBBComputationMethod.PCA:
            raise NotImplementedError("OBBComputationMethod.CONVEX_HULL_MINIMAL is not provided (PCA is the reference's default)")
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        obb = np.zeros(10, np.float64)
        L.check(L.load().hv_compute_obb_pca(L.ptr(pts), pts.shape[0], L.ptr(obb)))
        return OrientedBoundingBox3D(obb[0:3], obb[3:7], obb[7:10])


class ObjectData:
    """voxel_grid_data.h:58-75."""

    def __init__(self, points, colors, object_id, class_id, confidence_min, confidence_max, oriented_bounding_box):
        self.points, self.colors = points, colors
        self.object_id, self.class_id = int(object_id), int(class_id)
        self.confidence_min, self.confidence_max = float(confidence_min), float(confidence_max)
        self.oriented_bounding_box = oriented_bounding_box


class ClassData:
    """voxel_grid_data.h:107-120."""

    def __init__(self, points, colors, class_id, confidence_min, confidence_max):
        self.points, self.colors, self.class_id = points, colors, int(class_id)
        self.confidence_min, self.confidence_max = float(confidence_min), float(confidence_max)


class ObjectDataGroup:
    """voxel_grid_data.h:83-93: object_vector + the redundant class_ids / object_ids lists."""

    def __init__(self, object_vector):
        self.object_vector = list(object_vector)
        self.class_ids = np.array([o.class_id for o in self.object_vector], np.int32)
        self.object_ids = np.array([o.object_id for o in self.object_vector], np.int32)


class ClassDataGroup:
    """voxel_grid_data.h:131-139: class_vector + the redundant class_ids list (what get_class_segments returns).  Iterates, indexes and
    measures like its class_vector (rounds 3-5 returned the bare list)."""

    def __init__(self, class_vector=()):
        self.class_vector = list(class_vector)
        self.class_ids = np.array([c.class_id for c in self.class_vector], np.int32)

    def __iter__(self):
        return iter(self.class_vector)

    def __len__(self):
        return len(self.class_vector)

    def __getitem__(self, i):
        return self.class_vector[i]


def check_image_size(image, height, width, image_name="image"):
    """``volumetric.check_image_size`` (image_utils.h:30-41): False - with the reference's message - for an empty image or one whose
    rows x cols differ from the expected size."""
    a = np.asarray(image) if not hasattr(image, "shape") else image
    rows, cols = (int(a.shape[0]), int(a.shape[1])) if len(a.shape) >= 2 else (0, 0)
    if rows == 0 or cols == 0 or rows != int(height) or cols != int(width):
        print(f"check_image_size: {image_name} size: {rows}x{cols}\n\tExpected height: {height}\n\tExpected width: {width}\n"
              "\tImage size does not match expected size")
        return False
    return True


# OpenCV's depth codes (CV_8U .. CV_64F), the `expected_type` of convert_image_type_if_needed for single-channel images
_CV_DEPTH_DTYPES = {0: np.uint8, 1: np.int8, 2: np.uint16, 3: np.int16, 4: np.int32, 5: np.float32, 6: np.float64}


def convert_image_type_if_needed(image, expected_type, image_name="image"):
    """``volumetric.convert_image_type_if_needed`` (image_utils.h:43-59): the image itself when it is empty or already of the expected
    type, else a converted copy (cv::Mat::convertTo: saturating, round-to-nearest-even for float -> integer).  expected_type: a numpy
    dtype or one of OpenCV's single-channel type codes (CV_32S = 4 ...)."""
    a = np.asarray(image)
    if a.size == 0:
        return image
    dt = np.dtype(_CV_DEPTH_DTYPES[int(expected_type) & 7] if isinstance(expected_type, (int, np.integer)) else expected_type)
    if a.dtype == dt:
        return image
    print(f"check_image_type: {image_name} type: {a.dtype}\n\tConverting image to {dt}")
    if np.issubdtype(dt, np.integer):
        info = np.iinfo(dt)
        src = np.rint(a) if np.issubdtype(a.dtype, np.floating) else a
        return np.clip(src, info.min, info.max).astype(dt)
    return a.astype(dt)


def _i32_image(img, name):
    a = np.ascontiguousarray(img)
    if a.ndim != 2:
        raise RuntimeError(f"{name} must be single-channel")
    return np.ascontiguousarray(a, dtype=np.int32)  # convert_image_type_if_needed(..., CV_32S)


def _is_device(a):
    return a is not None and hasattr(a, "is_cuda") and bool(a.is_cuda)


def _device_images(*images, device=0):
    """The per-keyframe images of one call, all at ONE location: when any of them is a torch CUDA tensor the numpy ones are
    uploaded (torch's blocking copy), so that a caller can keep a keyframe's depth / label images in HBM across
    filter_shadow_points -> assign_object_ids_to_instance_ids -> remap_instance_ids -> integrate_rgbd instead of staging each
    of them again in every call.  The images must live on the volume's GPU (`device`: hv_config.device): a tensor of another GPU
    is an error, not a silent peer access.  -> (list of images, HV_DEVICE | HV_HOST)."""
    if not any(_is_device(a) for a in images):
        return list(images), L.HV_HOST
    import torch

    dev = torch.device("cuda", int(device))
    out = []
    for a in images:
        if a is None:
            out.append(None)
        elif _is_device(a):
            if a.device != dev:
                raise RuntimeError(f"image tensor lives on {a.device}, the volume on {dev}")
            out.append(a.contiguous())
        else:
            out.append(torch.from_numpy(np.ascontiguousarray(a)).to(dev))
    return out, L.HV_DEVICE


class LazyIdMap(Mapping):
    """The instance -> object map of an association whose result is still in device memory.  A read-only dict: the first access
    fetches it (one synchronisation); remap_instance_ids on the same volume uses the device copy without fetching."""

    def __init__(self, volume, serial):
        self._volume, self._serial, self._d = volume, serial, None

    def on_device_of(self, volume):
        """True while `volume` still holds this map as its last association."""
        return volume is self._volume and getattr(volume, "_assoc_serial", None) == self._serial

    def _get(self):
        if self._d is None:
            if not self.on_device_of(self._volume):
                raise RuntimeError("the id map of an earlier association was not read before the next one replaced it")
            v = self._volume
            cap = 1 << 16
            mi, mo = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
            n = ctypes.c_int64()
            L.check(v._lib.hv_assoc_map_fetch(v._h, L.ptr(mi), L.ptr(mo), cap, ctypes.byref(n)))
            m = min(n.value, cap)
            self._d = {int(k): int(o) for k, o in zip(mi[:m], mo[:m])}
        return self._d

    def __getitem__(self, k):
        return self._get()[k]

    def __iter__(self):
        return iter(self._get())

    def __len__(self):
        return len(self._get())

    def __repr__(self):
        return repr(self._get())


def _remap_with_last_map(instance_ids, id_map, volume):
    """remap_instance_ids with the device-resident map of the volume's last association (hv_remap_instance_ids_last)."""
    v = volume
    if _is_device(instance_ids):
        import torch

        img = instance_ids.contiguous()
        if img.dim() != 2:
            raise RuntimeError("Instance ids must be single-channel")
        if img.dtype != torch.int32:
            raise RuntimeError("Instance ids must be int32")
        out = torch.empty_like(img)
        ts = v._torch_in(img, out)
        L.check(v._lib.hv_remap_instance_ids_last(v._h, L.ptr(img), int(img.shape[0]), int(img.shape[1]), L.ptr(out), L.HV_DEVICE))
        v._torch_out(ts, img.device)
        return out
    img = np.ascontiguousarray(instance_ids)
    if img.size == 0:
        return img
    if img.ndim != 2:
        raise RuntimeError("Instance ids must be single-channel")
    if img.dtype in (np.int8, np.uint8, np.int16, np.uint16):
        # the binding's narrower instantiations (image_utils_module.h:66-88): looked up as int, written back in the image's own type
        # (an object id, and the invalid id -1, narrowed the way the C++ assignment does: modulo 2^bits)
        if len(instance_id_to_object_id) == 0:
            return img
        return remap_instance_ids(img.astype(np.int32), instance_id_to_object_id, volume).astype(img.dtype)
    if img.dtype != np.int32:
        raise RuntimeError("Unsupported instance id type")
    out = np.empty_like(img)
    L.check(v._lib.hv_remap_instance_ids_last(v._h, L.ptr(img), img.shape[0], img.shape[1], L.ptr(out), L.HV_HOST))
    return out


def remap_instance_ids(instance_ids, instance_id_to_object_id, volume=None):
    """volumetric.remap_instance_ids(image int32 HxW, map) (image_utils.h:69-163, binding image_utils_module.h):
    ids absent from the map become -1; an EMPTY map returns the image as it is (the binding's early return, image_utils_module.h:52-58 -
    the C++ template behind it would set every id to -1).  Runs on the GPU of ``volume`` (any volume).  A torch CUDA int32
    image stays on the device (the result is a CUDA tensor).  The map an association on ``volume`` just returned is used where it
    lies, in device memory."""
    if isinstance(instance_id_to_object_id, LazyIdMap) and volume is not None and instance_id_to_object_id.on_device_of(volume):
        return _remap_with_last_map(instance_ids, instance_id_to_object_id, volume)
    if _is_device(instance_ids):
        import torch

        img = instance_ids.contiguous()
        if img.dim() != 2:
            raise RuntimeError("Instance ids must be single-channel")
        if img.dtype != torch.int32:
            raise RuntimeError("Instance ids must be int32")
        if volume is None:
            volume = _scratch_volume()
        keys = np.fromiter(instance_id_to_object_id.keys(), np.int32, len(instance_id_to_object_id))
        vals = np.fromiter(instance_id_to_object_id.values(), np.int32, len(instance_id_to_object_id))
        out = torch.empty_like(img)
        ts = volume._torch_in(img, out)
        L.check(volume._lib.hv_remap_instance_ids(volume._h, L.ptr(img), int(img.shape[0]), int(img.shape[1]), L.ptr(keys), L.ptr(vals),
                                                  len(keys), L.ptr(out), L.HV_DEVICE))
        volume._torch_out(ts, img.device)
        return out
    img = np.ascontiguousarray(instance_ids)
    if img.size == 0:
        return img
    if img.ndim != 2:
        raise RuntimeError("Instance ids must be single-channel")
    if img.dtype in (np.int8, np.uint8, np.int16, np.uint16):
        # the binding's narrower instantiations (image_utils_module.h:66-88): looked up as int, written back in the image's own type
        # (an object id, and the invalid id -1, narrowed the way the C++ assignment does: modulo 2^bits)
        if len(instance_id_to_object_id) == 0:
            return img
        return remap_instance_ids(img.astype(np.int32), instance_id_to_object_id, volume).astype(img.dtype)
    if img.dtype != np.int32:
        raise RuntimeError("Unsupported instance id type")
    if volume is None:
        volume = _scratch_volume()
    keys = np.fromiter(instance_id_to_object_id.keys(), np.int32, len(instance_id_to_object_id))
    vals = np.fromiter(instance_id_to_object_id.values(), np.int32, len(instance_id_to_object_id))
    out = np.empty_like(img)
    L.check(volume._lib.hv_remap_instance_ids(volume._h, L.ptr(img), img.shape[0], img.shape[1], L.ptr(keys), L.ptr(vals),
                                              len(keys), L.ptr(out), L.HV_HOST))
    return out


_scratch = None


def _scratch_volume():
    global _scratch
    if _scratch is None:
        _scratch = VoxelBlockSemanticGrid(0.05, 8, max_blocks=64, max_points=1 << 12)
    return _scratch


class VoxelSemanticData(VoxelData):
    """``volumetric.VoxelSemanticData`` (volumetric_grid_module.h:952-966; voxel_data_semantic.h:106-202): VoxelData with float64
    position sums plus the voting label state - ``get_object_id()`` / ``get_class_id()`` (-1 when unlabelled), ``get_confidence()`` =
    min(1, counter / count) (0 for an empty voxel), ``get_confidence_counter()``.  A host-side value class as in the reference."""

    _pos_dtype = np.float64

    def __init__(self):
        super().__init__()
        self.object_id, self.class_id, self.confidence_counter = -1, -1, 0

    def get_object_id(self):
        return self.object_id

    def get_class_id(self):
        return self.class_id

    def get_confidence(self):
        if self.count == 0:
            return 0.0
        return float(min(np.float32(1.0), np.float32(self.confidence_counter) / np.float32(self.count)))

    def get_confidence_counter(self):
        return self.confidence_counter


class _SemanticGridBase(_Volume):
    _MODE = None

    def __init__(self, voxel_size=0.05, block_size=8, device=0, max_blocks=None, max_points=None):
        voxel_size = float(np.float32(voxel_size))
        super().__init__(self._MODE, voxel_size, 0.0, block_size, 1, device, max_blocks, max_points)
        self.voxel_size, self.block_size = voxel_size, int(block_size)

    def set_owner(self, rank, world_size):
        """Multi-GPU block ownership (hv_set_owner): this grid fuses and stores only the blocks owner(block key) == rank."""
        L.check(self._lib.hv_set_owner(self._h, int(rank), int(world_size)))

    def set_depth_threshold(self, depth_threshold):
        L.check(self._lib.hv_set_depth_threshold(self._h, float(depth_threshold)))

    def set_depth_decay_rate(self, depth_decay_rate):
        L.check(self._lib.hv_set_depth_decay_rate(self._h, float(depth_decay_rate)))

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

    def integrate_rgbd(self, depth, rgb, fx, fy, cx, cy, T_cw, class_ids_image=None, object_ids_image=None, max_depth=np.inf,
                       min_depth=0.0, use_depths=True):
        """Fused per-keyframe prep + integrate (hv_integrate_rgbd_semantic): depth f32 [H,W] metres, rgb u8 [H,W,3]
        (already RGB), label images i32 [H,W] or None, T_cw world->camera."""
        if any(_is_device(a) for a in (depth, rgb, class_ids_image, object_ids_image)):
            # device-resident keyframe (torch CUDA tensors: depth f32, rgb u8, labels i32): nothing is staged
            import torch

            (depth, rgb, cls, obj), loc = _device_images(
                depth if _is_device(depth) else np.ascontiguousarray(depth, dtype=np.float32),
                rgb if _is_device(rgb) else np.ascontiguousarray(rgb, dtype=np.uint8),
                class_ids_image if class_ids_image is None or _is_device(class_ids_image) else np.ascontiguousarray(class_ids_image, dtype=np.int32),
                object_ids_image if object_ids_image is None or _is_device(object_ids_image) else np.ascontiguousarray(object_ids_image, dtype=np.int32),
                device=self._cfg.device)
            if depth.dtype != torch.float32 or rgb.dtype != torch.uint8 or any(a is not None and a.dtype != torch.int32 for a in (cls, obj)):
                raise RuntimeError("device images must be float32 depth, uint8 colour, int32 labels")
            H, W = int(depth.shape[0]), int(depth.shape[1])
            ts = self._torch_in(depth, rgb, cls, obj)
        else:
            loc = L.HV_HOST
            depth = np.ascontiguousarray(depth, dtype=np.float32)
            rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
            H, W = depth.shape
            cls = None if class_ids_image is None else np.ascontiguousarray(class_ids_image, dtype=np.int32)
            obj = None if object_ids_image is None else np.ascontiguousarray(object_ids_image, dtype=np.int32)
        if tuple(rgb.shape[:2]) != (H, W):
            raise RuntimeError("depth and colour image sizes differ")
        for a, name in ((cls, "class_ids"), (obj, "object_ids")):
            if a is not None and tuple(a.shape) != (H, W):
                raise RuntimeError(f"depth and {name} image sizes differ")
        intr = np.array([fx, fy, cx, cy], np.float64)
        T = np.ascontiguousarray(T_cw, dtype=np.float64)
        big = float(np.finfo(np.float32).max)
        L.check(self._lib.hv_integrate_rgbd_semantic(self._h, L.ptr(depth), L.ptr(rgb), L.ptr(cls), L.ptr(obj), H, W, L.ptr(intr), L.ptr(T),
                                                     float(min_depth), float(min(max_depth, big)), int(bool(use_depths)), loc))
        if loc == L.HV_DEVICE:
            self._torch_out(ts, depth.device)

    def fuse_keyframe(self, camera_frustrum, depth, rgb, class_ids_image, instance_ids_image, fx, fy, cx, cy, T_cw, filter_shadow_points=True,
                      use_instance_ids=True, depth_threshold=0.1, do_carving=False, min_vote_ratio=0.5, min_votes=3, max_depth=np.inf,
                      min_depth=0.0, use_depths=True, depth_is_filtered=False):
        """One keyframe of the integrator's flow - filter_shadow_points -> assign_object_ids_to_instance_ids -> remap_instance_ids (or
        carve) -> integrate_rgbd - in ONE call into the library (hv_semantic_fuse_keyframe) on device-resident images (torch CUDA
        tensors: depth f32, rgb u8, labels i32 or None); queued on the volume's stream, nothing waits.  The same kernels in the same
        order as the separate calls; what goes away is the host time of five calls.  Multi-GPU grids (a pair exchange between vote and
        decide) keep the staged calls."""
        import torch

        if self._pair_exchange is not None or self._pair_exchange_device is not None:
            raise RuntimeError("fuse_keyframe: a sharded grid exchanges its pair lists between vote and decide - use the staged calls")
        (depth, rgb, cls, inst), _ = _device_images(depth, rgb, class_ids_image, instance_ids_image, device=self._cfg.device)
        if depth.dtype != torch.float32 or rgb.dtype != torch.uint8 or any(a is not None and a.dtype != torch.int32 for a in (cls, inst)):
            raise RuntimeError("device images must be float32 depth, uint8 colour, int32 labels")
        H, W = int(depth.shape[0]), int(depth.shape[1])
        f = camera_frustrum
        if tuple(rgb.shape[:2]) != (H, W) or any(a is not None and tuple(a.shape) != (H, W) for a in (cls, inst)) or (f.height, f.width) != (H, W):
            raise RuntimeError("fuse_keyframe: image sizes differ")
        use_inst = bool(use_instance_ids) and inst is not None
        if use_inst and cls is not None:
            prev = getattr(self, "_last_map_ref", None)
            prev = prev() if prev is not None else None
            if prev is not None and prev._d is None:
                prev._get()  # somebody still holds the previous association's map and has not read it: fetch it before it is replaced
        ts = self._torch_in(depth, rgb, cls, inst)
        intr = np.array([fx, fy, cx, cy], np.float64)
        T = np.ascontiguousarray(T_cw, dtype=np.float64)
        f.set_T_cw(T)
        big = float(np.finfo(np.float32).max)
        L.check(self._lib.hv_semantic_fuse_keyframe(
            self._h, L.ptr(depth), L.ptr(rgb), L.ptr(cls), L.ptr(inst), H, W, L.ptr(f.intr), f.depth_max, f.depth_min, L.ptr(intr), L.ptr(T),
            int(bool(filter_shadow_points) and not depth_is_filtered), int(use_inst), float(depth_threshold), int(bool(do_carving)),
            float(min_vote_ratio), int(min_votes), float(min_depth), float(min(max_depth, big)), int(bool(use_depths))))
        if use_inst and cls is not None:
            self._assoc_serial = getattr(self, "_assoc_serial", 0) + 1  # (a LazyIdMap of an earlier association is stale now)
        self._torch_out(ts, depth.device)

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

    def _query(self, call):
        n = ctypes.c_int64()
        L.check(call(None, None, None, None, None, 0, ctypes.byref(n)))
        m = n.value
        out = VoxelGridData(np.zeros((m, 3), np.float64), np.zeros((m, 3), np.float32))
        out.class_ids, out.object_ids = np.zeros(m, np.int32), np.zeros(m, np.int32)
        out.confidences = np.zeros(m, np.float32)
        if m:
            L.check(call(L.ptr(out.points), L.ptr(out.colors), L.ptr(out.class_ids), L.ptr(out.object_ids), L.ptr(out.confidences), m,
                         ctypes.byref(n)))
        return out

    @staticmethod
    def _strip_semantics(out, include_semantics):
        if not include_semantics:  # IncludeSemantics=false: only points / colours are filled (voxel_block_grid.hpp:1004-1010)
            out.class_ids, out.object_ids = np.zeros(0, np.int32), np.zeros(0, np.int32)
            out.confidences = np.zeros(0, np.float32)
        return out

    def get_voxels_in_bb(self, bbox, min_count=1, min_confidence=0.0, include_semantics=False):
        bb = bbox.as_array() if hasattr(bbox, "as_array") else np.ascontiguousarray(bbox, dtype=np.float64)
        out = self._query(lambda *a: self._lib.hv_get_voxels_semantic_in_bb(self._h, L.ptr(bb), int(min_count), float(min_confidence), *a))
        return self._strip_semantics(out, include_semantics)

    def get_voxels_in_camera_frustrum(self, camera_frustrum, min_count=1, min_confidence=0.0, include_semantics=False):
        f = camera_frustrum
        out = self._query(lambda *a: self._lib.hv_get_voxels_semantic_in_frustum(
            self._h, L.ptr(f.intr), f.width, f.height, L.ptr(f.T_cw), f.depth_max, f.depth_min, int(min_count), float(min_confidence), *a))
        return self._strip_semantics(out, include_semantics)

    def integrate_segment(self, points, colors, object_id, class_id):
        """integrate_segment(points, colors, object_id, class_id) (voxel_block_semantic_grid.hpp:39-96): every point
        carries the same ids; nothing happens for negative ids."""
        if object_id < 0 or class_id < 0:
            return
        n = np.asarray(points).shape[0]
        self.integrate(points, colors, np.full(n, class_id, np.int32), np.full(n, object_id, np.int32))

    def get_class_segments(self, min_count=1, min_confidence=0.0):
        """-> ClassDataGroup (.class_vector of ClassData, .class_ids; voxel_block_semantic_grid.hpp:269-313): voxels with count > min_count,
        confidence >= min_confidence and class id >= 0 grouped by class (ascending)."""
        v = self.get_voxels(int(min_count) + 1, min_confidence)  # strict '>' on the count, like get_object_segments
        keep = v.class_ids >= 0
        out = []
        for c in np.unique(v.class_ids[keep]):
            m = keep & (v.class_ids == c)
            out.append(ClassData(v.points[m], v.colors[m], int(c), float(v.confidences[m].min()), float(v.confidences[m].max())))
        return ClassDataGroup(out)

    def get_points(self):
        return self.get_voxels(1, -1.0).points

    def get_colors(self):
        return self.get_voxels(1, -1.0).colors

    def get_ids(self):
        """-> (class_ids, object_ids) of every voxel with count > 0 (voxel_block_semantic_grid.hpp:198-213)."""
        vg = self.get_voxels(1, -1.0)
        return vg.class_ids, vg.object_ids

    def carve(self, camera_frustrum, depth_image, depth_threshold=1e-2):
        self._carve(camera_frustrum, depth_image, depth_threshold)

    def assign_object_ids_to_instance_ids(self, camera_frustrum, class_ids_image, semantic_instances_image, depth_image=None,
                                          depth_threshold=0.1, do_carving=False, min_vote_ratio=0.5, min_votes=3):
        """-> dict instance_id -> object_id (voxel_semantic_data_association.h:70-373), as a LazyIdMap: vote -> (multi-GPU: exchange
        of the pair lists) -> decide are queued on the volume's stream and nothing waits; the map stays in device memory until
        somebody reads it (remap_instance_ids on this volume uses it there)."""
        if not self.assoc_vote(camera_frustrum, class_ids_image, semantic_instances_image, depth_image, depth_threshold, do_carving):
            return {}
        if self._pair_exchange_device is not None:
            self._pair_exchange_device(self)  # RCCL: the lists never leave the device (hv_assoc_pairs_export / _import)
        elif self._pair_exchange is not None:
            self.assoc_set_pairs(*self._pair_exchange(*self.assoc_pairs()))
        return self.assoc_decide(min_vote_ratio, min_votes)

    _pair_exchange = None  # multi-GPU: callable(keys u64[n], counts i32[n]) -> (keys, counts) of all ranks, concatenated (ShardedSemanticGrid)
    _pair_exchange_device = None  # multi-GPU over RCCL: callable(grid) that runs assoc_pairs_export -> all-gather -> assoc_pairs_import
    ASSOC_PAIRS_CAP = 4096  # pairs in one GPU's exchange message (HV_RULES_MAX)

    def assoc_pairs_export(self, msg):
        """This GPU's pair list of stage 1 into `msg`, a torch CUDA int64 tensor of 1 + 2 * ASSOC_PAIRS_CAP words ([n, keys, votes]);
        queued on the volume's stream, ordered against torch's current stream - no host synchronisation."""
        cap = (msg.numel() - 1) // 2
        ts = self._torch_in(msg)
        L.check(self._lib.hv_assoc_pairs_export(self._h, L.ptr(msg), cap))
        self._torch_out(ts, msg.device)

    def assoc_pairs_import(self, msgs, world):
        """The ranks' messages back to back (the all-gather's output) become the pair list stage 2 decides on; equal pairs add up."""
        cap = (msgs.numel() // int(world) - 1) // 2
        ts = self._torch_in(msgs)
        L.check(self._lib.hv_assoc_pairs_import(self._h, L.ptr(msgs), int(world), cap))
        self._torch_out(ts, msgs.device)

    # -- the association in stages (hv_assoc_*): what a multi-GPU driver interleaves with its exchange ----------------------------
    def assoc_vote(self, camera_frustrum, class_ids_image, semantic_instances_image, depth_image=None, depth_threshold=0.1,
                   do_carving=False):
        """Stage 1: this grid's voxels vote (instance, object) pairs.  -> False where the reference returns an empty map without
        looking at the grid (missing / mis-sized label images)."""
        f = camera_frustrum
        if class_ids_image is None or semantic_instances_image is None:
            return False
        loc = L.HV_HOST
        ts = None
        if any(_is_device(a) for a in (class_ids_image, semantic_instances_image, depth_image)):
            # device-resident label / depth images (torch CUDA: int32, int32, float32): used in place
            import torch

            (cls, inst, depth), loc = _device_images(
                class_ids_image if _is_device(class_ids_image) else _i32_image(np.asarray(class_ids_image), "Class ids"),
                semantic_instances_image if _is_device(semantic_instances_image) else _i32_image(np.asarray(semantic_instances_image), "Instance ids"),
                depth_image if depth_image is None or _is_device(depth_image) else np.ascontiguousarray(depth_image, dtype=np.float32),
                device=self._cfg.device)
            if cls.dim() != 2 or inst.dim() != 2 or cls.dtype != torch.int32 or inst.dtype != torch.int32:
                raise RuntimeError("Class ids / Instance ids must be single-channel int32")
            if tuple(inst.shape) != (f.height, f.width) or tuple(cls.shape) != (f.height, f.width):
                return False
            if depth is not None and (depth.dtype != torch.float32 or tuple(depth.shape) != (f.height, f.width)):
                depth = None
            ts = self._torch_in(cls, inst, depth)
        else:
            cls, inst = np.asarray(class_ids_image), np.asarray(semantic_instances_image)
            if cls.size == 0 or inst.size == 0:
                return False
            cls, inst = _i32_image(cls, "Class ids"), _i32_image(inst, "Instance ids")
            if inst.shape != (f.height, f.width) or cls.shape != (f.height, f.width):
                return False  # check_image_size(): message + empty map
            depth = None
            if depth_image is not None and np.asarray(depth_image).size > 0:
                depth = np.ascontiguousarray(depth_image, dtype=np.float32)
                if depth.shape != (f.height, f.width):
                    depth = None  # use_depth_filter = false
        prev = getattr(self, "_last_map_ref", None)
        prev = prev() if prev is not None else None
        if prev is not None and prev._d is None:
            prev._get()  # somebody still holds the previous association's map and has not read it: fetch it before it is replaced
        L.check(self._lib.hv_assoc_vote(
            self._h, L.ptr(f.intr), f.width, f.height, L.ptr(f.T_cw), f.depth_max, f.depth_min, L.ptr(cls), L.ptr(inst), L.ptr(depth),
            float(depth_threshold), int(bool(do_carving)), loc))
        if loc == L.HV_DEVICE:
            self._torch_out(ts, cls.device)
        return True

    def assoc_pairs(self):
        """-> (keys u64 [n] = instance << 32 | object, votes i32 [n]) of stage 1 (synchronises)."""
        n = ctypes.c_int64()
        L.check(self._lib.hv_assoc_pairs_fetch(self._h, None, None, 0, ctypes.byref(n)))
        keys, counts = np.zeros(n.value, np.uint64), np.zeros(n.value, np.int32)
        if n.value:
            L.check(self._lib.hv_assoc_pairs_fetch(self._h, L.ptr(keys), L.ptr(counts), n.value, ctypes.byref(n)))
        return keys, counts

    def assoc_set_pairs(self, keys, counts):
        """Replace the pairs stage 2 decides on (multi-GPU: the concatenation of every rank's list; equal pairs are added up)."""
        keys, counts = np.ascontiguousarray(keys, np.uint64), np.ascontiguousarray(counts, np.int32)
        L.check(self._lib.hv_assoc_pairs_set(self._h, L.ptr(keys), L.ptr(counts), len(keys)))

    def assoc_decide(self, min_vote_ratio=0.5, min_votes=3):
        """Stage 2: the reference's rules, new object ids, deferred assignments - on the device.  -> LazyIdMap."""
        L.check(self._lib.hv_assoc_decide(self._h, float(min_vote_ratio), int(min_votes)))
        self._assoc_serial = getattr(self, "_assoc_serial", 0) + 1
        m = LazyIdMap(self, self._assoc_serial)
        self._last_map_ref = weakref.ref(m)
        return m

    def remap_instance_ids(self, instance_ids, instance_id_to_object_id):
        return remap_instance_ids(instance_ids, instance_id_to_object_id, volume=self)

    def get_object_segments(self, min_count=1, min_confidence=0.0):
        """-> ObjectDataGroup (voxel_block_semantic_grid.hpp:217-267); objects in ascending object-id order."""
        nr, no = ctypes.c_int64(), ctypes.c_int64()
        L.check(self._lib.hv_object_segments_compute(self._h, int(min_count), float(min_confidence), ctypes.byref(nr), ctypes.byref(no)))
        R, O = nr.value, no.value
        pts, cols = np.zeros((R, 3), np.float64), np.zeros((R, 3), np.float32)
        ids, conf, obb = np.zeros((O, 3), np.int32), np.zeros((O, 2), np.float32), np.zeros((O, 10), np.float64)
        if R:
            L.check(self._lib.hv_object_segments_fetch(self._h, L.ptr(pts), L.ptr(cols), None, L.ptr(ids), L.ptr(conf), L.ptr(obb)))
        objs, at = [], 0
        for o in range(O):
            k = int(ids[o, 2])
            objs.append(ObjectData(pts[at:at + k], cols[at:at + k], ids[o, 0], ids[o, 1], conf[o, 0], conf[o, 1],
                                   OrientedBoundingBox3D(obb[o, 0:3], obb[o, 3:7], obb[o, 7:10])))
            at += k
        return ObjectDataGroup(objs)

    def merge_segments(self, instance_id1, instance_id2):
        L.check(self._lib.hv_merge_segments(self._h, int(instance_id1), int(instance_id2)))

    def remove_segment(self, object_id):
        L.check(self._lib.hv_remove_segment(self._h, int(object_id)))

    def remove_low_confidence_segments(self, min_confidence):
        L.check(self._lib.hv_remove_low_confidence_segments(self._h, int(min_confidence)))

    def remove_low_count_voxels(self, min_count):
        L.check(self._lib.hv_remove_low_count_voxels(self._h, int(min_count)))

    def remove_low_confidence_voxels(self, min_confidence):
        L.check(self._lib.hv_remove_low_confidence_voxels(self._h, float(min_confidence)))

    def label_overflows(self):
        """Label observations the probabilistic payload dropped (a map past 254 pairs / the overflow-node pool exhausted); 0 otherwise."""
        n = ctypes.c_int64()
        L.check(self._lib.hv_label_overflows(self._h, ctypes.byref(n)))
        return n.value

    def prob_nodes_used(self):
        """Overflow nodes of the probabilistic label maps handed out so far (hv_prob_nodes_used)."""
        n = ctypes.c_int64()
        L.check(self._lib.hv_prob_nodes_used(self._h, ctypes.byref(n)))
        return n.value

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
        """-> keys [B,3], ints [B,bs^3,4] {count, object_id, class_id, confidence_counter}, pos_sums f64, col_sums f32."""
        return self.dump2()[:4]

    def dump2(self, max_labels=7):
        """dump() + conf [B,bs^3] f32, label_counts [B,bs^3] (each voxel's map size), labels [B,bs^3,max_labels,2] and
        log_probs [B,bs^3,max_labels] (each map's first max_labels pairs, insertion order)."""
        nb, nv = self.num_blocks(), self.block_size ** 3
        keys = np.zeros((nb, 3), np.int32)
        ints = np.zeros((nb, nv, 4), np.int32)
        conf = np.zeros((nb, nv), np.float32)
        pos = np.zeros((nb, nv, 3), np.float64)
        col = np.zeros((nb, nv, 3), np.float32)
        nlab = np.zeros((nb, nv), np.int32)
        labels = np.zeros((nb, nv, max_labels, 2), np.int32)
        logp = np.zeros((nb, nv, max_labels), np.float32)
        n = ctypes.c_int64()
        L.check(self._lib.hv_dump_blocks_semantic2(self._h, L.ptr(keys), L.ptr(ints), L.ptr(conf), L.ptr(pos), L.ptr(col), L.ptr(nlab),
                                                   L.ptr(labels), L.ptr(logp), int(max_labels), ctypes.byref(n)))
        return keys, ints, pos, col, conf, nlab, labels, logp

    def dump_marginals(self):
        """-> (object confidence, class confidence) [B,bs^3] f32 in dump()'s order: ``get_object_confidence()`` /
        ``get_class_confidence()`` of the two ``*2`` payloads (voxel_data_semantic2.h:60-76, 528-560); -1 for the other two."""
        nb, nv = self.num_blocks(), self.block_size ** 3
        oc, cc = np.zeros((nb, nv), np.float32), np.zeros((nb, nv), np.float32)
        n = ctypes.c_int64()
        L.check(self._lib.hv_dump_marginals_semantic(self._h, L.ptr(oc), L.ptr(cc), ctypes.byref(n)))
        return oc, cc


class VoxelBlockSemanticGrid(_SemanticGridBase):
    _MODE = L.HV_MODE_VOXEL_SEMANTIC_GRID


class VoxelBlockSemanticProbabilisticGrid(_SemanticGridBase):
    _MODE = L.HV_MODE_VOXEL_SEMANTIC_PROBABILISTIC_GRID


class VoxelBlockSemanticGrid2(_SemanticGridBase):
    """``volumetric.VoxelBlockSemanticGrid2`` (volumetric_grid_module.h:1014-1018): the voting payload with one confidence counter
    for the object id and one for the class id (VoxelSemanticData2, voxel_data_semantic2.h:46-196)."""
    _MODE = L.HV_MODE_VOXEL_SEMANTIC_GRID2


class VoxelBlockSemanticProbabilisticGrid2(_SemanticGridBase):
    """``volumetric.VoxelBlockSemanticProbabilisticGrid2`` (volumetric_grid_module.h:1028-1032): one log-probability map per object
    id and one per class id (VoxelSemanticDataProbabilistic2, voxel_data_semantic2.h:256-787)."""
    _MODE = L.HV_MODE_VOXEL_SEMANTIC_PROBABILISTIC_GRID2


class VoxelSemanticGrid(VoxelBlockSemanticGrid):
    """``volumetric.VoxelSemanticGrid(voxel_size)``: the direct-hash variant of the voting grid
    (cpp/volumetric/voxel_semantic_grid.h); same payload and observable results as the block grid."""

    def __init__(self, voxel_size=0.05, device=0, max_blocks=None, max_points=None):
        super().__init__(voxel_size, 8, device=device, max_blocks=max_blocks, max_points=max_points)


class VoxelSemanticGridProbabilistic(VoxelBlockSemanticProbabilisticGrid):
    """``volumetric.VoxelSemanticGridProbabilistic(voxel_size)`` (direct-hash variant, same payload)."""

    def __init__(self, voxel_size=0.05, device=0, max_blocks=None, max_points=None):
        super().__init__(voxel_size, 8, device=device, max_blocks=max_blocks, max_points=max_points)


class VoxelSemanticGrid2(VoxelBlockSemanticGrid2):
    """``volumetric.VoxelSemanticGrid2(voxel_size)`` (direct-hash variant, same payload; volumetric_grid_module.h:987-990)."""

    def __init__(self, voxel_size=0.05, device=0, max_blocks=None, max_points=None):
        super().__init__(voxel_size, 8, device=device, max_blocks=max_blocks, max_points=max_points)


class VoxelSemanticGridProbabilistic2(VoxelBlockSemanticProbabilisticGrid2):
    """``volumetric.VoxelSemanticGridProbabilistic2(voxel_size)`` (direct-hash variant, same payload; :1000-1004)."""

    def __init__(self, voxel_size=0.05, device=0, max_blocks=None, max_points=None):
        super().__init__(voxel_size, 8, device=device, max_blocks=max_blocks, max_points=max_points)


def get_next_object_id_peek():
    return L.load().hv_peek_next_object_id()


def set_next_object_id(value):
    L.load().hv_set_next_object_id(int(value))
