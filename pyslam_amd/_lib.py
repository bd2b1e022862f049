"""ctypes binding of libpyslam_hipvol.so (include/hipvol.h).

There is no CPU fallback: if the shared object is missing it is built with hipcc; if that fails, or
no gfx950 device is usable when a volume is created, the call raises.
"""
import ctypes
import os

import numpy as np

from . import build as _build

_c = ctypes
_vp, _i64, _i32, _f32, _f64 = _c.c_void_p, _c.c_int64, _c.c_int32, _c.c_float, _c.c_double
_pi64 = _c.POINTER(_c.c_int64)
_pi32 = _c.POINTER(_c.c_int32)

HV_OK = 0
HV_MODE_VOXEL_GRID = 0
HV_MODE_VOXEL_SEMANTIC_GRID = 1
HV_MODE_VOXEL_SEMANTIC_PROBABILISTIC_GRID = 2
HV_MODE_TSDF = 3
HV_MODE_VOXEL_SEMANTIC_GRID2 = 11
HV_MODE_VOXEL_SEMANTIC_PROBABILISTIC_GRID2 = 12
HV_HOST, HV_DEVICE = 0, 1
HV_COLOR_NONE, HV_COLOR_U8, HV_COLOR_F32 = 0, 1, 2
HV_DEPTH_F32, HV_DEPTH_U16 = 0, 1


class HvConfig(_c.Structure):
    _fields_ = [
        ("mode", _i32),
        ("device", _i32),
        ("voxel_size", _f64),
        ("sdf_trunc", _f64),
        ("block_size", _i32),
        ("depth_sampling_stride", _i32),
        ("max_blocks", _i64),
        ("max_points", _i64),
    ]


# name -> (restype, argtypes); mirrors include/hipvol.h one to one
SIGNATURES = {
    "hv_last_error": (_c.c_char_p, []),
    "hv_device_count": (_i32, []),
    "hv_default_config": (None, [_i32, _c.POINTER(HvConfig)]),
    "hv_create": (_i32, [_c.POINTER(HvConfig), _c.POINTER(_vp)]),
    "hv_destroy": (None, [_vp]),
    "hv_reset": (_i32, [_vp]),
    "hv_synchronize": (_i32, [_vp]),
    "hv_set_stream": (_i32, [_vp, _vp]),
    "hv_get_stream": (_vp, [_vp]),
    "hv_num_blocks": (_i32, [_vp, _pi64]),
    "hv_block_size": (_i32, [_vp, _pi32]),
    "hv_reserve_blocks": (_i32, [_vp, _i64]),
    "hv_max_blocks": (_i32, [_vp, _pi64]),
    "hv_bytes_per_block": (_i32, [_vp, _pi64]),
    "hv_dropped_points": (_i32, [_vp, _pi64]),
    "hv_integrate_points": (_i32, [_vp, _vp, _i64, _vp, _i32, _i32]),
    "hv_integrate_points_f64": (_i32, [_vp, _vp, _i64, _vp, _i32, _i32]),
    "hv_integrate_rgbd_points": (_i32, [_vp, _vp, _i32, _f64, _vp, _i32, _i32, _vp, _vp, _f64, _f64, _i32]),
    "hv_integrate_rgbd_points_batch": (_i32, [_vp, _vp, _i32, _f64, _vp, _i32, _i32, _i32, _vp, _vp, _f64, _f64, _i32]),
    "hv_remap": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32]),
    "hv_filter_shadow_points": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _i32]),
    "hv_filter_shadow_points_on_stream": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp]),
    "hv_get_voxels": (_i32, [_vp, _i32, _f32, _vp, _vp, _i64, _pi64, _i32]),
    "hv_get_voxels_in_bb": (_i32, [_vp, _vp, _i32, _f32, _vp, _vp, _i64, _pi64, _i32]),
    "hv_get_voxels_in_frustum": (_i32, [_vp, _vp, _i32, _i32, _vp, _f32, _f32, _i32, _f32, _vp, _vp, _i64, _pi64, _i32]),
    "hv_carve": (_i32, [_vp, _vp, _i32, _i32, _vp, _f32, _f32, _vp, _f32, _i32]),
    "hv_remove_low_count_voxels": (_i32, [_vp, _i32]),
    "hv_size": (_i32, [_vp, _pi64]),
    "hv_dump_blocks": (_i32, [_vp, _vp, _vp, _vp, _vp, _pi64]),
    "hv_keys_from_points": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "hv_integrate_points_semantic": (_i32, [_vp, _vp, _i32, _i64, _vp, _i32, _vp, _vp, _vp, _i32]),
    "hv_get_voxels_semantic": (_i32, [_vp, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _i64, _pi64]),
    "hv_get_voxels_semantic_in_bb": (_i32, [_vp, _vp, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _i64, _pi64]),
    "hv_get_voxels_semantic_in_frustum": (_i32, [_vp, _vp, _i32, _i32, _vp, _f32, _f32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _i64, _pi64]),
    "hv_set_depth_threshold": (_i32, [_vp, _f32]),
    "hv_dump_blocks_semantic": (_i32, [_vp, _vp, _vp, _vp, _vp, _pi64]),
    "hv_dump_blocks_semantic2": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _pi64]),
    "hv_dump_marginals_semantic": (_i32, [_vp, _vp, _vp, _pi64]),
    "hv_set_depth_decay_rate": (_i32, [_vp, _f32]),
    "hv_integrate_rgbd_semantic": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _f64, _f64, _i32, _i32]),
    "hv_label_overflows": (_i32, [_vp, _pi64]),
    "hv_prob_nodes_used": (_i32, [_vp, _pi64]),
    "hv_assign_object_ids_to_instance_ids": (_i32, [_vp, _vp, _i32, _i32, _vp, _f32, _f32, _vp, _vp, _vp, _f32, _i32, _f32, _i32,
                                                    _vp, _vp, _i64, _pi64, _i32]),
    "hv_assoc_vote": (_i32, [_vp, _vp, _i32, _i32, _vp, _f32, _f32, _vp, _vp, _vp, _f32, _i32, _i32]),
    "hv_assoc_pairs_fetch": (_i32, [_vp, _vp, _vp, _i64, _pi64]),
    "hv_assoc_pairs_set": (_i32, [_vp, _vp, _vp, _i64]),
    "hv_assoc_pairs_export": (_i32, [_vp, _vp, _i64]),
    "hv_assoc_pairs_import": (_i32, [_vp, _vp, _i32, _i64]),
    "hv_assoc_decide": (_i32, [_vp, _f32, _i32]),
    "hv_assoc_map_fetch": (_i32, [_vp, _vp, _vp, _i64, _pi64]),
    "hv_remap_instance_ids_last": (_i32, [_vp, _vp, _i32, _i32, _vp, _i32]),
    "hv_semantic_fuse_keyframe": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _f32, _f32, _vp, _vp, _i32, _i32, _f32, _i32, _f32, _i32,
                                         _f64, _f64, _i32]),
    "hv_peek_next_object_id": (_i32, []),
    "hv_set_next_object_id": (None, [_i32]),
    "hv_remap_instance_ids": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _i64, _vp, _i32]),
    "hv_object_segments_compute": (_i32, [_vp, _i32, _f32, _pi64, _pi64]),
    "hv_object_segments_fetch": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hv_compute_obb_pca": (_i32, [_vp, _i64, _vp]),
    "hv_merge_segments": (_i32, [_vp, _i32, _i32]),
    "hv_remove_segment": (_i32, [_vp, _i32]),
    "hv_remove_low_confidence_segments": (_i32, [_vp, _i32]),
    "hv_remove_low_confidence_voxels": (_i32, [_vp, _f32]),
    "hv_tsdf_integrate": (_i32, [_vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _f64, _f64, _i32]),
    "hv_tsdf_integrate_batch": (_i32, [_vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _f64, _f64, _i32]),
    "hv_tsdf_integrate_frames": (_i32, [_vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _f64, _f64]),
    "hv_tsdf_set_color_order": (_i32, [_vp, _i32]),
    "hv_host_register": (_i32, [_vp, _i64]),
    "hv_host_unregister": (_i32, [_vp]),
    "hv_tsdf_set_tile": (_i32, [_vp, _i32, _i32, _i32, _i32]),
    "hv_tsdf_set_rectify_maps": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32]),
    "hv_tsdf_set_owner": (_i32, [_vp, _i32, _i32]),
    "hv_tsdf_extract_mesh": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _pi64, _pi64]),
    "hv_tsdf_extract_mesh_f32": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _pi64, _pi64]),
    "hv_tsdf_extract_points": (_i32, [_vp, _vp, _vp, _i64, _pi64]),
    "hv_tsdf_extract_points_f32": (_i32, [_vp, _vp, _vp, _i64, _pi64]),
    "hv_tsdf_extract_point_normals": (_i32, [_vp, _vp, _i64, _pi64]),
    "hv_tsdf_dump": (_i32, [_vp, _vp, _vp, _vp, _vp, _pi64]),
    "hv_tsdf_touched": (_i32, [_vp, _vp, _i64, _pi64]),
    "hv_tsdf_export_numerators": (_i32, [_vp, _vp, _i64, _vp, _i32]),
    "hv_tsdf_import_numerators": (_i32, [_vp, _vp, _i64, _vp, _i32]),
    "hv_tsdf_unit_keys": (_i32, [_vp, _vp, _i64, _pi64]),
    "hv_tsdf_dirty_keys": (_i32, [_vp, _vp, _i64, _pi64]),
    "hv_tsdf_mark_merged": (_i32, [_vp]),
    "hv_set_owner": (_i32, [_vp, _i32, _i32]),
    "hv_block_owner": (_i32, [_vp, _i64, _i32, _vp]),
    "hv_merge_halo_plan": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _i64, _pi64]),
    "hv_merge_halo_plan_held": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _i64, _pi64]),
    "hv_merge_halo_pack": (_i32, [_vp, _vp, _i64, _vp, _i32]),
    "hv_merge_halo_unpack": (_i32, [_vp, _vp, _i64, _vp, _vp, _i32]),
    "hv_merge_halo_lists_device": (_i32, [_vp, _vp, _i64, _vp, _i64, _pi64, _pi64]),
    "hv_merge_halo_plan_device": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _pi64]),
    "hv_merge_halo_plan_fetch": (_i32, [_vp, _vp, _vp, _i64, _pi64]),
    "hv_merge_halo_pack_planned": (_i32, [_vp, _i64, _i64, _vp]),
    "hv_merge_halo_unpack_planned": (_i32, [_vp, _i64, _i64, _vp]),
    "hv_profile_enable": (_i32, [_vp, _i32]),
    "hv_profile_read": (_i32, [_vp, _c.POINTER(_f64), _pi64, _pi64]),
    "hv_profile_read_launches": (_i32, [_vp, _vp, _i64, _pi64]),
}

_lib = None


class HipVolError(RuntimeError):
    """Raised for every non-zero hv_status; message = hv_last_error()."""


def library_path():
    return _build.LIB_PATH


def load():
    """Load (building if needed) libpyslam_hipvol.so and bind every symbol of include/hipvol.h."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    if not os.path.exists(path):
        _build.build(verbose=False)
    # One HIP runtime per process: PyTorch-ROCm dlopens its bundled libamdhip64/libhsa-runtime64 by
    # absolute path, so if this library pulled in /opt/rocm's copies first the process would hold
    # two HSA runtimes and whichever initialises second sees no devices.  Importing torch first
    # makes our DT_NEEDED libamdhip64.so.7 resolve to the already-loaded runtime, which is also
    # what lets torch CUDA tensors and torch streams be passed straight into the C ABI.
    import torch  # noqa: F401

    lib = _c.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc):
    if rc != HV_OK:
        msg = load().hv_last_error()
        raise HipVolError(msg.decode() if msg else f"hipvol error {rc}")


def ptr(a):
    """Device or host address of a numpy array / torch tensor / None."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(_vp)
    if hasattr(a, "data_ptr"):
        return _vp(a.data_ptr())
    raise TypeError(f"unsupported buffer type {type(a)}")


def location(a):
    """HV_DEVICE for torch tensors living on a GPU, HV_HOST otherwise."""
    if a is not None and hasattr(a, "is_cuda") and a.is_cuda:
        return HV_DEVICE
    return HV_HOST
