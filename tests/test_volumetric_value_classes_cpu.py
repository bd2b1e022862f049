"""CPU: the value classes the reference's module binds beside the grids - VoxelData, VoxelSemanticData, VoxelGridData - through the
calls of the reference's own API-surface script (cpp/test_volumetric.py:613-699: default construction, getters, the read / write
`count`, the attributes of VoxelGridData), plus the values that follow from voxel_data.h / voxel_data_semantic.h for a default object."""
import math

import numpy as np


def test_voxel_data_classes_like_the_reference_script():
    from pyslam_amd.volumetric import VoxelData
    from pyslam_amd.volumetric_semantic import VoxelSemanticData

    v = VoxelData()
    assert all(math.isnan(x) for x in v.get_position()) and all(math.isnan(x) for x in v.get_color())  # 0 / 0: release build, no zero-count check
    v.count = 5
    assert v.count == 5 and v.get_position() == [0.0, 0.0, 0.0] and v.get_color() == [0.0, 0.0, 0.0]
    s = VoxelSemanticData()
    assert (s.get_object_id(), s.get_class_id(), s.get_confidence(), s.get_confidence_counter()) == (-1, -1, 0.0, 0)
    s.count = 10
    assert s.count == 10 and s.get_position() == [0.0, 0.0, 0.0] and s.get_confidence() == 0.0
    s.confidence_counter = 4
    assert s.get_confidence() == float(np.float32(4) / np.float32(10))
    s.confidence_counter = 40
    assert s.get_confidence() == 1.0  # std::min(1.0f, counter / count)


def test_voxel_grid_data_like_the_reference_script():
    from pyslam_amd.volumetric import VoxelGridData

    d = VoxelGridData()
    for name in ("points", "colors", "object_ids", "class_ids", "confidences"):
        assert hasattr(d, name) and len(getattr(d, name)) == 0


def test_tbb_utils_like_the_reference_script():
    """cpp/test_volumetric.py:82-112: get, set 4, get, restore."""
    from pyslam_amd.volumetric import TBBUtils

    before = TBBUtils.get_max_threads()
    assert before >= 1
    assert TBBUtils.set_max_threads(4) == 4 and TBBUtils.get_max_threads() == 4
    assert TBBUtils.set_max_threads(before) == before and TBBUtils.get_max_threads() == before
    assert TBBUtils.set_max_threads(0) >= 1  # <= 0: the default


def test_extraction_dtype_argument_is_float64_or_float32():
    """extract_triangle_mesh / extract_point_cloud(dtype=...): Open3D's float64 (default) or float32 (hv_tsdf_extract_mesh_f32 /
    _points_f32); anything else is refused before the library is called."""
    import pytest

    from pyslam_amd.volumetric import ScalableTSDFVolume

    assert ScalableTSDFVolume._out_dtype(None) == np.float64 and ScalableTSDFVolume._out_dtype("float32") == np.float32
    assert ScalableTSDFVolume._out_dtype(np.float64) == np.float64 and ScalableTSDFVolume._out_dtype(np.dtype("f4")) == np.float32
    for bad in (np.float16, np.int32, "uint8"):
        with pytest.raises(TypeError):
            ScalableTSDFVolume._out_dtype(bad)
