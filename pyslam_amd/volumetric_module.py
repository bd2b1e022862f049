"""One namespace with the names of the reference's ``volumetric`` extension module (cpp/volumetric/volumetric_module.cpp:33-62 and the
bind_* functions it calls): code written against ``import volumetric`` runs against ``import pyslam_amd.volumetric_module as volumetric``.

Every bound name is here (tests/test_volumetric_module_names_cpu.py).  The ``F`` twins of the result classes (float32 positions, voxel_grid_data_module.h:179-188) are the same
Python classes: the arrays carry their dtype."""
from .bounding_boxes_2d import BoundingBox2D, OrientedBoundingBox2D  # noqa: F401
from .volumetric import (BoundingBox3D, CameraFrustrum, ImagePoint, Quaterniond, TBBUtils, VoxelBlockGrid, VoxelData, VoxelGrid,  # noqa: F401
                         VoxelGridData)
from .volumetric_semantic import (ClassData, ClassDataGroup, OBBComputationMethod, ObjectData, ObjectDataGroup, OrientedBoundingBox3D,  # noqa: F401
                                  VoxelBlockSemanticGrid, VoxelBlockSemanticGrid2, VoxelBlockSemanticProbabilisticGrid,
                                  VoxelBlockSemanticProbabilisticGrid2, VoxelSemanticData, VoxelSemanticGrid, VoxelSemanticGrid2,
                                  VoxelSemanticGridProbabilistic, VoxelSemanticGridProbabilistic2, check_image_size,
                                  convert_image_type_if_needed, remap_instance_ids)

VoxelGridDataF, ObjectDataF, ObjectDataGroupF, ClassDataF, ClassDataGroupF = VoxelGridData, ObjectData, ObjectDataGroup, ClassData, ClassDataGroup
