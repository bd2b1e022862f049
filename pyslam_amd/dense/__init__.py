"""Drop-in mirror of pySLAM's ``pyslam/dense`` front (reference: pyslam/dense/*.py).

Same class, method, enum and field names as the reference so that main_map_dense_reconstruction.py
and Slam can use it unchanged; the worker process owns the GPU context and
keeps the volume resident in HBM.
"""
from .volumetric_integrator_types import VolumetricIntegratorType  # noqa: F401
from .volumetric_integrator_base import (  # noqa: F401
    VolumetricIntegrationKeyframeData,
    VolumetricIntegrationMesh,
    VolumetricIntegrationOutput,
    VolumetricIntegrationPointCloud,
    VolumetricIntegrationTask,
    VolumetricIntegrationTaskType,
    VolumetricIntegratorBase,
)
from .volumetric_integrator_factory import volumetric_integrator_factory  # noqa: F401
from .volumetric_integrator_tsdf import VolumetricIntegratorTsdf  # noqa: F401
from .volumetric_integrator_voxel_grid import VolumetricIntegratorVoxelGrid  # noqa: F401
from .volumetric_integrator_voxel_semantic_grid import (  # noqa: F401
    VolumetricIntegrationObject,
    VolumetricIntegrationObjectList,
    VolumetricIntegratorVoxelSemanticGrid,
)
