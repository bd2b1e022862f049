"""pyslam/dense/volumetric_integrator_factory.py:58-150."""
from .parameters import get_parameters
from .volumetric_integrator_tsdf import VolumetricIntegratorTsdf
from .volumetric_integrator_types import VolumetricIntegratorType
from .volumetric_integrator_voxel_grid import VolumetricIntegratorVoxelGrid
from .volumetric_integrator_voxel_semantic_grid import VolumetricIntegratorVoxelSemanticGrid

Parameters = get_parameters()


def volumetric_integrator_factory(volumetric_integrator_type, camera, environment_type, sensor_type, viewer_queue=None,
                                  **kwargs):
    """Same signature and associations as the reference factory (:71-86):
        VOXEL_GRID                        -> VolumetricIntegratorVoxelGrid
        VOXEL_SEMANTIC_GRID               -> VolumetricIntegratorVoxelSemanticGrid
        VOXEL_SEMANTIC_PROBABILISTIC_GRID -> VolumetricIntegratorVoxelSemanticGrid(use_semantic_probabilistic=True)
        TSDF                              -> VolumetricIntegratorTsdf
    GAUSSIAN_SPLATTING is outside the hot path this package replaces."""
    name = getattr(volumetric_integrator_type, "name", str(volumetric_integrator_type))
    # reference :88-110: with sparse semantic mapping on, a plain VOXEL_GRID request is upgraded
    if getattr(Parameters, "kDoSparseSemanticMappingAndSegmentation", False) and name == VolumetricIntegratorType.VOXEL_GRID.name:
        volumetric_integrator_type = VolumetricIntegratorType.VOXEL_SEMANTIC_PROBABILISTIC_GRID
        name = volumetric_integrator_type.name
    common = dict(camera=camera, environment_type=environment_type, sensor_type=sensor_type,
                  volumetric_integrator_type=volumetric_integrator_type, viewer_queue=viewer_queue)
    if name == VolumetricIntegratorType.VOXEL_GRID.name:
        return VolumetricIntegratorVoxelGrid(use_voxel_blocks=Parameters.kVolumetricIntegrationUseVoxelBlocks, **common, **kwargs)
    if name == VolumetricIntegratorType.VOXEL_SEMANTIC_GRID.name:
        return VolumetricIntegratorVoxelSemanticGrid(use_voxel_blocks=Parameters.kVolumetricIntegrationUseVoxelBlocks, **common, **kwargs)
    if name == VolumetricIntegratorType.VOXEL_SEMANTIC_PROBABILISTIC_GRID.name:
        return VolumetricIntegratorVoxelSemanticGrid(use_semantic_probabilistic=True,
                                                     use_voxel_blocks=Parameters.kVolumetricIntegrationUseVoxelBlocks, **common, **kwargs)
    if name == VolumetricIntegratorType.TSDF.name:
        return VolumetricIntegratorTsdf(**common, **kwargs)
    raise ValueError(f"Invalid VolumetricIntegratorType: {name} (GPU path: VOXEL_GRID, VOXEL_SEMANTIC_GRID, "
                     f"VOXEL_SEMANTIC_PROBABILISTIC_GRID, TSDF)")
