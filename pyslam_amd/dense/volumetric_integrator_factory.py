"""pyslam/dense/volumetric_integrator_factory.py:58-150."""
from .parameters import get_parameters
from .volumetric_integrator_tsdf import VolumetricIntegratorTsdf
from .volumetric_integrator_types import VolumetricIntegratorType
from .volumetric_integrator_voxel_grid import VolumetricIntegratorVoxelGrid

Parameters = get_parameters()


def volumetric_integrator_factory(volumetric_integrator_type, camera, environment_type, sensor_type, viewer_queue=None,
                                  **kwargs):
    """Same signature and associations as the reference factory.  Semantic grids and Gaussian
    splatting are outside the hot path this package replaces."""
    name = getattr(volumetric_integrator_type, "name", str(volumetric_integrator_type))
    if name == VolumetricIntegratorType.VOXEL_GRID.name:
        return VolumetricIntegratorVoxelGrid(camera=camera, environment_type=environment_type, sensor_type=sensor_type,
                                             volumetric_integrator_type=volumetric_integrator_type,
                                             use_voxel_blocks=Parameters.kVolumetricIntegrationUseVoxelBlocks,
                                             viewer_queue=viewer_queue, **kwargs)
    if name == VolumetricIntegratorType.TSDF.name:
        return VolumetricIntegratorTsdf(camera=camera, environment_type=environment_type, sensor_type=sensor_type,
                                        volumetric_integrator_type=volumetric_integrator_type,
                                        viewer_queue=viewer_queue, **kwargs)
    raise ValueError(f"Invalid VolumetricIntegratorType: {name} (GPU path provides VOXEL_GRID and TSDF)")
