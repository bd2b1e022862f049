"""pyslam/dense/volumetric_integrator_types.py:8-28."""
from enum import Enum


class VolumetricIntegratorType(Enum):
    VOXEL_GRID = 0
    VOXEL_SEMANTIC_GRID = 1
    VOXEL_SEMANTIC_PROBABILISTIC_GRID = 2
    TSDF = 3
    GAUSSIAN_SPLATTING = 4

    @staticmethod
    def from_string(name: str):
        try:
            return VolumetricIntegratorType[name]
        except KeyError:
            raise ValueError(f"Invalid VolumetricIntegratorType: {name}")


class DatasetEnvironmentType(Enum):
    """pyslam.io.dataset_types.DatasetEnvironmentType (only the two values the dense path reads)."""
    INDOOR = 1
    OUTDOOR = 2


class SensorType(Enum):
    """pyslam.io.dataset_types.SensorType."""
    MONOCULAR = 0
    STEREO = 1
    RGBD = 2
