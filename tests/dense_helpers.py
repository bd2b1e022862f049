"""Shared helpers for the dense-front tests: synthetic keyframes/cameras and (CPU only) oracle-backed
stand-ins for the GPU volumes so the process/queue protocol can be exercised without a GPU."""
import types

import numpy as np


class FakeCamera:
    def __init__(self, s):
        self.fx, self.fy, self.cx, self.cy = s.intrinsics
        self.width, self.height = s.width, s.height
        self.D = np.zeros(5)
        self.depth_factor = 1.0


class FakeKeyFrame:
    """The KeyFrame fields pyslam/dense consumes (volumetric_integrator_base.py:112-137,1161,1303-1305)."""

    def __init__(self, i, stream, camera, lba_count=1):
        depth, rgb, T = stream[i]
        self.id = self.kid = self.img_id = i
        self.timestamp = float(i) / 30.0
        self._pose = T
        self.camera = camera
        self.img = np.ascontiguousarray(rgb[..., ::-1])  # pySLAM hands BGR
        self.img_right = None
        self.depth_img = depth
        self.semantic_img = None
        self.semantic_instances_img = None
        self.lba_count = lba_count

    def pose(self):
        return self._pose

    def is_bad(self):
        return False

    def is_semantics_available(self):
        return True


class FakeMap:
    def __init__(self, keyframes):
        self.keyframes = keyframes


class OracleTsdfVolume:
    """ScalableTSDFVolume protocol on top of oracle.PortTsdf (tests only)."""

    def __init__(self, voxel_length, sdf_trunc):
        import oracle

        self.vol = oracle.PortTsdf(voxel_length, sdf_trunc)

    def integrate(self, image, intrinsic, extrinsic):
        self.vol.integrate(image.depth, image.color, intrinsic.as_array(), extrinsic, image.depth_scale, image.depth_trunc)

    def reset(self):
        self.vol.reset()

    def extract_triangle_mesh(self):
        v, t, c = self.vol.extract_triangle_mesh()
        return types.SimpleNamespace(vertices=v, triangles=t, vertex_colors=c, vertex_normals=np.zeros((0, 3)))

    def extract_point_cloud(self):
        p, c = self.vol.extract_point_cloud()
        return types.SimpleNamespace(points=p, colors=c)


class OracleVoxelGrid:
    """VoxelBlockGrid protocol on top of oracle.PortGrid + host_prep (tests only)."""

    def __init__(self, voxel_size, block_size):
        import oracle

        self.grid = oracle.PortGrid(voxel_size, block_size)

    def filter_shadow_points(self, depth):
        from oracle import host_prep

        return host_prep.filter_shadow_points(depth)

    def integrate_rgbd(self, depth, rgb, fx, fy, cx, cy, T_cw, max_depth=np.inf, min_depth=0.0):
        from oracle import host_prep

        p, c, _ = host_prep.frame_to_world_f32(depth, rgb, fx, fy, cx, cy, T_cw, max_depth, min_depth)
        self.grid.integrate(p, c)

    def carve(self, *a):
        pass

    def get_voxels(self, min_count=1, min_confidence=0.0):
        p, c = self.grid.get_voxels(min_count, min_confidence)
        return types.SimpleNamespace(points=p, colors=c)

    def reset(self):
        self.grid.clear()


def oracle_tsdf_factory(voxel_length, sdf_trunc, device, max_blocks, max_points):
    return OracleTsdfVolume(voxel_length, sdf_trunc)


def oracle_grid_factory(voxel_size, block_size, device, max_blocks, max_points):
    return OracleVoxelGrid(voxel_size, block_size)
