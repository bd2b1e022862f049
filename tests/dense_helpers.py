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

    def __init__(self, i, stream, camera, lba_count=1, semantic=False):
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
        if semantic:
            from tests.semantic_helpers import semantic_frame

            _, _, _, self.semantic_img, self.semantic_instances_img = semantic_frame(stream, i)
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

    def integrate_rgbd_batch(self, depth, rgb, fx, fy, cx, cy, T_cw, max_depth=np.inf, min_depth=0.0):
        for f in range(len(depth)):
            self.integrate_rgbd(depth[f], rgb[f], fx, fy, cx, cy, T_cw[f], max_depth, min_depth)

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


class OracleSemanticGrid:
    """VoxelBlockSemantic(Probabilistic)Grid protocol on top of the compiled reference (tests only)."""

    def __init__(self, probabilistic, voxel_size, block_size):
        from oracle.semantic import RefSemGrid2

        self.grid = RefSemGrid2(1 if probabilistic else 0, voxel_size, block_size)

    def set_depth_threshold(self, t):
        self.grid.set_depth_threshold(t)

    def set_depth_decay_rate(self, r):
        self.grid.set_depth_decay_rate(r)

    def filter_shadow_points(self, depth):
        from oracle import host_prep

        return host_prep.filter_shadow_points(depth)

    def assign_object_ids_to_instance_ids(self, fr, class_img, inst_img, depth=None, depth_threshold=0.1, do_carving=False,
                                          min_vote_ratio=0.5, min_votes=3):
        return self.grid.assign_object_ids_to_instance_ids(fr.intr, fr.width, fr.height, fr.T_cw, fr.depth_max, fr.depth_min, class_img,
                                                           inst_img, depth, depth_threshold, do_carving, min_vote_ratio, min_votes)

    def remap_instance_ids(self, inst_img, mapping):
        from oracle.semantic import ref_remap_instance_ids

        return ref_remap_instance_ids(inst_img, mapping)

    def carve(self, fr, depth, thr):
        self.grid.carve(fr.intr, fr.width, fr.height, fr.T_cw, fr.depth_max, fr.depth_min, depth, thr)

    def integrate_rgbd(self, depth, rgb, fx, fy, cx, cy, T_cw, class_ids_image=None, object_ids_image=None, max_depth=np.inf,
                       min_depth=0.0, use_depths=True):
        from oracle import host_prep

        pts_c, cols, valid = host_prep.depth2pointcloud(depth, rgb, fx, fy, cx, cy, max_depth, min_depth)
        pw = np.ascontiguousarray(host_prep.world_points(pts_c, T_cw), dtype=np.float32)
        cls = None if class_ids_image is None else np.ascontiguousarray(class_ids_image[valid], dtype=np.int32)
        obj = None if object_ids_image is None else np.ascontiguousarray(object_ids_image[valid], dtype=np.int32)
        dep = np.ascontiguousarray(pts_c[:, 2], dtype=np.float32) if use_depths else None
        self.grid.integrate(pw, np.ascontiguousarray(cols, dtype=np.float32), cls, obj, dep)

    def get_voxels(self, min_count=1, min_confidence=0.0):
        p, c, cls, obj, conf = self.grid.get_voxels(min_count, min_confidence)
        return types.SimpleNamespace(points=p, colors=c, class_ids=cls, object_ids=obj, confidences=conf)

    def get_object_segments(self, min_count=1, min_confidence=0.0):
        from pyslam_amd.volumetric_semantic import ObjectData, ObjectDataGroup

        class Box:
            def __init__(self, obb):
                self.center, self.orientation, self.size = obb[0:3], obb[3:7], obb[7:10]

            def get_matrix(self):
                w, x, y, z = self.orientation
                M = np.eye(4)
                M[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                             [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                             [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
                M[:3, 3] = self.center
                return M

        return ObjectDataGroup([ObjectData(o["points"], o["colors"], o["object_id"], o["class_id"], o["conf_min"], o["conf_max"],
                                           Box(o["obb"])) for o in self.grid.get_object_segments(min_count, min_confidence)])

    def reset(self):
        self.grid.clear()


def oracle_semantic_factory(probabilistic, voxel_size, block_size, device, max_blocks, max_points):
    return OracleSemanticGrid(probabilistic, voxel_size, block_size)
