"""Mirror of pyslam/dense/volumetric_integrator_voxel_semantic_grid.py over the HIP library's semantic
modes (VOXEL_SEMANTIC_GRID: voting payload; VOXEL_SEMANTIC_PROBABILISTIC_GRID: log-probability payload).

Per keyframe (reference :255-461): rectify -> shadow-point filter -> assign_object_ids_to_instance_ids
(+ carving) -> remap_instance_ids -> depth2pointcloud with labels + world transform -> integrate.  Here every
step after the rectification runs on the GPU; the last three are one fused call (integrate_rgbd).
Outputs (reference :517-700): an object list (per-object points + PCA boxes) when instance ids are integrated,
else one labelled point cloud."""
import time
import os
import traceback

import numpy as np

from .parameters import get_parameters
from .volumetric_integrator_base import (
    TimerFps,
    VolumetricIntegrationOutput,
    VolumetricIntegrationPointCloud,
    VolumetricIntegrationTaskType,
    VolumetricIntegratorBase,
    take_integrate_backlog,
)
from .volumetric_integrator_types import DatasetEnvironmentType

Parameters = get_parameters()

kGenerateObjectsDefault = True  # reference :83: objects representation instead of one point cloud


class IdsColorTable:
    """Stand-in for pyslam.utilities.color_utils.IdsColorTable (outside this path): a fixed pseudo-random
    palette indexed by id; ids < 0 are black.  When real pySLAM is importable its table is used instead."""

    def __init__(self, size=4096, seed=0):
        rng = np.random.default_rng(seed)
        self.table = rng.integers(32, 256, (size, 3)).astype(np.float32) / 255.0
        self.table[0] = 0.5

    def ids_to_rgb_float(self, ids, bgr=False):
        ids = np.asarray(ids, dtype=np.int64)
        out = self.table[np.mod(ids, len(self.table))]
        out = np.where((ids < 0)[..., None], 0.0, out).astype(np.float32)
        return out[..., ::-1] if bgr else out


def _make_color_table():
    try:
        from pyslam.utilities import color_utils  # type: ignore

        return color_utils.IdsColorTable()
    except Exception:
        return IdsColorTable()


class VolumetricIntegratinOrientedBoundingBox3D:  # base.py:245-259 (the reference's spelling)
    def __init__(self, box_matrix, box_size):
        matrix = np.asarray(box_matrix, dtype=np.float64)
        if matrix.shape == (4, 4):
            matrix = matrix.T  # OpenGL consumes column-major matrices
        self.box_matrix = np.ascontiguousarray(matrix, dtype=np.float64)
        self.box_size = np.ascontiguousarray(box_size, dtype=np.float64)


class VolumetricIntegrationObject:  # base.py:262-282
    def __init__(self, object_data):
        self.points = np.ascontiguousarray(object_data.points, dtype=Parameters.kDenseMappingDtypeVertices)
        self.colors = np.ascontiguousarray(object_data.colors, dtype=Parameters.kDenseMappingDtypeColors)
        self.class_id = int(object_data.class_id)
        self.object_id = int(object_data.object_id)
        self.confidence_min = float(object_data.confidence_min)
        self.confidence_max = float(object_data.confidence_max)
        self.oriented_bounding_box = VolumetricIntegratinOrientedBoundingBox3D(
            object_data.oriented_bounding_box.get_matrix(), object_data.oriented_bounding_box.size)


class VolumetricIntegrationObjectList:  # base.py:285-306
    def __init__(self, object_data_group, semantic_colors, object_colors, num_objects):
        self.object_list = [VolumetricIntegrationObject(o) for o in object_data_group.object_vector]
        self.semantic_colors = None if semantic_colors is None else np.ascontiguousarray(semantic_colors, dtype=Parameters.kDenseMappingDtypeColors)
        self.object_colors = None if object_colors is None else np.ascontiguousarray(object_colors, dtype=Parameters.kDenseMappingDtypeColors)
        self.num_objects = int(num_objects)


def _default_semantic_grid(probabilistic, voxel_size, block_size, device, max_blocks, max_points):
    from ..volumetric_semantic import VoxelBlockSemanticGrid, VoxelBlockSemanticProbabilisticGrid

    cls = VoxelBlockSemanticProbabilisticGrid if probabilistic else VoxelBlockSemanticGrid
    return cls(voxel_size=voxel_size, block_size=block_size, device=device, max_blocks=max_blocks, max_points=max_points)


class VolumetricIntegratorVoxelSemanticGrid(VolumetricIntegratorBase):
    def init(self, camera, environment_type, sensor_type, parameters_dict, constructor_kwargs):  # reference :108-206
        VolumetricIntegratorBase.init(self, camera, environment_type, sensor_type, parameters_dict, constructor_kwargs)
        from ..volumetric import CameraFrustrum

        self.ids_color_table = _make_color_table()
        indoor = getattr(environment_type, "name", str(environment_type)) == DatasetEnvironmentType.INDOOR.name
        self.volumetric_integration_depth_trunc = (
            Parameters.kVolumetricIntegrationTsdfDepthTruncIndoor if indoor else Parameters.kVolumetricIntegrationTsdfDepthTruncOutdoor
        )
        # use_voxel_blocks=False (the reference's direct hash VoxelSemanticGrid): same payloads and observable
        # results; on the GPU it is the same block hash underneath
        probabilistic = bool(constructor_kwargs.get("use_semantic_probabilistic", False))
        factory = constructor_kwargs.get("volume_factory", _default_semantic_grid)
        self.volume = factory(probabilistic, Parameters.kVolumetricIntegrationVoxelLength, Parameters.kVolumetricIntegrationBlockSize,
                              Parameters.kVolumetricIntegrationHipDevice, Parameters.kVolumetricIntegrationHipMaxBlocks,
                              max(camera.width * camera.height, 1 << 16))
        carving_depth_max = (Parameters.kVolumetricIntegrationVoxelGridCarvingDepthMaxIndoor if indoor
                             else Parameters.kVolumetricIntegrationVoxelGridCarvingDepthMaxOutdoor)
        fx, fy, cx, cy = self.get_camera_intrinsics_for_depth()
        self.camera_frustrum = CameraFrustrum(fx, fy, cx, cy, camera.width, camera.height, np.eye(4),
                                              depth_max=carving_depth_max,
                                              depth_min=Parameters.kVolumetricIntegrationVoxelGridCarvingDepthMin)
        # reference :170-190 (there: process-wide statics of the payload types; here: per volume)
        if indoor:
            self.volume.set_depth_threshold(Parameters.kVolumetricSemanticProbabilisticIntegrationDepthThresholdIndoor)
            self.volume.set_depth_decay_rate(Parameters.kVolumetricSemanticProbabilisticIntegrationDepthDecayRateIndoor)
        else:
            self.volume.set_depth_threshold(Parameters.kVolumetricSemanticProbabilisticIntegrationDepthThresholdOutdoor)
            self.volume.set_depth_decay_rate(Parameters.kVolumetricSemanticProbabilisticIntegrationDepthDecayRateOutdoor)
        self.dtype_semantics = np.dtype(Parameters.kDenseMappingDtypeSemantics)
        self.dtype_object_ids = np.dtype(Parameters.kDenseMappingDtypeObjectIds)
        self.integrate_2d_instance_ids = False

    # -- colours of labels (SemanticMappingShared.sem_img_to_rgb is outside this path) ---------------
    def _semantic_colors(self, class_ids):
        if class_ids is None or len(class_ids) == 0:
            return None
        try:
            from pyslam.semantics.semantic_mapping_shared import SemanticMappingShared  # type: ignore

            if SemanticMappingShared.is_semantic_mapping_enabled():
                return np.ascontiguousarray(SemanticMappingShared.sem_img_to_rgb(class_ids, bgr=True), dtype=self.dtype_colors) / 255.0
            return None
        except Exception:
            return np.ascontiguousarray(self.ids_color_table.ids_to_rgb_float(class_ids, bgr=True), dtype=self.dtype_colors)

    def _object_colors(self, object_ids, expected):
        if object_ids is None or len(object_ids) == 0:
            return None
        try:
            rgb = self.ids_color_table.ids_to_rgb_float(object_ids, bgr=True)
            if rgb is not None and rgb.ndim == 2 and rgb.shape == (expected, 3):
                return np.ascontiguousarray(rgb, dtype=self.dtype_colors)
        except Exception:
            traceback.print_exc()
        return None

    def _get_voxels(self):
        return self.volume.get_voxels(min_count=Parameters.kVolumetricIntegrationVoxelGridMinCount,
                                      min_confidence=Parameters.kVolumetricIntegrationVoxelGridMinConfidence)

    def integrate_keyframe(self, color, depth, pose, semantic_classes, semantic_instances):
        """The INTEGRATE body, reference :322-461.  color RGB u8, depth f32 metres, pose = T_cw."""
        self.integrate_2d_instance_ids = bool(
            Parameters.kVolumetricSemanticIntegrationUseInstanceIds and semantic_instances is not None
            and np.asarray(semantic_instances).size > 0)
        if self._device_flow():
            # One upload per image: the keyframe's depth / colour / label images go to HBM once (torch tensors) and every step
            # of the body - shadow filter, association, remap, fused integrate - works on them in place, instead of each call
            # staging its inputs again and the filter / remap results making a round trip through host memory (19 -> 5 copies
            # per keyframe in profiles/r03/kernel_stats_semantic.csv).
            return self._integrate_keyframe_on_device(color, depth, pose, semantic_classes, semantic_instances)
        depth_filtered = depth
        if Parameters.kVolumetricIntegrationVoxelGridShadowPointsFilter:
            depth_filtered = self.volume.filter_shadow_points(depth)  # depth.py:103-146 on the GPU
        depth_filtered = np.ascontiguousarray(depth_filtered, dtype=self.dtype_depths)
        self.camera_frustrum.set_T_cw(pose)
        object_ids_image = None
        if self.integrate_2d_instance_ids:
            id_map = self.volume.assign_object_ids_to_instance_ids(
                self.camera_frustrum, semantic_classes, semantic_instances, depth_filtered,
                depth_threshold=Parameters.kVolumetricIntegrationVoxelGridCarvingDepthThreshold,
                do_carving=Parameters.kVolumetricIntegrationVoxelGridUseCarving,
                min_vote_ratio=Parameters.kVolumetricSemanticIntegrationMinVoteRatio,
                min_votes=Parameters.kVolumetricSemanticIntegrationMinVotes)
            object_ids_image = self.volume.remap_instance_ids(np.ascontiguousarray(semantic_instances, dtype=np.int32), id_map)
        elif Parameters.kVolumetricIntegrationVoxelGridUseCarving:
            self.volume.carve(self.camera_frustrum, depth_filtered, Parameters.kVolumetricIntegrationVoxelGridCarvingDepthThreshold)
        fx, fy, cx, cy = self.get_camera_intrinsics_for_depth()
        self.volume.integrate_rgbd(depth_filtered, color, fx, fy, cx, cy, pose, class_ids_image=semantic_classes,
                                   object_ids_image=object_ids_image, max_depth=self.volumetric_integration_depth_trunc,
                                   use_depths=Parameters.kVolumetricSemanticProbabilisticIntegrationUseDepth)

    def _device_flow(self):
        """True when the volume is the HIP one (tests swap in oracle stand-ins that only take numpy) and torch sees a GPU."""
        if os.environ.get("PYSLAM_AMD_SEMANTIC_DEVICE_FLOW", "1") == "0":
            return False
        try:
            import torch

            from ..volumetric_semantic import _SemanticGridBase

            return isinstance(self.volume, _SemanticGridBase) and torch.cuda.is_available()
        except Exception:
            return False

    def _keyframe_arrays(self, color, depth, semantic_classes, semantic_instances):
        """What a keyframe uploads: name -> (host array | None, dtype)."""
        has_cls = semantic_classes is not None and np.asarray(semantic_classes).size > 0
        use_inst = bool(Parameters.kVolumetricSemanticIntegrationUseInstanceIds and semantic_instances is not None
                        and np.asarray(semantic_instances).size > 0)
        return {"depth": (depth, np.float32), "color": (color, np.uint8), "cls": (semantic_classes if has_cls else None, np.int32),
                "inst": (semantic_instances if use_inst else None, np.int32)}

    def _integrate_keyframe_on_device(self, color, depth, pose, semantic_classes, semantic_instances):
        self.integrate_keyframes_on_device([(color, depth, pose, semantic_classes, semantic_instances)])

    def integrate_keyframes_on_device(self, keyframes):
        """keyframes: list of (color RGB u8, depth f32, T_cw, class image | None, instance image | None), fused in this order; the
        images of keyframe k + 1 cross PCIe while keyframe k is fused (device_pipeline.KeyframeUploader)."""
        from .device_pipeline import KeyframeUploader

        if getattr(self, "_uploader", None) is None:
            self._uploader = KeyframeUploader(self.volume)

        shadow = bool(Parameters.kVolumetricIntegrationVoxelGridShadowPointsFilter)

        def prep(t, stream):  # upload side: depends on the keyframe's depth alone, runs beside the previous keyframe's kernels
            t["depth"] = self.volume.filter_shadow_points(t["depth"], stream=stream)

        def body(kf, t):
            _, _, pose, _, _ = kf
            self.integrate_2d_instance_ids = t["inst"] is not None
            self._fuse_device_keyframe(t["color"], t["depth"], pose, t["cls"], t["inst"], depth_filtered=shadow)

        self._uploader.run(keyframes, lambda kf: self._keyframe_arrays(kf[0], kf[1], kf[3], kf[4]), body, prep if shadow else None)

    def _fuse_device_keyframe(self, color_d, depth_d, pose, cls_d, inst_d, depth_filtered=False):
        """The INTEGRATE body on device-resident images (torch CUDA tensors on the volume's stream): nothing waits for the GPU."""
        vol = self.volume
        if hasattr(vol, "fuse_keyframe") and getattr(vol, "_pair_exchange", None) is None and getattr(vol, "_pair_exchange_device", None) is None:
            # one call into the library for the whole body (hv_semantic_fuse_keyframe: the same stages in the same order)
            fx, fy, cx, cy = self.get_camera_intrinsics_for_depth()
            vol.fuse_keyframe(self.camera_frustrum, depth_d, color_d, cls_d, inst_d, fx, fy, cx, cy, pose,
                              filter_shadow_points=bool(Parameters.kVolumetricIntegrationVoxelGridShadowPointsFilter),
                              use_instance_ids=inst_d is not None,
                              depth_threshold=Parameters.kVolumetricIntegrationVoxelGridCarvingDepthThreshold,
                              do_carving=Parameters.kVolumetricIntegrationVoxelGridUseCarving,
                              min_vote_ratio=Parameters.kVolumetricSemanticIntegrationMinVoteRatio,
                              min_votes=Parameters.kVolumetricSemanticIntegrationMinVotes,
                              max_depth=self.volumetric_integration_depth_trunc,
                              use_depths=Parameters.kVolumetricSemanticProbabilisticIntegrationUseDepth, depth_is_filtered=depth_filtered)
            return
        if Parameters.kVolumetricIntegrationVoxelGridShadowPointsFilter and not depth_filtered:
            depth_d = self.volume.filter_shadow_points(depth_d)  # stays in HBM
        self.camera_frustrum.set_T_cw(pose)
        object_ids_d = None
        if inst_d is not None:  # same branches as the host flow (no class image: empty map, every id -> -1)
            id_map = self.volume.assign_object_ids_to_instance_ids(
                self.camera_frustrum, cls_d, inst_d, depth_d,
                depth_threshold=Parameters.kVolumetricIntegrationVoxelGridCarvingDepthThreshold,
                do_carving=Parameters.kVolumetricIntegrationVoxelGridUseCarving,
                min_vote_ratio=Parameters.kVolumetricSemanticIntegrationMinVoteRatio,
                min_votes=Parameters.kVolumetricSemanticIntegrationMinVotes)
            object_ids_d = self.volume.remap_instance_ids(inst_d, id_map)
        elif Parameters.kVolumetricIntegrationVoxelGridUseCarving:
            self.volume.carve(self.camera_frustrum, depth_d, Parameters.kVolumetricIntegrationVoxelGridCarvingDepthThreshold)
        fx, fy, cx, cy = self.get_camera_intrinsics_for_depth()
        self.volume.integrate_rgbd(depth_d, color_d, fx, fy, cx, cy, pose, class_ids_image=cls_d, object_ids_image=object_ids_d,
                                   max_depth=self.volumetric_integration_depth_trunc,
                                   use_depths=Parameters.kVolumetricSemanticProbabilisticIntegrationUseDepth)

    def make_output(self, task_type):
        """The output block, reference :511-700."""
        pc_out = objects_out = None
        if kGenerateObjectsDefault and self.integrate_2d_instance_ids:
            group = self.volume.get_object_segments(min_count=Parameters.kVolumetricIntegrationVoxelGridMinCount,
                                                    min_confidence=Parameters.kVolumetricIntegrationVoxelGridMinConfidence)
            n = len(group.object_vector)
            objects_out = VolumetricIntegrationObjectList(group, self._semantic_colors(group.class_ids),
                                                          self._object_colors(group.object_ids, n), n)
        else:
            data = self._get_voxels()
            points = np.ascontiguousarray(data.points, dtype=self.dtype_vertices)
            colors = np.ascontiguousarray(data.colors, dtype=self.dtype_colors)
            semantics = np.ascontiguousarray(data.class_ids, dtype=self.dtype_semantics) if len(data.class_ids) > 0 else None
            object_ids = np.ascontiguousarray(data.object_ids, dtype=self.dtype_object_ids) if len(data.object_ids) > 0 else None
            pc_out = VolumetricIntegrationPointCloud(points=points, colors=colors, semantics=semantics, object_ids=object_ids,
                                                     semantic_colors=self._semantic_colors(semantics),
                                                     object_colors=self._object_colors(object_ids, len(points)))
        return VolumetricIntegrationOutput(task_type, self.last_integrated_id, pc_out, None, objects_out)

    def volume_integration(self, q_in, q_out, q_out_condition, q_management, viewer_queue, is_running,
                           load_request_completed, load_request_condition, save_request_completed,
                           save_request_condition, time_volumetric_integration):  # reference :208-748
        last_output = None
        do_output = False
        timer = TimerFps("VolumetricIntegratorVoxelSemanticGrid")
        timer.start()
        try:
            if is_running.value == 1:
                self.last_management_task = None
                try:
                    self.last_management_task = q_management.get_nowait()
                except Exception:
                    pass
                if (self.last_management_task is not None
                        and self.last_management_task.task_type == VolumetricIntegrationTaskType.RESET):
                    self.volume.reset()
                    self.last_output = None  # reference :243-246
                    self.last_integrated_id = -1
                try:
                    self.last_input_task = q_in.get(timeout=0.5)
                except Exception:
                    return
                if self.last_input_task is None:
                    is_running.value = 0
                else:
                    ttype = self.last_input_task.task_type
                    if ttype == VolumetricIntegrationTaskType.INTEGRATE:
                        # a backlog (offline reconstruction, rebuild()) is fused in queue order with the uploads of keyframe k + 1
                        # beside the kernels of keyframe k; a single keyframe takes the same path
                        tasks = [self.last_input_task]
                        if self._device_flow():
                            tasks += take_integrate_backlog(q_in, 7)
                        ready = []
                        for task in tasks:
                            keyframe_data = task.keyframe_data
                            if not Parameters.kVolumetricSemanticIntegrationUseInstanceIds:
                                keyframe_data.semantic_instances_img = None
                            color, depth, _, sem_cls, sem_inst = self.estimate_depth_if_needed_and_rectify(keyframe_data)
                            if color is not None and depth is not None:
                                ready.append((color, depth, keyframe_data.pose, sem_cls, sem_inst, keyframe_data.id))
                        if ready and self._device_flow():
                            self.integrate_keyframes_on_device([r[:5] for r in ready])
                        else:
                            for color, depth, pose, sem_cls, sem_inst, _ in ready:
                                self.integrate_keyframe(color, depth, pose, sem_cls, sem_inst)
                        if ready:
                            self.last_integrated_id = ready[-1][5]
                            do_output = True
                            if self.last_output is not None:
                                if time.perf_counter() - self.last_output.timestamp < Parameters.kVolumetricIntegrationOutputTimeInterval:
                                    do_output = False
                    elif ttype == VolumetricIntegrationTaskType.SAVE:
                        data = self._get_voxels()
                        points = np.ascontiguousarray(data.points, dtype=self.dtype_vertices)
                        colors = np.ascontiguousarray(data.colors, dtype=self.dtype_colors)
                        if len(points) > 0 and len(colors) > 0:
                            self._save_points(self.last_input_task.load_save_path, points, colors)
                        last_output = VolumetricIntegrationOutput(ttype)
                        self.last_output = last_output
                    elif ttype == VolumetricIntegrationTaskType.UPDATE_OUTPUT:
                        do_output = True
                    if do_output:
                        last_output = self.make_output(ttype)
                        self.last_output = last_output
                    self._publish(last_output, q_out, q_out_condition, is_running, save_request_completed, save_request_condition)
        except Exception:
            traceback.print_exc()
        timer.refresh()
        time_volumetric_integration.value = timer.last_elapsed
