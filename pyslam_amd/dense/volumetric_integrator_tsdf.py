"""Mirror of pyslam/dense/volumetric_integrator_tsdf.py over the HIP library's TSDF mode."""
import time
import traceback

import numpy as np

from .parameters import get_parameters
from .volumetric_integrator_base import (
    TimerFps,
    VolumetricIntegrationMesh,
    VolumetricIntegrationOutput,
    VolumetricIntegrationPointCloud,
    VolumetricIntegrationTaskType,
    VolumetricIntegratorBase,
    take_integrate_backlog,
)
from .volumetric_integrator_types import DatasetEnvironmentType

Parameters = get_parameters()


def _default_tsdf_volume(voxel_length, sdf_trunc, device, max_blocks, max_points):
    from ..volumetric import ScalableTSDFVolume

    return ScalableTSDFVolume(voxel_length=voxel_length, sdf_trunc=sdf_trunc, device=device, max_blocks=max_blocks,
                              max_points=max_points)


class VolumetricIntegratorTsdf(VolumetricIntegratorBase):
    def init(self, camera, environment_type, sensor_type, parameters_dict, constructor_kwargs):  # tsdf.py:86-119
        VolumetricIntegratorBase.init(self, camera, environment_type, sensor_type, parameters_dict, constructor_kwargs)
        from ..volumetric import PinholeCameraIntrinsic

        indoor = getattr(environment_type, "name", str(environment_type)) == DatasetEnvironmentType.INDOOR.name
        self.volumetric_integration_depth_trunc = (
            Parameters.kVolumetricIntegrationTsdfDepthTruncIndoor if indoor else Parameters.kVolumetricIntegrationTsdfDepthTruncOutdoor
        )
        factory = constructor_kwargs.get("volume_factory", _default_tsdf_volume)
        self.volume = factory(Parameters.kVolumetricIntegrationVoxelLength, Parameters.kVolumetricIntegrationTSdfTrunc,
                              Parameters.kVolumetricIntegrationHipDevice, Parameters.kVolumetricIntegrationHipMaxBlocks,
                              max(camera.width * camera.height, 1 << 16))
        # keyframes arrive B, G, R (OpenCV); the HIP volume swaps the channels while it packs its frame records, so the
        # reference's per-keyframe cv2.cvtColor (base.py:1054) - a strided 0.9 MB host copy - disappears and a keyframe's
        # colour plane can be DMA'd straight from its shared-memory slot
        self.volume_takes_bgr = False
        if hasattr(self.volume, "set_color_order"):
            self.volume.set_color_order(bgr=True)
            self.volume_takes_bgr = True
        # a distorted camera (TUM1's k1..k3): the reference remaps colour and depth of every keyframe on the host with cv2.remap
        # (base.py:1017-1043); the HIP volume takes the maps once and remaps each batch on the device, beside its touch + pack launch
        self.volume_rectifies = False
        if self.calib_map1 is not None and hasattr(self.volume, "set_rectify_maps"):
            self.volume.set_rectify_maps(self.calib_map1, self.calib_map2)
            self.volume_rectifies = True
        fx, fy, cx, cy = self.get_camera_intrinsics_for_depth()
        self.o3d_camera = PinholeCameraIntrinsic(width=camera.width, height=camera.height, fx=fx, fy=fy, cx=cx, cy=cy)

    def volume_integration(self, q_in, q_out, q_out_condition, q_management, viewer_queue, is_running,
                           load_request_completed, load_request_condition, save_request_completed,
                           save_request_condition, time_volumetric_integration):  # tsdf.py:121-314
        from ..volumetric import RGBDImage

        last_output = None
        do_output = False
        timer = TimerFps("VolumetricIntegratorTsdf")
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
                try:
                    self.last_input_task = q_in.get(timeout=0.5)
                except Exception:
                    return
                if self.last_input_task is None:
                    is_running.value = 0
                else:
                    ttype = self.last_input_task.task_type
                    if ttype == VolumetricIntegrationTaskType.INTEGRATE:
                        # Backlog (offline reconstruction, rebuild() after loop closure): drain the queued
                        # INTEGRATE tasks and fuse them with one multi-frame sweep (hv_tsdf_integrate_batch);
                        # same result as fusing them one by one in this order.
                        tasks = [self.last_input_task] + take_integrate_backlog(q_in, 63)
                        frames = []
                        for task in tasks:
                            color, depth, _, _, _ = self.estimate_depth_if_needed_and_rectify(task.keyframe_data)
                            if depth is not None:
                                frames.append((color, depth, task.keyframe_data.pose, task.keyframe_data.id))
                        on_host = all(isinstance(f[1], np.ndarray) for f in frames)  # estimator depth may live in HBM
                        if (on_host and len(frames) > 1 and hasattr(self.volume, "integrate_frames")
                                and len({f[1].shape for f in frames}) == 1):
                            # one pointer per frame: the library stages the keyframes' own arrays (no np.stack copy)
                            self.volume.integrate_frames([f[1] for f in frames], [f[0] for f in frames],
                                                         self.o3d_camera, np.stack([f[2] for f in frames]),
                                                         depth_scale=self.depth_factor,
                                                         depth_trunc=self.volumetric_integration_depth_trunc)
                        else:
                            for color, depth, pose, _ in frames:
                                rgbd = RGBDImage.create_from_color_and_depth(
                                    color, depth, depth_scale=self.depth_factor,
                                    depth_trunc=self.volumetric_integration_depth_trunc, convert_rgb_to_intensity=False)
                                self.volume.integrate(rgbd, self.o3d_camera, pose)  # pose = Tcw
                        if frames:
                            self.last_integrated_id = frames[-1][3]
                            do_output = True
                            if self.last_output is not None:
                                if time.perf_counter() - self.last_output.timestamp < Parameters.kVolumetricIntegrationOutputTimeInterval:
                                    do_output = False
                    elif ttype == VolumetricIntegrationTaskType.SAVE:
                        save_path = self.last_input_task.load_save_path
                        if Parameters.kVolumetricIntegrationTsdfExtractMesh:
                            self._save_mesh(save_path, self.volume.extract_triangle_mesh())
                        else:
                            # o3d.io.write_point_cloud stores the normals Open3D's extract_point_cloud() attaches (tsdf.py:246-247)
                            try:
                                pc = self.volume.extract_point_cloud(normals=True)
                            except TypeError:  # a volume stand-in without normals (tests)
                                pc = self.volume.extract_point_cloud()
                            self._save_points(save_path, pc.points, pc.colors, getattr(pc, "normals", None))
                        last_output = VolumetricIntegrationOutput(ttype)
                    elif ttype == VolumetricIntegrationTaskType.UPDATE_OUTPUT:
                        do_output = True
                    if do_output:
                        mesh_out, pc_out = None, None
                        kw = {}
                        if (getattr(Parameters, "kVolumetricIntegrationTsdfOutputInDenseMappingDtype", False)
                                and self.dtype_vertices == np.float32 and self.dtype_colors == np.float32):
                            kw["dtype"] = np.float32  # (opt-in: float64 rounded once on the device instead of by the consumer)
                        if Parameters.kVolumetricIntegrationTsdfExtractMesh:
                            mesh_out = VolumetricIntegrationMesh(self.volume.extract_triangle_mesh(**kw))
                        else:
                            pc_out = VolumetricIntegrationPointCloud(self.volume.extract_point_cloud(**kw))
                        last_output = VolumetricIntegrationOutput(ttype, self.last_integrated_id, pc_out, mesh_out)
                        self.last_output = last_output
                    self._publish(last_output, q_out, q_out_condition, is_running, save_request_completed, save_request_condition)
        except Exception:
            traceback.print_exc()
        timer.refresh()
        time_volumetric_integration.value = timer.last_elapsed
