#!/usr/bin/env python3
"""Secondary benchmark: throughput of the path the drop-in REALLY drives.  pySLAM hands every keyframe to the integrator as
pageable host numpy arrays (pyslam/dense/volumetric_integrator_base.py:101-137, volumetric_integrator_tsdf.py:215-223);
bench.py's headline starts its clock with the frames already in HBM.  Three figures, one JSON object:

  staged        ScalableTSDFVolume.integrate_frames on the headline's sliding stream, frames given as SEPARATE pageable numpy
                arrays (one per keyframe, exactly what pyslam_amd/dense/volumetric_integrator_tsdf.py passes): caller memory ->
                page-locked slots (worker threads) -> DMA on a copy stream -> touch + pack + sweep; PCIe-inclusive frames/s,
                next to what the H2D rate measured in the same process allows (bytes per frame / pinned-memory H2D GB/s).
  online_host   one integrate() per frame from host arrays (the live SLAM flow: a keyframe every few hundred ms).
  front         the whole front once: add_keyframe -> multiprocessing queue -> worker process -> integrate_frames -> mesh
                extraction -> pop_output (pickling both ways included) for a short run at the same configuration.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _Camera:
    """The camera fields pyslam/dense reads (volumetric_integrator_base.py:758-786)."""

    def __init__(self, s):
        self.fx, self.fy, self.cx, self.cy = s.intrinsics
        self.width, self.height = s.width, s.height
        self.D = np.zeros(5)
        self.depth_factor = 1.0


class _KeyFrame:
    """The KeyFrame fields pyslam/dense consumes (volumetric_integrator_base.py:112-137)."""

    def __init__(self, i, depth, rgb, T, camera):
        self.id = self.kid = self.img_id = i
        self.timestamp = float(i) / 30.0
        self._pose = T
        self.camera = camera
        self.img = np.ascontiguousarray(rgb[..., ::-1])  # pySLAM hands BGR
        self.img_right = None
        self.depth_img = depth
        self.semantic_img = None
        self.semantic_instances_img = None
        self.lba_count = 1

    def pose(self):
        return self._pose

    def is_bad(self):
        return False

    def is_semantics_available(self):
        return False


def h2d_rate_gbs(n_bytes=256 << 20, reps=5):
    """Pinned-memory H2D rate of this host / GPU pair, measured with one large asynchronous copy per repetition."""
    import torch

    src = torch.empty(n_bytes, dtype=torch.uint8, pin_memory=True)
    dst = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
    dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        best = max(best, n_bytes / (time.perf_counter() - t0) / 1e9)
    return best


def host_leg(s, depth_h, rgb_h, T_h, voxel, sdf_trunc, depth_trunc, B=32, steps=6, front_frames=48, front=True):
    import torch

    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume

    n = min(len(depth_h), steps * B)
    steps = n // B
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    # one allocation per keyframe, pageable: what arrives through pySLAM's queue
    depths = [np.array(depth_h[i]) for i in range(n)]
    colors = [np.array(rgb_h[i]) for i in range(n)]
    frame_bytes = depths[0].nbytes + colors[0].nbytes
    out = {"what": "the sliding stream handed over as pageable per-keyframe numpy arrays (PCIe inside the timed region)",
           "bytes_per_frame": int(frame_bytes), "frames_per_call": B}
    rate = h2d_rate_gbs()
    out["h2d_pinned_GBs"] = round(rate, 1)
    out["h2d_bound_frames_per_s"] = round(rate * 1e9 / frame_bytes, 1)
    vol = ScalableTSDFVolume(voxel, sdf_trunc, max_blocks=1 << 17, max_points=s.width * s.height)

    def run_staged():
        for k in range(steps):
            lo = k * B
            vol.integrate_frames(depths[lo:lo + B], colors[lo:lo + B], K, T_h[lo:lo + B], depth_scale=1.0, depth_trunc=depth_trunc)
        vol.synchronize()
        torch.cuda.synchronize()

    run_staged()  # warm-up: allocates units, page-locks the staging slots, starts the copy threads
    best = None
    for _ in range(3):
        vol.reset()
        vol.synchronize()
        t0 = time.perf_counter()
        run_staged()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    fps = steps * B / best
    out["staged"] = {"value": round(fps, 1), "unit": "frames/s", "ms_per_step": round(best / steps * 1e3, 3),
                     "frac_of_h2d_bound": round(fps / (rate * 1e9 / frame_bytes), 3),
                     "call": "ScalableTSDFVolume.integrate_frames (hv_tsdf_integrate_frames), 32 keyframes per call, volume empty when the clock starts"}
    # the same keyframes with the depth as the SENSOR delivers it: uint16, DepthMapFactor 5000 (settings/TUM1.yaml:54) - 5 bytes per pixel
    # cross PCIe instead of 7; the conversion to metres happens inside the fusion (Image::ConvertDepthToFloatImage)
    depths16 = [np.clip(np.rint(d * 5000.0), 0, 65535).astype(np.uint16) for d in depths]
    fb16 = depths16[0].nbytes + colors[0].nbytes

    def run_staged16():
        for k in range(steps):
            lo = k * B
            vol.integrate_frames(depths16[lo:lo + B], colors[lo:lo + B], K, T_h[lo:lo + B], depth_scale=5000.0, depth_trunc=depth_trunc)
        vol.synchronize()
        torch.cuda.synchronize()

    vol.reset()
    run_staged16()
    best16 = None
    for _ in range(3):
        vol.reset()
        vol.synchronize()
        t0 = time.perf_counter()
        run_staged16()
        dt = time.perf_counter() - t0
        best16 = dt if best16 is None else min(best16, dt)
    fps16 = steps * B / best16
    out["staged"]["bytes_per_frame"] = int(frame_bytes)
    out["staged_u16"] = {"value": round(fps16, 1), "unit": "frames/s", "ms_per_step": round(best16 / steps * 1e3, 3), "bytes_per_frame": int(fb16),
                         "h2d_bound_frames_per_s": round(rate * 1e9 / fb16, 1), "frac_of_h2d_bound": round(fps16 / (rate * 1e9 / fb16), 3),
                         "call": "integrate_frames with uint16 depth, depth_scale 5000: 5 bytes per pixel over PCIe"}
    # the same through ONE contiguous pageable array per batch (integrate_batch(np.stack(...)), the round-2 call), np.stack inside the clock
    vol.reset()
    vol.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        lo = k * B
        vol.integrate_batch(np.stack(depths[lo:lo + B]), np.stack(colors[lo:lo + B]), K, T_h[lo:lo + B], depth_scale=1.0, depth_trunc=depth_trunc)
    vol.synchronize()
    torch.cuda.synchronize()
    out["stacked"] = {"value": round(steps * B / (time.perf_counter() - t0), 1), "unit": "frames/s",
                      "call": "integrate_batch(np.stack(depths), np.stack(colours)): one extra host copy per batch"}
    # online from host arrays
    vol.reset()
    vol.synchronize()
    n_on = min(n, 2 * B)
    t0 = time.perf_counter()
    for i in range(n_on):
        vol.integrate(RGBDImage.create_from_color_and_depth(colors[i], depths[i], 1.0, depth_trunc, False), K, T_h[i])
    vol.synchronize()
    torch.cuda.synchronize()
    out["online_host"] = {"value": round(n_on / (time.perf_counter() - t0), 1), "unit": "frames/s",
                          "call": "one integrate() per keyframe from host arrays (hv_tsdf_integrate, HV_HOST)"}
    del vol
    if not front:
        return out
    # the whole front: keyframes through the shared-memory ring (the default), and - A/B - pickled through the queue like the reference
    try:
        out["front"] = front_leg(s, depths, colors, T_h, voxel, sdf_trunc, n)
    except Exception as e:  # the worker process is the fragile part of a benchmark box: never lose the line for it
        out["front"] = {"error": f"{type(e).__name__}: {e}"}
    try:
        out["front_pickled"] = front_leg(s, depths, colors, T_h, voxel, sdf_trunc, min(front_frames, n), shared=False)
    except Exception as e:
        out["front_pickled"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def front_leg(s, depths, colors, T_h, voxel, sdf_trunc, n_frames, shared=True):
    from pyslam_amd.dense import VolumetricIntegratorType, volumetric_integrator_factory
    from pyslam_amd.dense.parameters import Parameters
    from pyslam_amd.dense.volumetric_integrator_types import DatasetEnvironmentType, SensorType

    Parameters.kVolumetricIntegrationVoxelLength = voxel
    Parameters.kVolumetricIntegrationTSdfTrunc = sdf_trunc
    Parameters.kVolumetricIntegrationOutputTimeInterval = 1.0  # the reference's default: at most one mesh per second while the queue drains
    Parameters.kVolumetricIntegrationHipMaxBlocks = 1 << 16
    # the reference throttles its worker to <= 10 integrate calls/s once more than 10 tasks are queued (base.py:923-938, protecting the
    # SLAM threads from a CPU integrator); a throughput measurement switches that knob off
    Parameters.kVolumetricIntegrationFpsThrottleEnabled = False
    Parameters.kVolumetricIntegrationUseSharedMemory = bool(shared)
    Parameters.kVolumetricIntegrationSharedMemorySlots = 128
    cam = _Camera(s)
    integ = volumetric_integrator_factory(VolumetricIntegratorType.TSDF, cam, DatasetEnvironmentType.INDOOR, SensorType.RGBD)
    try:
        t0 = time.time()
        while not integ.is_ready():
            if time.time() - t0 > 120:
                raise RuntimeError("integrator worker did not start")
            time.sleep(0.02)
        # the stream three times over (new keyframe ids, the same images and poses: re-observing a scene is ordinary fusion work)
        passes = 3 if shared else 1
        kfs = [_KeyFrame(p * n_frames + i, depths[i], colors[i], T_h[i], cam) for p in range(passes) for i in range(n_frames)]
        n_frames = len(kfs)
        # first keyframe alone: the worker's one-off costs (library load, pool, first output)
        integ.add_keyframe(kfs[0], kfs[0].img, None, kfs[0].depth_img)
        integ.add_update_output_task()
        from pyslam_amd.dense import VolumetricIntegrationTaskType

        # both outputs of the first keyframe (its own and the requested one) are consumed before the clock starts
        seen, t0 = 0, time.time()
        while seen < 2 and time.time() - t0 < 120:
            seen += integ.pop_output(timeout=0.5) is not None
        t1 = time.perf_counter()
        for kf in kfs[1:]:
            integ.add_keyframe(kf, kf.img, None, kf.depth_img)
        t_enqueue = time.perf_counter() - t1
        # every keyframe consumed = q_in empty and every ring slot released (the worker releases a batch's slots when
        # integrate_frames has returned, i.e. when the DMA has read them; the last sweep may still be running)
        t_consumed = None
        if integ.frame_ring is not None:
            while time.perf_counter() - t1 < 300:
                if integ.q_in.qsize() == 0 and integ.frame_ring.held() == 0:
                    t_consumed = time.perf_counter() - t1
                    break
                time.sleep(0.0005)
        integ.add_update_output_task()
        last = None
        # (add_task pushes INTEGRATE tasks to the FRONT of the queue, as the reference does - base.py:1216-1232 - so the worker
        # sees the newest keyframe first; the UPDATE_OUTPUT task sits at the back and is reached when every keyframe is fused)
        while time.perf_counter() - t1 < 300:
            o = integ.pop_output(timeout=0.5)
            if o is not None and o.task_type == VolumetricIntegrationTaskType.UPDATE_OUTPUT and o.mesh is not None:
                last = o
                break
        dt = time.perf_counter() - t1
        if last is None:
            raise RuntimeError("no output after the last keyframe")
        return {"value": round((n_frames - 1) / dt, 1), "unit": "frames/s", "frames": n_frames - 1,
                "enqueue_s": round(t_enqueue, 4), "consumed_s": None if t_consumed is None else round(t_consumed, 4),
                "frames_per_s_until_consumed": None if t_consumed is None else round((n_frames - 1) / t_consumed, 1),
                "total_s": round(dt, 4), "mesh_vertices": int(len(last.mesh.vertices)),
                "mesh_bytes": int(last.mesh.vertices.nbytes + last.mesh.triangles.nbytes + last.mesh.vertex_colors.nbytes),
                "transport": "shared-memory ring (one memcpy per keyframe, page-locked, DMA'd in place), outputs through a shared segment, "
                             "control queue with put_front / get_batch" if shared else "images and outputs pickled through the queues (the reference's transport)",
                "what": "add_keyframe x N -> q_in -> worker process: integrate_frames (64 keyframes per call at most) -> extract_triangle_mesh -> "
                        "q_out -> pop_output, output interval 1 s, FPS throttle off; clock from the first add_keyframe to the mesh that holds every keyframe"}
    finally:
        integ.quit()


if __name__ == "__main__":
    import bench

    s, d, c, T = bench.load_frames("synthetic_640x480_5mm", 192)
    out = host_leg(s, d, c, T, bench.VOXEL, bench.SDF_TRUNC, bench.DEPTH_TRUNC, front="--no-front" not in sys.argv)
    out["env"] = {k: v for k, v in os.environ.items() if k.startswith("HV_STAGE")}
    out["cpus"] = os.cpu_count()
    print(json.dumps(out))
