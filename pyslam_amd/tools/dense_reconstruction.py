#!/usr/bin/env python3
"""Headless counterpart of pySLAM's main_map_dense_reconstruction.py (:73-230): reload a saved system state
(``-p <folder with map.json>``) or read a dataset with ground-truth poses, feed every keyframe to the volumetric
integrator (add_keyframe + add_update_output_task, exactly the reference's loop), then save ``dense_map.ply``.

    python -m pyslam_amd.tools.dense_reconstruction -p results/slam_state -o results/slam_state_dense_reconstruction
    python -m pyslam_amd.tools.dense_reconstruction --dataset tum --root /data/tum --name rgbd_dataset_freiburg1_desk \\
           --settings settings/TUM1.yaml --type TSDF -o out/
"""
import argparse
import os
import time


def run(keyframes, camera, environment_type, sensor_type, integrator_type, output_path, voxel_length=None, log=print,
        drain_timeout=120.0):
    from pyslam_amd.dense import VolumetricIntegrationTaskType, VolumetricIntegratorType, volumetric_integrator_factory
    from pyslam_amd.dense.parameters import get_parameters
    from pyslam_amd.dense.volumetric_integrator_types import SensorType

    P = get_parameters()
    if voxel_length is not None:
        P.kVolumetricIntegrationVoxelLength = float(voxel_length)
    P.kVolumetricIntegrationUseDepthEstimator = sensor_type == SensorType.STEREO  # main_map_dense_reconstruction.py:118-120
    P.kVolumetricIntegrationMinNumLBATimes = 0  # :126
    integ = volumetric_integrator_factory(VolumetricIntegratorType.from_string(integrator_type), camera, environment_type, sensor_type)
    t0 = time.time()
    while not integ.is_ready():
        if time.time() - t0 > 120:
            raise RuntimeError("the volumetric integrator worker did not start")
        time.sleep(0.05)
    n = last_id = 0
    outputs = 0
    try:
        for kf in keyframes:
            integ.add_keyframe(kf, kf.img, kf.img_right, kf.depth_img)
            last_id = kf.id
            n += 1
            while integ.q_out.qsize() > 0:
                if integ.pop_output(timeout=0.05) is not None:
                    outputs += 1
        log(f"inserted #keyframes: {n}")
        # Completion (the reference's script is GUI-driven and has none): keyframes wait in the integrator's keyframe queue until
        # its timer moves them to q_in, INTEGRATE tasks are pushed to the FRONT of q_in (newest first), UPDATE_OUTPUT tasks are
        # appended - so once the keyframe queue is empty, the answer to an UPDATE_OUTPUT sent NOW comes after every INTEGRATE.
        t0 = time.time()
        while len(integ.keyframe_queue) > 0 and time.time() - t0 < drain_timeout:
            integ.flush_keyframe_queue()
            time.sleep(0.02)
        integ.add_update_output_task()
        done = False
        while not done and time.time() - t0 < drain_timeout:
            out = integ.pop_output(timeout=1.0)
            while out is not None:
                outputs += 1
                done = done or out.task_type == VolumetricIntegrationTaskType.UPDATE_OUTPUT
                out = integ.pop_output(timeout=0.05)
        os.makedirs(output_path, exist_ok=True)
        integ.save(output_path)
        log(f"saved {os.path.join(output_path, 'dense_map.ply')} ({outputs} outputs, "
            f"{integ.time_volumetric_integration.value * 1e3:.2f} ms last integration)")
    finally:
        integ.quit()
    return n


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-p", "--path", help="path where the system state (map.json) was saved")
    ap.add_argument("-o", "--output_path", default=None, help="where to save dense_map.ply")
    ap.add_argument("--dataset", choices=["tum", "icl_nuim", "replica", "scannet", "euroc"], help="read a dataset instead of a saved state")
    ap.add_argument("--root", help="dataset base path")
    ap.add_argument("--name", help="sequence name")
    ap.add_argument("--settings", help="pySLAM settings yaml with the Camera.* block (settings/TUM1.yaml ...)")
    ap.add_argument("--step", type=int, default=1)
    ap.add_argument("--max-frames", type=int, default=None)
    ap.add_argument("--type", default="TSDF", help="VOXEL_GRID | VOXEL_SEMANTIC_GRID | VOXEL_SEMANTIC_PROBABILISTIC_GRID | TSDF")
    ap.add_argument("--voxel-length", type=float, default=None)
    args = ap.parse_args(argv)

    if args.dataset:
        from pyslam_amd.io.datasets import camera_from_settings, dataset_factory

        camera = camera_from_settings(args.settings)
        ds = dataset_factory(args.dataset, args.root, args.name, camera)
        keyframes = ds.keyframes(step=args.step, max_frames=args.max_frames)
        env, sensor = ds.environment_type, ds.sensor_type
        out = args.output_path or os.path.join(args.root, args.name + "_dense_reconstruction")
    else:
        from pyslam_amd.io.system_state import load_system_state

        st = load_system_state(args.path)
        camera, env, sensor = st.camera, st.environment_type, st.sensor_type
        keyframes = st.map.get_keyframes()[:: args.step]
        out = args.output_path or (args.path.rstrip("/") + "_dense_reconstruction")
    return run(keyframes, camera, env, sensor, args.type, out, args.voxel_length)


if __name__ == "__main__":
    main()
