#!/usr/bin/env python3
"""Secondary benchmark (BASELINE config 5 shape): keyframes/s of pySLAM's per-keyframe semantic flow
(shadow filter -> assign_object_ids_to_instance_ids -> remap_instance_ids -> depth2pointcloud with labels + world
transform -> integrate, volumetric_integrator_voxel_semantic_grid.py:322-461) on one MI355X for both semantic payloads,
next to the *compiled reference* (oracle/_ref: unmodified cpp/volumetric sources, sequential non-TBB branch, 1 core)
running the same flow with numpy host prep.  Also times get_voxels and get_object_segments.  One JSON line."""
import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


_FLOW_KERNELS = ("k_shadow_", "k_sem_assoc_", "k_remap_instance_ids", "k_semb_", "k_sem_keys", "k_sem_reduce", "k_sem_carve")


def flow_traffic(pmc_file, payload_tag):
    """HBM bytes of one keyframe's flow from the recorded --pmc passes of this tool on THIS build (tools/_final.sh; two separate
    passes, FETCH_SIZE then WRITE_SIZE, MI355X_MICROARCH.md's HBM section): sum over the flow's kernels of (2 x FETCH_SIZE +
    WRITE_SIZE) x 1024 x launches per keyframe.  A kernel instantiated per payload counts for its payload; the others ran for both
    payloads of the recorded command, half of their launches each.  -> (bytes, per-kernel dict) or (None, None)."""
    from bench import PROFILE_ROUND, current_build_digest, hbm_traffic

    try:
        with open(os.path.join(ROOT, "profiles", PROFILE_ROUND, pmc_file)) as f:
            z = json.load(f)
    except (OSError, ValueError):
        return None, None
    if z.get("build_digest") != current_build_digest():
        return None, None
    frames = None
    for tok in z.get("command_key", "").split("--frames")[1:]:
        frames = int(tok.split()[0])
    if not frames:
        return None, None
    per = {}
    for name, pk in z.get("kernels", {}).items():
        if not any(k in name for k in _FLOW_KERNELS):
            continue
        t = hbm_traffic(pk)
        if t is None:
            continue
        if "HvSemVoxel" in name or "HvProbVoxel" in name:
            if payload_tag not in name:
                continue
            share = pk["launches"] / frames
        else:
            share = pk["launches"] / (2.0 * frames)
        short = name.replace("void ", "").split("<")[0]
        per[short] = per.get(short, 0) + int(t * share)
    return (int(sum(per.values())), per) if per else (None, None)


def semantic_leg(n_frames=24, cpu_frames=4, voxel=0.01, config="synthetic_640x480_5mm", stride=3, star2=False):
    """-> dict (bench.py's `semantic` key / this tool's JSON line).  config="scannet_1296x968_2mm", voxel=0.002 is BASELINE
    configs[4]'s shape (1.25 M points per keyframe, 1.6 cm blocks)."""
    import types

    args = types.SimpleNamespace(frames=n_frames, cpu_frames=cpu_frames, voxel=voxel)
    import oracle
    from oracle import host_prep as hp
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import CameraFrustrum
    from pyslam_amd.volumetric_semantic import (VoxelBlockSemanticGrid, VoxelBlockSemanticGrid2, VoxelBlockSemanticProbabilisticGrid,
                                                VoxelBlockSemanticProbabilisticGrid2, remap_instance_ids, set_next_object_id)
    from pyslam_amd.dense.device_pipeline import KeyframeUploader
    from tests.semantic_helpers import frame_points, semantic_frame

    s = SyntheticRGBD(config)
    intr = s.intrinsics
    frames = [semantic_frame(s, stride * i, shuffle=i) for i in range(args.frames)]
    try:  # the keyframes as the worker sees them: in page-locked host memory (the front's registered shared-memory ring)
        import torch

        def pinned(a):
            t = torch.empty(a.shape, dtype=torch.from_numpy(a[:0] if a.ndim else a).dtype, pin_memory=True)
            t.numpy()[...] = a
            return t.numpy()

        frames = [tuple(pinned(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) and a.ndim >= 2 and a.shape[0] > 4 else a for a in f) for f in frames]
        out_pinned = True
    except Exception:
        out_pinned = False
    out = {"metric": f"keyframes/sec, semantic flow ({s.width}x{s.height}, assign+remap+integrate, cpp/volumetric semantics)", "unit": "keyframes/s",
           "n_gpus": 1, "voxel": args.voxel, "frames": args.frames, "config": config}
    # algorithmic bytes per keyframe (SURVEY 8d shape): every distinct voxel a keyframe touches is read and written once (64 B voting /
    # 128 B probabilistic record) + the keyframe's inputs (depth f32 + rgb u8x3 + class i32 + instance i32 = 15 B / pixel)
    v_touched = []
    for depth, rgb, T, cls_img, inst_img in frames[2:2 + max(1, min(3, len(frames) - 2))]:
        p = hp.frame_to_world_f32(hp.filter_shadow_points(depth), rgb, *intr, T, 4.0)[0]
        v_touched.append(len(np.unique(oracle.keys(p, args.voxel, 8, which="port")[0], axis=0)))
    b_in = s.width * s.height * 15
    host_flow = os.environ.get("PYSLAM_AMD_SEMANTIC_DEVICE_FLOW", "1") == "0"  # A/B: every call stages its own host inputs (round 2)
    out["flow"] = "host images staged by every call" if host_flow else ("one upload per image, steps on device tensors, the next keyframe's shadow filter beside this keyframe's kernels "
                                                                           "(the integrator's flow)")
    out["host_images"] = "page-locked (as in the front's registered ring), asynchronous uploads" if out_pinned else "pageable"
    payloads = [("voting", VoxelBlockSemanticGrid, 0), ("probabilistic", VoxelBlockSemanticProbabilisticGrid, 1)]
    if star2:  # the module's other two payloads (voxel_data_semantic2.h), on request: same flow, same checks
        payloads += [("voting2", VoxelBlockSemanticGrid2, 2), ("probabilistic2", VoxelBlockSemanticProbabilisticGrid2, 3)]
    for name, cls, kind in payloads:
        mb_log2 = int(os.environ.get("PYSLAM_AMD_SEMANTIC_MAX_BLOCKS_LOG2", "19" if args.voxel < 0.004 else "17"))  # A/B: the hash table has 4 slots per block of this
        g = cls(args.voxel, 8, max_blocks=1 << mb_log2, max_points=max(1 << 20, s.width * s.height))  # (a pool that does not have to grow inside the timed keyframes)
        fr = CameraFrustrum(*intr, s.width, s.height, np.eye(4), depth_max=8.0, depth_min=0.01)
        set_next_object_id(1)

        prep_side = not host_flow and os.environ.get("PYSLAM_AMD_SEMANTIC_PREP", "1") != "0"  # A/B: 0 = the filter in front of every keyframe's own kernels (rounds 4-5)

        one_call = not host_flow and os.environ.get("PYSLAM_AMD_SEMANTIC_ONE_CALL", "1") != "0"  # A/B: 0 = the staged calls (rounds 4-5)

        def fuse(frame, t):
            depth, rgb, T, cls_img, inst_img = frame
            if one_call:  # what _fuse_device_keyframe does: the whole body in one call into the library
                g.fuse_keyframe(fr, t["depth"], t["color"], t["cls"], t["inst"], *intr, T, filter_shadow_points=True, use_instance_ids=True,
                                depth_threshold=0.03, do_carving=False, min_vote_ratio=0.5, min_votes=3, max_depth=4.0, use_depths=True,
                                depth_is_filtered=prep_side)
                return
            d = t["depth"] if prep_side else g.filter_shadow_points(t["depth"] if t is not None else depth)
            c, cl, ins = (t["color"], t["cls"], t["inst"]) if t is not None else (rgb, cls_img, inst_img)
            fr.set_T_cw(T)
            m = g.assign_object_ids_to_instance_ids(fr, cl, ins, d, depth_threshold=0.03, do_carving=False, min_vote_ratio=0.5, min_votes=3)
            obj = remap_instance_ids(ins, m, volume=g)
            g.integrate_rgbd(d, c, *intr, T, class_ids_image=cl, object_ids_image=obj, max_depth=4.0, use_depths=True)

        uploader = None if host_flow else KeyframeUploader(g)

        def run(frames_):
            # what pyslam_amd/dense/volumetric_integrator_voxel_semantic_grid.py::integrate_keyframes_on_device does with a backlog of
            # keyframes: host images in, one upload each on the copy stream (keyframe k + 1 beside the kernels of keyframe k), every
            # step on the device copies, no host round trip inside a keyframe
            if host_flow:
                for f in frames_:
                    fuse(f, None)
                return
            def prep(t, stream):  # the shadow filter on the upload side (integrate_keyframes_on_device does the same)
                t["depth"] = g.filter_shadow_points(t["depth"], stream=stream)

            uploader.run(frames_, lambda f: {"depth": (f[0], np.float32), "color": (f[1], np.uint8), "cls": (f[3], np.int32),
                                             "inst": (f[4], np.int32)}, fuse, prep if prep_side else None)

        run(frames[:2])
        g.synchronize()
        t0 = time.perf_counter()
        run(frames[2:])
        g.synchronize()
        fps = (len(frames) - 2) / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        v = g.get_voxels(3, 0.6)
        t_get = time.perf_counter() - t0
        t0 = time.perf_counter()
        segs = g.get_object_segments(3, 0.6)
        t_seg = time.perf_counter() - t0
        rec = 64 if kind in (0, 2) else 128
        alg = 2 * rec * float(np.mean(v_touched)) + b_in
        # ... and what the association reads: every voxel in the last keyframe's view (count >= 1) is looked at once by
        # assign_object_ids_to_instance_ids - in the reference as well (iterate_voxels_in_camera_frustrum); one record each
        fr.set_T_cw(frames[-1][2])
        n_view = int(len(g.get_voxels_in_camera_frustrum(fr, 1, 0.0).points))
        # ... and at EVERY timed keyframe (an untimed second pass over the same stream on the cleared grid: the voxels in view of
        # keyframe k BEFORE it is fused - what its association looks at; the view fills up as the map grows)
        views = []
        if not host_flow and args.cpu_frames > 0:  # (like the parity check below: not in the profiled commands, --cpu-frames 0, whose launches are counted per keyframe)
            g.clear()
            set_next_object_id(1)
            for k, frame in enumerate(frames):
                if k >= 2:
                    fr.set_T_cw(frame[2])
                    views.append(int(len(g.get_voxels_in_camera_frustrum(fr, 1, 0.0).points)))
                run([frame])
                g.synchronize()
        n_view_mean = float(np.mean(views)) if views else float(n_view)
        res = {"value": round(fps, 1),
               "roofline": {"bound": "hbm", "what": "the WHOLE per-keyframe flow (shadow filter, association, remap, integrate), not one kernel: "
                            f"algorithmic bytes = 2 x {rec} B x distinct voxels touched (oracle keys) + 15 B / pixel of inputs",
                            "algorithmic_bytes_per_keyframe": int(alg), "achieved": round(alg * fps / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                            "frac": round(alg * fps / 1e9 / 8000.0, 5), "traffic": None,
                            "association": {"voxels_in_view_last_keyframe": n_view, "bytes": int(rec * n_view),
                                            "algorithmic_bytes_with_association": int(alg + rec * n_view),
                                            "voxels_in_view_mean_over_timed_keyframes": int(n_view_mean),
                                            "algorithmic_bytes_with_association_mean": int(alg + rec * n_view_mean),
                                            "frac_with_association": round((alg + rec * n_view_mean) * fps / 1e9 / 8000.0, 5),
                                            "what": "assign_object_ids_to_instance_ids looks at every voxel in the keyframe's view once (the reference "
                                                    "does too): one record each, counted at the LAST keyframe of the stream (the view fills up as the "
                                                    "map grows) and - *_mean - before every timed keyframe in an untimed second pass; not part of "
                                                    "algorithmic_bytes_per_keyframe; frac_with_association = (integrate bytes + mean association bytes) x "
                                                    "keyframes/s / 8 TB/s"}},
               "get_voxels_ms": round(t_get * 1e3, 2), "voxels_out": int(len(v.points)),
               "get_object_segments_ms": round(t_seg * 1e3, 2), "objects": len(segs.object_vector), "blocks": int(g.num_blocks()),
               "label_overflows": g.label_overflows()}
        # (the recorded counter passes ran the first two payloads only)
        traffic, per = flow_traffic("pmc_semantic_scannet_2mm.json" if args.voxel < 0.004 else "pmc_semantic.json", "HvSemVoxel" if kind == 0 else "HvProbVoxel") if kind < 2 else (None, None)
        if traffic is not None:
            res["roofline"]["traffic"] = traffic
            res["roofline"]["traffic_over_algorithmic"] = round(traffic / alg, 2)
            res["roofline"]["traffic_over_algorithmic_with_association"] = round(traffic / (alg + rec * n_view), 2)
            res["roofline"]["traffic_over_algorithmic_with_association_mean"] = round(traffic / (alg + rec * n_view_mean), 2)
            res["roofline"]["traffic_per_kernel"] = per
            res["roofline"]["traffic_source"] = "recorded rocprofv3 --pmc passes of tools/bench_semantic.py on this build (profiles/, build digest checked)"
        if oracle.ref_available() and args.cpu_frames > 0:
            # The compiled reference runs the same flow on the first cpu_frames + 1 keyframes of the stream - and is COMPARED with a fresh
            # HIP grid fed the same keyframes (tests/semantic_helpers.py::compare_keyframe_flow: filtered depth, id maps, id images
            # after every keyframe; block count, every occupied voxel, count distribution, segments at the end).  A mismatch raises:
            # a semantic number is never printed for a configuration where the flow is not the reference's.
            from tests.semantic_helpers import compare_keyframe_flow

            par = compare_keyframe_flow(kind, config, args.voxel, [stride * i for i in range(args.cpu_frames + 1)], max_blocks=1 << 17,
                                        max_points=max(1 << 20, s.width * s.height), depth_threshold=0.03, frustum_depth=(8.0, 0.01))
            dt = sum(par["ref_seconds"][1:])
            res["cpu_reference"] = {"value": round(args.cpu_frames / dt, 3), "unit": "keyframes/s", "cores": 1, "kind": "reference",
                                    "sample": f"{args.cpu_frames} keyframes, same flow: numpy shadow filter / depth2pointcloud + compiled "
                                              f"cpp/volumetric (sequential non-TBB branch)"}
            res["speedup_vs_cpu_reference"] = round(fps / res["cpu_reference"]["value"], 1)
            res["parity"] = {"checked": True, "against": "compiled reference (oracle/_ref), same keyframes", "keyframes": par["keyframes"],
                             "blocks": par["blocks"], "occupied_voxels": par["occupied_voxels"], "objects": par["objects"],
                             "conf_max_abs_diff": par["conf_max_abs_diff"], "label_overflows": par["label_overflows"]}
        out[name] = res
        del g
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--cpu-frames", type=int, default=4)
    ap.add_argument("--voxel", type=float, default=0.01)
    ap.add_argument("--config", default="synthetic_640x480_5mm")
    ap.add_argument("--stride", type=int, default=3)
    ap.add_argument("--star2", action="store_true", help="also time VoxelBlockSemanticGrid2 / VoxelBlockSemanticProbabilisticGrid2 (voxel_data_semantic2.h)")
    args = ap.parse_args()
    print(json.dumps(semantic_leg(args.frames, args.cpu_frames, args.voxel, args.config, args.stride, args.star2)))


if __name__ == "__main__":
    main()
