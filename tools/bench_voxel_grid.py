#!/usr/bin/env python3
"""Secondary benchmark (VOXEL_GRID row): frames/s of the cpp/volumetric-semantics path
(VoxelBlockGrid.integrate_rgbd = fused depth2pointcloud + world transform + integrate, then get_voxels)
on one MI355X, next to the *compiled reference* (oracle/_ref, kind "reference", 1 thread: its non-TBB
build) and the C restatement, on the same frames.  Prints one JSON line.  Not the headline metric."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--cpu-frames", type=int, default=6)
    args = ap.parse_args()
    import torch

    from bench import load_frames
    from pyslam_amd.volumetric import VoxelBlockGrid

    s, depth, rgb, T = load_frames("synthetic_640x480_5mm", args.frames)
    depth_d, rgb_d = torch.from_numpy(depth).cuda(), torch.from_numpy(rgb).cuda()
    g = VoxelBlockGrid(0.005, 8, max_blocks=1 << 18, max_points=1 << 20)

    def step():
        for f in range(args.frames):
            g.integrate_rgbd(depth_d[f], rgb_d[f], *s.intrinsics, T[f], max_depth=4.0)

    step()
    g.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    g.synchronize()
    fps = args.steps * args.frames / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    v = g.get_voxels(3, 0.6)
    t_get = time.perf_counter() - t0
    # batched replay (rebuild / offline reconstruction): one device sort per batch instead of one per frame
    gb = VoxelBlockGrid(0.005, 8, max_blocks=1 << 18, max_points=args.frames * s.width * s.height)
    gb.integrate_rgbd_batch(depth_d, rgb_d, *s.intrinsics, T, max_depth=4.0)
    gb.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gb.integrate_rgbd_batch(depth_d, rgb_d, *s.intrinsics, T, max_depth=4.0)
    gb.synchronize()
    fps_batch = args.steps * args.frames / (time.perf_counter() - t0)

    import oracle
    from oracle import host_prep as hp

    out = {"metric": "RGB-D frames/sec fused (640x480, 5 mm, VOXEL_GRID cpp/volumetric semantics)", "value": round(fps, 1),
           "unit": "frames/s", "n_gpus": 1, "get_voxels_ms": round(t_get * 1e3, 2), "voxels_out": int(len(v.points)),
           "blocks": int(g.num_blocks()),
           "batched_replay": {"value": round(fps_batch, 1), "unit": "frames/s", "frames_per_sort": args.frames}}
    pts = [hp.frame_to_world_f32(depth[i], rgb[i], *s.intrinsics, T[i], 4.0)[:2] for i in range(args.cpu_frames)]
    for kind, cls in (("reference", oracle.RefGrid if oracle.ref_available() else None), ("port", oracle.PortGrid)):
        if cls is None:
            continue
        c = cls(0.005, 8)
        c.integrate(*pts[0])
        t0 = time.perf_counter()
        for p, col in pts[1:]:
            c.integrate(p, col)
        dt = time.perf_counter() - t0
        out[f"cpu_{kind}"] = {"value": round((len(pts) - 1) / dt, 3), "unit": "frames/s", "cores": 1, "kind": kind,
                              "sample": f"{len(pts) - 1} frames, integrate_raw<float,float> on the same float32 world points"
                                        + (" (unmodified cpp/volumetric sources, sequential non-TBB branch)" if kind == "reference" else "")}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
