#!/usr/bin/env python3
"""Secondary benchmark (VOXEL_GRID row): bench.py's voxel_grid leg on its own - frames/s of the
cpp/volumetric-semantics path (VoxelBlockGrid.integrate_rgbd = fused depth2pointcloud + world transform + integrate) on one
MI355X, per frame and batched, next to the COMPILED REFERENCE (oracle/_ref, kind "reference", 1 thread: its non-TBB build)
on the same frames.  Prints one JSON line.  Run it under rocprofv3 --kernel-trace --stats for the kernel table."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--cpu-frames", type=int, default=6)
    args = ap.parse_args()
    import torch

    import bench

    s, depth, rgb, T = bench.load_frames("synthetic_640x480_5mm", args.frames)
    out = bench.voxel_grid_leg(s, depth, rgb, T, torch.from_numpy(depth).cuda(), torch.from_numpy(rgb).cuda(), args.frames, args.steps,
                               args.cpu_frames)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
