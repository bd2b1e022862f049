#!/usr/bin/env python3
"""Projects the strong-scaling curve of bench.py's unit-ownership sharding on ONE GPU: for world = 1, 2, 4, 8 it
runs every rank's share of the headline step (same frames, owner(unit) == rank) back to back and reports the
slowest rank — what the N-GPU job's step time would be with no communication (there is none while fusing).
Output: one JSON line per world size (gpurun_out/ when run on the GPU box)."""
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
    ap.add_argument("--frames-per-step", type=int, default=64)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--mode", default="batch")
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--window", default="replay", choices=["replay", "sliding"],
                    help="replay: the same frames every step (round 3's curve); sliding: step k fuses frames 32k ... 32k + 31 of the loop "
                         "(the headline's window: the camera moves)")
    ap.add_argument("--sharding", default="owner", choices=["owner", "tile"],
                    help="owner: owner(unit) == rank; tile: vertical image tiles (a unit several tiles see is fused by each of them, "
                         "partial means - what merge_halo() reconciles; not timed here)")
    ap.add_argument("--only-rank", type=int, default=-1, help="run just this rank's share (for a kernel trace)")
    args = ap.parse_args()
    import torch

    from bench import DEPTH_TRUNC, SDF_TRUNC, VOXEL, load_frames
    from pyslam_amd.distributed import ShardedTSDF
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage

    B = args.frames_per_step
    n_frames = B if args.window == "replay" else B * args.steps
    s, depth_h, rgb_h, T_h = load_frames("synthetic_640x480_5mm", n_frames)
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    depth_d, rgb_d = torch.from_numpy(depth_h).cuda(), torch.from_numpy(rgb_h).cuda()
    base = None
    for world in [int(w) for w in args.worlds.split(",")]:
        per_rank = []
        for rank in range(world):
            if args.only_rank >= 0 and rank != args.only_rank:
                continue
            f = ShardedTSDF(VOXEL, SDF_TRUNC, s.width, s.height, device=0, max_blocks=1 << 15, rank=rank, world_size=world, sharding=args.sharding)
            vol = f.volume

            def step(k=0):
                lo = 0 if args.window == "replay" else k * B
                if args.mode == "batch":
                    vol.integrate_batch(depth_d[lo:lo + B], rgb_d[lo:lo + B], K, T_h[lo:lo + B], depth_scale=1.0, depth_trunc=DEPTH_TRUNC)
                else:
                    for i in range(lo, lo + B):
                        vol.integrate(RGBDImage(rgb_d[i], depth_d[i], 1.0, DEPTH_TRUNC), K, T_h[i])

            if args.window == "replay":
                step(); step()
            else:  # warm pass over the whole window, then the volume starts empty like the headline's
                for k in range(args.steps):
                    step(k)
                vol.reset()
            vol.synchronize()
            t0 = time.perf_counter()
            for k in range(args.steps):
                step(k)
            vol.synchronize()
            per_rank.append((time.perf_counter() - t0) / args.steps * 1e3)
            units = vol.num_blocks()
            del f, vol
        ms = max(per_rank)
        base = base or ms
        print(json.dumps({"world": world, "sharding": args.sharding, "mode": args.mode, "window": args.window, "ms_per_step_slowest_rank": round(ms, 4),
                          "ms_per_rank": [round(x, 4) for x in per_rank], "frames_per_s": round(B / ms * 1e3, 1),
                          "speedup_vs_1": round(base / ms, 3), "units_last_rank": int(units)}), flush=True)


if __name__ == "__main__":
    main()
