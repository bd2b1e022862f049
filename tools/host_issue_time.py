"""Is a small rank's step bound by the HOST issuing it?  Times tools/sweep_variants.py's loop two ways for one rank's share of the
ownership sharding: how long the host takes to ISSUE the calls of K steps (return of the last integrate_batch, no synchronisation)
and how long until the device has finished them.  usage: python tools/host_issue_time.py [--owner 3/8] [--steps 40]"""
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
    ap.add_argument("--owner", default="3/8")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--config", default="synthetic_640x480_5mm")
    args = ap.parse_args()
    import torch

    import bench
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, ScalableTSDFVolume

    B = 32
    n_distinct = min(args.steps * B, SyntheticRGBD(args.config).n_poses)
    s, depth_h, rgb_h, T_h = bench.load_frames(args.config, n_distinct)
    wrap = np.arange(n_distinct + B) % n_distinct
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    depth_d = torch.from_numpy(depth_h[wrap]).cuda()
    rgb_d = torch.from_numpy(rgb_h[wrap]).cuda()
    T_res = T_h[wrap]
    out = []
    for owner in (args.owner, ""):
        vol = ScalableTSDFVolume(bench.VOXEL, bench.SDF_TRUNC, max_blocks=1 << 17, max_points=s.width * s.height)
        if owner:
            r, w = (int(x) for x in owner.split("/"))
            vol.set_owner(r, w)
        for rep in range(3):
            vol.reset()
            for k in range(8):
                lo = (k * B) % n_distinct
                vol.integrate_batch(depth_d[lo:lo + B], rgb_d[lo:lo + B], K, T_res[lo:lo + B], depth_scale=1.0, depth_trunc=bench.DEPTH_TRUNC)
            vol.synchronize()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(args.steps):
                lo = (k * B) % n_distinct
                vol.integrate_batch(depth_d[lo:lo + B], rgb_d[lo:lo + B], K, T_res[lo:lo + B], depth_scale=1.0, depth_trunc=bench.DEPTH_TRUNC)
            t1 = time.perf_counter()
            vol.synchronize()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        out.append({"owner": owner or "1/1", "steps": args.steps, "host_issue_us_per_step": round((t1 - t0) / args.steps * 1e6, 1),
                    "device_done_us_per_step": round((t2 - t0) / args.steps * 1e6, 1)})
        del vol
    print(json.dumps(out))


if __name__ == "__main__":
    main()
