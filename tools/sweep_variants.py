"""A/B of the multi-frame sweep's switches inside ONE process (a gpurun call is charged for the box, not only for the run:
one import of torch, one synthetic stream, many variants).  Every variant times bench.py's headline step - sliding window,
B = 64 posed 640x480 frames (--frames-per-step) resident in HBM, 5 mm - on a volume that is empty when the clock starts, and reports frames/s and
the sweep kernel's mean launch duration (HIP events on its stream).

usage: python tools/sweep_variants.py [--steps 12] [--warmup 3] 'HV_TSDF_SWEEP=4' 'HV_TSDF_SWEEP=2' ...
Prints one JSON line per variant (also appended to gpurun_out/sweep_variants.jsonl)."""
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
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-step", type=int, default=64)
    ap.add_argument("--config", default="synthetic_640x480_5mm")
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--owner", default="", help="rank/world: time one rank's share of the unit-ownership sharding (hv_tsdf_set_owner)")
    ap.add_argument("variants", nargs="*", default=["HV_TSDF_SWEEP=4"])
    args = ap.parse_args()
    import torch

    import bench
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, ScalableTSDFVolume

    B = args.frames_per_step
    n_distinct = min(args.steps * B, SyntheticRGBD(args.config).n_poses)
    s, depth_h, rgb_h, T_h = bench.load_frames(args.config, n_distinct)
    wrap = np.arange(n_distinct + B) % n_distinct
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    depth_d = torch.from_numpy(depth_h[wrap]).cuda()
    rgb_d = torch.from_numpy(rgb_h[wrap]).cuda()
    T_res = T_h[wrap]
    vol = ScalableTSDFVolume(bench.VOXEL, bench.SDF_TRUNC, max_blocks=1 << 17, max_points=s.width * s.height)
    if args.owner:
        r, w = (int(x) for x in args.owner.split("/"))
        vol.set_owner(r, w)
    results = {}
    out = os.path.join(ROOT, "gpurun_out", "sweep_variants.jsonl")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    for _ in range(args.repeat):
        for variant in args.variants:
            env = dict(kv.split("=", 1) for kv in variant.split())
            for k, v in env.items():
                os.environ[k] = v
            try:
                vol.reset()
                for k in range(args.warmup):
                    lo = (k * B) % n_distinct
                    vol.integrate_batch(depth_d[lo:lo + B], rgb_d[lo:lo + B], K, T_res[lo:lo + B], depth_scale=1.0, depth_trunc=bench.DEPTH_TRUNC)
                vol.synchronize()
                vol.reset()
                vol.synchronize()
                torch.cuda.synchronize()
                vol.profile_enable(True)
                t0 = time.perf_counter()
                for k in range(args.steps):
                    lo = (k * B) % n_distinct
                    vol.integrate_batch(depth_d[lo:lo + B], rgb_d[lo:lo + B], K, T_res[lo:lo + B], depth_scale=1.0, depth_trunc=bench.DEPTH_TRUNC)
                vol.synchronize()
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                launches = vol.profile_launches()
                vol.profile_read()
                vol.profile_enable(False)
                line = {"variant": variant, "frames_per_s": round(args.steps * B / el, 1), "ms_per_step": round(el / args.steps * 1e3, 4),
                        "sweep_us_mean": round(float(np.mean(launches)) * 1e3, 1) if len(launches) else None,
                        "units": int(vol.num_blocks())}
            finally:
                for k in env:
                    os.environ.pop(k, None)
            results.setdefault(variant, []).append(line)
            with open(out, "a") as f:
                f.write(json.dumps(line) + "\n")
    # per variant: median over the repeats (the first pass of a process runs on cold clocks: +5 .. 10 %)
    for variant, lines in results.items():
        summary = {"variant": variant, "n": len(lines),
                   "frames_per_s_median": float(np.median([x["frames_per_s"] for x in lines])),
                   "sweep_us_median": float(np.median([x["sweep_us_mean"] for x in lines if x["sweep_us_mean"] is not None] or [0])),
                   "sweep_us_all": [x["sweep_us_mean"] for x in lines]}
        print(json.dumps(summary), flush=True)


if __name__ == "__main__":
    main()
