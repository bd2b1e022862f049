#!/usr/bin/env python3
"""merge_halo over RCCL with one rank on the one GPU (force_collectives: every dirty unit goes through lists -> gather -> plan -> pack ->
all-reduce -> unpack as its own keeper): wall time of the merge with the key lists / plan on the device (hv_halo.hip) and through the host
(rounds 3-5), first call and repeated calls.  usage: python tools/halo_timing.py [device|host]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

import bench
from pyslam_amd.distributed import ShardedTSDF
from pyslam_amd.volumetric import PinholeCameraIntrinsic

mode = sys.argv[1] if len(sys.argv) > 1 else "device"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1)
s, depth, rgb, T = bench.load_frames("synthetic_640x480_5mm", 128)
K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
dd, rr = torch.from_numpy(depth).cuda(), torch.from_numpy(rgb).cuda()
fuser = ShardedTSDF(bench.VOXEL, bench.SDF_TRUNC, s.width, s.height, device=0, max_blocks=1 << 17, rank=0, world_size=1, sharding="tile",
                    force_collectives=True)
if mode == "host":
    fuser._merge_halo_device = None
    import pyslam_amd.volumetric as V

    del V.ScalableTSDFVolume.halo_lists_device  # merge_halo takes the host path
out = []
for k in range(4):
    fuser.integrate_batch(dd[32 * k:32 * k + 32], rr[32 * k:32 * k + 32], K, T[32 * k:32 * k + 32], depth_scale=1.0, depth_trunc=bench.DEPTH_TRUNC)
    fuser.volume.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_shared, n_dirty = fuser.merge_halo()
    torch.cuda.synchronize()
    fuser.volume.synchronize()
    out.append({"ms": round((time.perf_counter() - t0) * 1e3, 3), "shared": int(n_shared), "dirty": int(n_dirty), "bytes": int(fuser.last_halo["payload_bytes"])})
print("HALO " + json.dumps({"mode": mode, "merges": out}), flush=True)
dist.destroy_process_group()
