#!/usr/bin/env python3
"""NS1, closed by measurement: the 3x4 . 4xN point transform of the VOXEL_GRID unprojection on the matrix
cores (k_vg_unproject_mfma: four v_mfma_f64_16x16x4_f64 per wave) against the VALU form (k_vg_unproject), on the same frames:

  * voxels: both forms fuse the frames through the radix path (HV_VG_PATH=sort: the path that launches the unprojection as a
    kernel of its own) into a 5 mm block grid; the occupied voxels and their counts are compared with the grid the NUMPY-ORDER points
    produce (oracle/host_prep.py: depth2pointcloud + R @ P.T as pyslam/dense/volumetric_integrator_voxel_grid.py:262-265 and
    pyslam/utilities/depth.py:64-76 compute them - `blas=True`) and with the explicit-order points the parity contract names;
  * time: run under `rocprofv3 --kernel-trace --stats` (tools/_final.sh) - the two kernels' rows are the figure; this script also
    prints the wall time of the whole integrate call per form.

Reference call sites: pyslam/dense/volumetric_integrator_voxel_grid.py:262-265, pyslam/utilities/depth.py:64-76.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def occupied(grid):
    keys, _, counts, _ = grid.dump()
    b, l = np.nonzero(counts > 0)
    bs = grid.block_size
    vox = keys[b].astype(np.int64) * bs + np.stack([l % bs, (l // bs) % bs, l // (bs * bs)], axis=1)
    return {tuple(v): int(c) for v, c in zip(vox.tolist(), counts[b, l].tolist())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch

    import bench
    import oracle
    from oracle import host_prep as hp
    from pyslam_amd.volumetric import VoxelBlockGrid

    os.environ["HV_VG_PATH"] = "sort"
    s, depth_h, rgb_h, T_h = bench.load_frames("synthetic_640x480_5mm", args.frames)
    depth_d, rgb_d = torch.from_numpy(depth_h).cuda(), torch.from_numpy(rgb_h).cuda()
    out = {"frames": args.frames, "config": "synthetic_640x480_5mm, 5 mm voxels, 8^3 blocks"}
    grids = {}
    for form in ("valu", "mfma"):
        os.environ["HV_VG_UNPROJECT"] = form
        g = VoxelBlockGrid(bench.VOXEL, 8, max_blocks=1 << 17, max_points=1 << 20)
        for f in range(args.frames):
            g.integrate_rgbd(depth_d[f], rgb_d[f], *s.intrinsics, T_h[f], max_depth=bench.DEPTH_TRUNC)
        grids[form] = occupied(g)
        t = VoxelBlockGrid(bench.VOXEL, 8, max_blocks=1 << 17, max_points=1 << 20)
        t.integrate_rgbd(depth_d[0], rgb_d[0], *s.intrinsics, T_h[0], max_depth=bench.DEPTH_TRUNC)
        t.synchronize()
        t0 = time.perf_counter()
        for r in range(args.reps):
            t.integrate_rgbd(depth_d[r % args.frames], rgb_d[r % args.frames], *s.intrinsics, T_h[r % args.frames], max_depth=bench.DEPTH_TRUNC)
        t.synchronize()
        out[f"{form}_integrate_call_us"] = round((time.perf_counter() - t0) / args.reps * 1e6, 1)
    for name, blas in (("numpy_order", True), ("explicit_order", False)):
        ref = {}
        for f in range(args.frames):
            pts, _, _ = hp.frame_to_world_f32(depth_h[f], rgb_h[f], *s.intrinsics, T_h[f], bench.DEPTH_TRUNC, blas=blas)
            vk = oracle.keys(pts, bench.VOXEL, 8, which="port")[0]
            uk, cnt = np.unique(vk, axis=0, return_counts=True)
            for k, c in zip(uk.tolist(), cnt.tolist()):
                ref[tuple(k)] = ref.get(tuple(k), 0) + c
        n_pts = sum(ref.values())
        for form in ("valu", "mfma"):
            got = grids[form]
            keys = set(ref) | set(got)
            moved = sum(abs(ref.get(k, 0) - got.get(k, 0)) for k in keys) // 2  # points that landed in another voxel
            out[f"{form}_vs_{name}"] = {"points": n_pts, "voxels_ref": len(ref), "voxels_only_in_one": len(set(ref) ^ set(got)),
                                        "points_in_another_voxel": int(moved), "fraction": round(moved / max(n_pts, 1), 9)}
    out["what"] = ("points_in_another_voxel: half the L1 distance between the per-voxel point counts = points whose voxel key differs from the "
                   "reference order's.  explicit_order = ((r0 x + r1 y) + r2 z) + t with every product and sum rounded (the parity contract, "
                   "DESIGN section 2); numpy_order = R @ P.T through BLAS as the reference calls it")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
