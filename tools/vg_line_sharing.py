#!/usr/bin/env python3
"""VOXEL_GRID per-frame fold: how many 128-byte lines do the cells a frame touches live in?  ("measure the
layout you argued against".)  CPU-only, from the oracle's keys of the bench's own frames: the voxel record is 32 bytes (4 per line),
a block's 512 records are laid out x + 8 y + 64 z; the alternative cell order is Morton (z-order) inside the block.  The fold reads
and writes every touched cell's record once per frame, so 2 x 128 B x (distinct lines) is the floor of its HBM traffic under either
order, and 2 x 32 B x (distinct cells) what a layout that packed a frame's cells densely could reach.
usage: python tools/vg_line_sharing.py [frames]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def morton3(x, y, z):
    def spread(v):
        v = v.astype(np.int64)
        v = (v | (v << 4)) & 0x0C3
        v = (v | (v << 2)) & 0x249
        return v

    return spread(x) | (spread(y) << 1) | (spread(z) << 2)


def main():
    import bench
    import oracle
    from oracle import host_prep as hp

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    s, depth_h, rgb_h, T_h = bench.load_frames("synthetic_640x480_5mm", n)
    rows = []
    for f in range(n):
        pts, _, _ = hp.frame_to_world_f32(depth_h[f], rgb_h[f], *s.intrinsics, T_h[f], bench.DEPTH_TRUNC)
        _, bk, lk, _ = oracle.keys(pts, bench.VOXEL, 8, which="port")
        cells = np.unique(np.concatenate([bk, lk], axis=1), axis=0)  # distinct (block, local cell)
        l = cells[:, 3:]
        _, bid = np.unique(cells[:, :3], axis=0, return_inverse=True)
        bid = bid.astype(np.int64).reshape(-1)
        lin = (l[:, 0] + 8 * l[:, 1] + 64 * l[:, 2]).astype(np.int64)
        mor = morton3(l[:, 0], l[:, 1], l[:, 2])
        rows.append((len(pts), len(cells), int(bid.max()) + 1, len(np.unique(bid * 512 + lin // 4)), len(np.unique(bid * 512 + mor // 4))))
    r = np.array(rows, float).mean(0)
    out = {"frames": n, "points_per_frame": round(r[0]), "cells_per_frame": round(r[1]), "blocks_per_frame": round(r[2]),
           "lines_128B_xyz_order": round(r[3]), "lines_128B_morton_order": round(r[4]),
           "cells_per_line_xyz": round(r[1] / r[3], 3), "cells_per_line_morton": round(r[1] / r[4], 3),
           "fold_floor_MB_per_frame_xyz": round(2 * 128 * r[3] / 1e6, 1), "fold_floor_MB_per_frame_morton": round(2 * 128 * r[4] / 1e6, 1),
           "dense_cell_floor_MB_per_frame": round(2 * 32 * r[1] / 1e6, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
