#!/usr/bin/env python3
"""Host-side profile (cProfile) of the semantic keyframe flow tools/bench_semantic.py times: where a keyframe's wall time goes
between the ctypes calls, torch uploads and the library's host synchronisations.  Output: the top entries by cumulative time."""
import cProfile
import io
import os
import pstats
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import CameraFrustrum
    from pyslam_amd.volumetric_semantic import VoxelBlockSemanticGrid, remap_instance_ids, set_next_object_id
    from tests.semantic_helpers import semantic_frame

    s = SyntheticRGBD("synthetic_640x480_5mm")
    frames = [semantic_frame(s, 3 * i, shuffle=i) for i in range(24)]
    g = VoxelBlockSemanticGrid(0.01, 8, max_blocks=1 << 15, max_points=1 << 19)
    fr = CameraFrustrum(*s.intrinsics, s.width, s.height, np.eye(4), depth_max=8.0, depth_min=0.01)
    set_next_object_id(1)

    def run(fs):
        for depth, rgb, T, cls_img, inst_img in fs:
            d = g.filter_shadow_points(torch.from_numpy(depth).cuda())
            c, cl, ins = torch.from_numpy(rgb).cuda(), torch.from_numpy(cls_img).cuda(), torch.from_numpy(inst_img).cuda()
            fr.set_T_cw(T)
            m = g.assign_object_ids_to_instance_ids(fr, cl, ins, d, depth_threshold=0.03, do_carving=False, min_vote_ratio=0.5, min_votes=3)
            obj = remap_instance_ids(ins, m, volume=g)
            g.integrate_rgbd(d, c, *s.intrinsics, T, class_ids_image=cl, object_ids_image=obj, max_depth=4.0, use_depths=True)
        g.synchronize()

    run(frames[:4])
    pr = cProfile.Profile()
    pr.enable()
    run(frames[4:])
    pr.disable()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(28)
    print(out.getvalue())


if __name__ == "__main__":
    main()
