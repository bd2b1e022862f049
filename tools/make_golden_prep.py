#!/usr/bin/env python3
"""Generate tests/golden/prep_depth.npz by importing the REFERENCE's own pyslam/utilities/depth.py
(/root/reference, dev container only: the file needs nothing but numpy) and running depth2pointcloud and
filter_shadow_points on small seeded inputs.  The fixture pins oracle/host_prep.py (the restatement that travels to the GPU
box) and, through it, the GPU prep kernels (hv_integrate_rgbd_points, hv_filter_shadow_points): rows P3 / P4 / N2 of
SURVEY 8.  tests/test_golden.py also compares the restatement with the imported reference directly when /root/reference is
present."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/pyslam/utilities/depth.py"


def load_reference_depth():
    spec = importlib.util.spec_from_file_location("ref_depth", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def inputs():
    sys.path.insert(0, ROOT)
    from pyslam_amd.synthetic import SyntheticRGBD

    s = SyntheticRGBD("tiny_160x120_2cm")
    depth, rgb, _ = s[7]
    depth = depth.copy()
    depth[10:14, 20:30] = 0.0            # a hole
    depth[40:60, 80:83] += 0.6           # a depth step: shadow points on its rim
    return s, depth, rgb


def main():
    ref = load_reference_depth()
    s, depth, rgb = inputs()
    pc = ref.depth2pointcloud(depth, rgb, s.fx, s.fy, s.cx, s.cy, max_depth=3.5, min_depth=0.2)
    out = {"depth": depth, "rgb": rgb, "intr": np.array([s.fx, s.fy, s.cx, s.cy]), "max_depth": 3.5, "min_depth": 0.2,
           "points": np.asarray(pc.points), "colors": np.asarray(pc.colors),
           "shadow_mad": ref.filter_shadow_points(depth, delta_depth=None),
           "shadow_fixed": ref.filter_shadow_points(depth, delta_depth=0.05, delta_x=3, delta_y=1, fill_value=0.0)}
    path = os.path.join(ROOT, "tests", "golden", "prep_depth.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
