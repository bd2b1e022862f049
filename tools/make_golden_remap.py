#!/usr/bin/env python3
"""PIN for the undistort / rectify rows: run wherever `import cv2` works and commit the output.

Writes tests/golden/remap_cv2.npz: cv2.getOptimalNewCameraMatrix + cv2.initUndistortRectifyMap + cv2.remap (linear for
colour, nearest for depth / labels) exactly as pyslam/dense/volumetric_integrator_base.py:758-786,1017-1043 calls them, on a
seeded distorted frame.  tests/test_prep_undistort.py consumes the file when it exists (pyslam_amd/prep.py's maps and
hv_remap against OpenCV's).  OpenCV is neither vendored by the reference nor installed in this image: until this script has
run somewhere, those rows stay "unpinned" in DESIGN.md."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    try:
        import cv2
    except ImportError:
        print("cv2 is not importable here: nothing written (P2 / N1 stay unpinned)")
        return 1
    rng = np.random.default_rng(3)
    W, H = 160, 120
    K = np.array([[131.25, 0, 79.5], [0, 131.25, 59.5], [0, 0, 1.0]])
    D = np.array([-0.28, 0.07, 0.0002, -0.0001, 0.0])
    img = rng.integers(0, 255, (H, W, 3)).astype(np.uint8)
    depth = (1.0 + rng.random((H, W))).astype(np.float32)
    labels = rng.integers(0, 40, (H, W)).astype(np.int32)
    new_K, roi = cv2.getOptimalNewCameraMatrix(K, D, (W, H), 0, (W, H))
    map_x, map_y = cv2.initUndistortRectifyMap(K, D, None, new_K, (W, H), cv2.CV_32FC1)
    out = {"K": K, "D": D, "size": np.array([W, H]), "new_K": new_K, "roi": np.array(roi), "map_x": map_x, "map_y": map_y, "img": img,
           "depth": depth, "labels": labels, "img_linear": cv2.remap(img, map_x, map_y, cv2.INTER_LINEAR),
           "depth_nearest": cv2.remap(depth, map_x, map_y, cv2.INTER_NEAREST),
           "labels_nearest": cv2.remap(labels.astype(np.float32), map_x, map_y, cv2.INTER_NEAREST).astype(np.int32),
           "cv2_version": cv2.__version__}
    path = os.path.join(ROOT, "tests", "golden", "remap_cv2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)
    return 0


if __name__ == "__main__":
    sys.exit(main())
