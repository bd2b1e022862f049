#!/usr/bin/env python3
"""PIN for the TSDF column: run wherever `import open3d` works (the reference pins open3d git
02674268f706be4b004bbbf3d39b95fa9de35f74 / conda 0.19.0, scripts/install_open3d_python.sh:118) and commit the output.

Writes tests/golden/tsdf_open3d.npz: a short seeded stream fused by the REAL
o3d.pipelines.integration.ScalableTSDFVolume exactly as pyslam/dense/volumetric_integrator_tsdf.py drives it
(:104-108 ctor, :215-223 RGBDImage + integrate, :239/:260 extract_triangle_mesh, :246/:267 extract_point_cloud), plus the
inputs.  tests/test_tsdf_open3d_pin.py consumes the file when it exists: the C restatement (oracle/tsdf_oracle.c) must
reproduce Open3D's mesh and point cloud - and with it everything the GPU path is compared against.  Open3D does not expose
its voxels to Python, so the pin is on what it does expose: vertices, vertex colours, triangles, points (and, through
extract_voxel_point_cloud, the voxel centres with weight > 0 and their tsdf as grey levels).

Neither this image nor the reference tree contains open3d: until this script has run somewhere, DESIGN.md keeps saying
"TSDF parity unpinned"."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    try:
        import open3d as o3d
    except ImportError:
        print("open3d is not importable here: nothing written (the TSDF column stays unpinned)")
        return 1
    from pyslam_amd.synthetic import SyntheticRGBD

    s = SyntheticRGBD("tiny_160x120_2cm")
    voxel, trunc, depth_scale, depth_trunc = 0.02, 0.08, 1.0, 4.0
    vol = o3d.pipelines.integration.ScalableTSDFVolume(voxel_length=voxel, sdf_trunc=trunc,
                                                       color_type=o3d.pipelines.integration.TSDFVolumeColorType.RGB8)
    K = o3d.camera.PinholeCameraIntrinsic(width=s.width, height=s.height, fx=s.fx, fy=s.fy, cx=s.cx, cy=s.cy)
    frames = [s[i] for i in (0, 3, 6, 40)]
    for depth, rgb, T in frames:
        rgbd = o3d.geometry.RGBDImage.create_from_color_and_depth(o3d.geometry.Image(rgb), o3d.geometry.Image(depth), depth_scale=depth_scale,
                                                                  depth_trunc=depth_trunc, convert_rgb_to_intensity=False)
        vol.integrate(rgbd, K, T)
    mesh = vol.extract_triangle_mesh()
    pc = vol.extract_point_cloud()
    vpc = vol.extract_voxel_point_cloud()
    out = {"open3d_version": o3d.__version__, "voxel": voxel, "trunc": trunc, "depth_scale": depth_scale, "depth_trunc": depth_trunc,
           "frame_ids": np.array([0, 3, 6, 40]), "config": "tiny_160x120_2cm",
           "vertices": np.asarray(mesh.vertices), "vertex_colors": np.asarray(mesh.vertex_colors), "triangles": np.asarray(mesh.triangles),
           "points": np.asarray(pc.points), "point_colors": np.asarray(pc.colors),
           "voxel_points": np.asarray(vpc.points), "voxel_tsdf_grey": np.asarray(vpc.colors)}
    path = os.path.join(ROOT, "tests", "golden", "tsdf_open3d.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)
    return 0


if __name__ == "__main__":
    sys.exit(main())
