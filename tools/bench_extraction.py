"""Extraction on its own: kernel and wall time of extract_triangle_mesh / extract_point_cloud on the volume ten 32-frame
batches of the synthetic 640x480 / 5 mm stream build (~24 k units), three ticks (one keyframe fused between ticks, so every
tick recomputes).  The first tick also page-locks the result arrays.  Do NOT run under rocprofv3 on a box without cached
frames: the frame generator's worker processes inherit the profiler."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from pyslam_amd.volumetric import ScalableTSDFVolume, PinholeCameraIntrinsic, RGBDImage

s, depth, rgb, T = bench.load_frames("synthetic_640x480_5mm", 320)
K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
dd, rr = torch.from_numpy(depth).cuda(), torch.from_numpy(rgb).cuda()
vol = ScalableTSDFVolume(bench.VOXEL, bench.SDF_TRUNC, max_blocks=1 << 17)
for k in range(10):
    vol.integrate_batch(dd[32 * k:32 * k + 32], rr[32 * k:32 * k + 32], K, T[32 * k:32 * k + 32], depth_scale=1.0, depth_trunc=bench.DEPTH_TRUNC)
vol.synchronize()
print("units", vol.num_blocks())
for rep in range(3):
    vol.integrate(RGBDImage(rr[0], dd[0], 1.0, bench.DEPTH_TRUNC), K, T[0])
    vol.synchronize()
    vol.profile_enable(True)
    t0 = time.perf_counter(); m = vol.extract_triangle_mesh(); t1 = time.perf_counter()
    km = vol.profile_read()[0]
    pc = vol.extract_point_cloud(); t2 = time.perf_counter()
    kp = vol.profile_read()[0]
    print("rep", rep, "mesh kernels ms %.3f wall %.2f | points kernels ms %.3f wall %.2f |" % (km, (t1 - t0) * 1e3, kp, (t2 - t1) * 1e3), len(m.vertices), len(m.triangles), len(pc.points))
    vol.profile_enable(False)
