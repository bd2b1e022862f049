"""TEST INFRASTRUCTURE ONLY — numpy restatement of pySLAM's per-frame host prep (L3).

A restatement because /root/reference does not exist on the GPU box; it is PINNED to the reference's own
pyslam/utilities/depth.py, which needs nothing but numpy: tools/make_golden_prep.py imports that file and writes
tests/golden/prep_depth.npz, tests/test_golden.py checks these functions against the fixture bit for bit (and against
the imported reference directly where /root/reference is present).  geometry.inv_T is restated with an explicit
operation order (the reference's goes through numba / BLAS).  Used by tests/ to check the fused GPU unprojection
(hv_integrate_rgbd_points), the shadow-point filter, and by bench.py's voxel-grid cpu_baseline leg.
"""
import numpy as np


def depth2pointcloud(depth, image, fx, fy, cx, cy, max_depth=np.inf, min_depth=0.0):
    """pyslam/utilities/depth.py:45-85 (points/colours only).  float64 results, row-major order."""
    valid = (depth > min_depth) & (depth < max_depth)
    z = depth[valid]
    inv_fx = 1.0 / fx
    inv_fy = 1.0 / fy
    rows, cols = np.where(valid)
    x = (cols - cx) * z * inv_fx
    y = (rows - cy) * z * inv_fy
    points = np.column_stack([x, y, z])
    colors = image[valid] / 255.0
    return points, colors, valid


def inv_T(T):
    """pyslam/utilities/geometry.py:98-104 with the 3x3 products spelled out left to right
    (numba/BLAS order is unspecified; this is the order the HIP library uses)."""
    ret = np.eye(4)
    R_T = T[:3, :3].T
    t = T[:3, 3]
    ret[:3, :3] = R_T
    ret[:3, 3] = -((R_T[:, 0] * t[0] + R_T[:, 1] * t[1]) + R_T[:, 2] * t[2])
    return ret


def world_points(points_cam, T_cw, blas=False):
    """volumetric_integrator_voxel_grid.py:262-265.  blas=False: explicit ((R0*x + R1*y) + R2*z) + t
    order (what the GPU computes); blas=True: the reference's literal `R @ P.T + t` (library order)."""
    Twc = inv_T(np.asarray(T_cw, dtype=np.float64))
    R, t = Twc[:3, :3], Twc[:3, 3]
    if blas:
        return (R @ points_cam.T + t.reshape(3, 1)).T
    x, y, z = points_cam[:, 0], points_cam[:, 1], points_cam[:, 2]
    out = np.empty_like(points_cam)
    for r in range(3):
        out[:, r] = ((R[r, 0] * x + R[r, 1] * y) + R[r, 2] * z) + t[r]
    return out


def frame_to_world_f32(depth, rgb, fx, fy, cx, cy, T_cw, max_depth, min_depth=0.0, blas=False):
    """Full L3 chain of VolumetricIntegratorVoxelGrid (…voxel_grid.py:251-281): float32 world points
    and float32 colours in [0,1], ready for VoxelBlockGrid.integrate()."""
    pts, cols, valid = depth2pointcloud(depth, rgb, fx, fy, cx, cy, max_depth, min_depth)
    pw = world_points(pts, T_cw, blas=blas)
    return np.ascontiguousarray(pw, dtype=np.float32), np.ascontiguousarray(cols, dtype=np.float32), valid


def filter_shadow_points(depth, delta_depth=None, delta_x=2, delta_y=2, fill_value=-1):
    """pyslam/utilities/depth.py:103-146."""
    depth_out = depth.copy()
    mask = np.zeros_like(depth, dtype=bool)
    delta_values = []
    if delta_y > 0:
        delta_depth_y = np.abs(depth[delta_y:, :] - depth[:-delta_y, :])
        if delta_depth is None:
            delta_values.append(delta_depth_y.flatten())
    if delta_x > 0:
        delta_depth_x = np.abs(depth[:, delta_x:] - depth[:, :-delta_x])
        if delta_depth is None:
            delta_values.append(delta_depth_x.flatten())
    if delta_depth is None:
        delta_values = np.concatenate(delta_values)
        delta_values = delta_values[delta_values > 0]
        mad = np.median(delta_values)
        sigma_depth = 1.4826 * mad
        delta_depth = 3 * sigma_depth
    if delta_y > 0:
        big = delta_depth_y > delta_depth
        mask[delta_y:, :] |= big
        mask[:-delta_y, :] |= big
    if delta_x > 0:
        big = delta_depth_x > delta_depth
        mask[:, delta_x:] |= big
        mask[:, :-delta_x] |= big
    depth_out[mask] = fill_value
    return depth_out


def _cv_round(a):
    """cvRound: round half to even (float32 maps, as cv2.remap converts them)."""
    return np.rint(np.asarray(a, dtype=np.float32)).astype(np.int64)


def remap_nearest(src, map_x, map_y):
    """cv2.remap(src, map_x, map_y, cv2.INTER_NEAREST), BORDER_CONSTANT 0 (volumetric_integrator_base.py:1030-1043), restated:
    dst(y, x) = src(cvRound(map_y), cvRound(map_x)).  OpenCV is not in this image: unpinned (tests/test_prep_undistort.py)."""
    H, W = src.shape[:2]
    sx, sy = _cv_round(map_x), _cv_round(map_y)
    ok = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H)
    out = np.zeros(map_x.shape + src.shape[2:], dtype=src.dtype)
    out[ok] = src[sy[ok], sx[ok]]
    return out


def remap_linear_u8(src, map_x, map_y):
    """cv2.remap(src uint8 HxWxC, ..., cv2.INTER_LINEAR), BORDER_CONSTANT 0 (volumetric_integrator_base.py:1019-1028), restated:
    fixed-point bilinear with INTER_BITS = 5 - coordinates quantised to 1/32 pixel, integer weights summing to 2^15,
    (sum + 2^14) >> 15.  Unpinned, like remap_nearest."""
    H, W = src.shape[:2]
    sx = _cv_round(np.asarray(map_x, np.float32) * np.float32(32.0))
    sy = _cv_round(np.asarray(map_y, np.float32) * np.float32(32.0))
    ix, iy, fx, fy = sx >> 5, sy >> 5, sx & 31, sy & 31
    w = [(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32]
    acc = np.zeros(map_x.shape + src.shape[2:], dtype=np.int64)
    for (dy, dx), wk in zip(((0, 0), (0, 1), (1, 0), (1, 1)), w):
        x, y = ix + dx, iy + dy
        ok = (x >= 0) & (x < W) & (y >= 0) & (y < H)
        px = np.zeros_like(acc)
        px[ok] = src[y[ok], x[ok]]
        acc += px * (wk[..., None] if src.ndim == 3 else wk)
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
