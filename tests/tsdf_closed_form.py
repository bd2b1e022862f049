"""A second, differently structured TSDF evaluator - test infrastructure, numpy only.

PARITY UNPINNED: Open3D is not installed, so nothing here is a pin.  What this narrows is what a wrong READING of Open3D could
hide: `oracle/tsdf_oracle.c` and the HIP kernels restate ScalableTSDFVolume operation by operation (float32, the repeated `+=`
walk along z, the rounding of every intermediate); this module shares no helper and no structure with them.  It evaluates a whole
frame in CLOSED FORM, in float64 geometry, straight from what SURVEY.md 8a (T2-T4) says the algorithm computes (reference call
sites pyslam/dense/volumetric_integrator_tsdf.py:104-108,215-223):

    a frame updates the units  floor((p -/+ sdf_trunc) / unit_length)  of every stride-th valid depth sample p (back-projected);
    inside such a unit a voxel centre c = (index + 0.5) * voxel_length maps to  pc = T_cw c,  (u, v) = floor(f pc_xy / pc_z + c + 0.5);
    it is updated iff pc_z > 0, (u, v) lies in the image, 0 < depth(u, v) < depth_trunc and
        sdf = (depth(u, v) - pc_z) * |K^-1 (u, v, 1)|  >  -sdf_trunc;
    then  tsdf <- running mean of min(1, sdf / sdf_trunc),  colour <- running mean of rgb(u, v),  weight += 1.

Because the arithmetic here is float64 and direct while the implementations round to float32 along a different path, a voxel whose
decision (which pixel, inside or outside the truncation band, inside the image) sits within rounding distance of a boundary is
marked FRAGILE and left out of the value comparison - counted, and bounded by the tests.  Everything else must agree: the unit set
exactly, the weight of every non-fragile voxel exactly, tsdf to 6e-5 (the float32 z-walk of the implementations: <= ~1.5e-6 m
on the sdf, divided by sdf_trunc = 0.04 m), colour to 1e-6 of the 0..255 scale.

The scene is analytic (a tilted plane with a sphere in front of it, ray-cast per pixel at 640x480), so the depth images carry no
generator shared with pyslam_amd.synthetic either.
"""
import numpy as np

W, H = 640, 480
K = np.array([525.0, 525.0, 319.5, 239.5])
VOXEL, TRUNC, DEPTH_TRUNC, STRIDE, RES = 0.005, 0.04, 4.0, 4, 16
UNIT = VOXEL * RES
PLANE_N = np.array([0.15, -0.25, 1.0]) / np.linalg.norm([0.15, -0.25, 1.0])
PLANE_D = 1.6
SPHERE_C, SPHERE_R = np.array([0.1, -0.05, 1.1]), 0.3


def pose(eye, target, up=(0.05, -1.0, 0.1)):
    """-> T_cw (world -> camera, z forward), float64."""
    eye, target, up = np.asarray(eye, float), np.asarray(target, float), np.asarray(up, float)
    z = target - eye
    z /= np.linalg.norm(z)
    x = np.cross(z, up)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z])  # rows: camera axes in world coordinates
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = -R @ eye
    return T


POSES = (pose((0.25, -0.15, -0.35), (0.1, 0.0, 1.3)), pose((-0.35, 0.2, -0.2), (0.05, -0.05, 1.2)), pose((0.0, 0.35, -0.1), (0.1, -0.1, 1.25)))


def render(T_cw):
    """z-depth (float32 metres, 0 = invalid) and colour (uint8 RGB) of the analytic scene: nearest of plane and sphere per pixel."""
    fx, fy, cx, cy = K
    T_wc = np.linalg.inv(T_cw)
    v, u = np.mgrid[0:H, 0:W].astype(np.float64)
    d_cam = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], axis=-1)  # z = 1: the ray parameter IS the z-depth
    d_w = d_cam @ T_wc[:3, :3].T
    eye = T_wc[:3, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        t_plane = (PLANE_D - eye @ PLANE_N) / (d_w @ PLANE_N)
    t_plane = np.where(np.isfinite(t_plane) & (t_plane > 0.05), t_plane, np.inf)
    oc = eye - SPHERE_C
    a = (d_w * d_w).sum(-1)
    b = 2.0 * (d_w @ oc)
    c = oc @ oc - SPHERE_R ** 2
    disc = b * b - 4 * a * c
    with np.errstate(invalid="ignore"):
        t_sph = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.inf)
    t_sph = np.where(t_sph > 0.05, t_sph, np.inf)
    t = np.minimum(t_plane, t_sph)
    depth = np.where(np.isfinite(t), t, 0.0).astype(np.float32)
    depth[::37, ::41] = 0.0  # a sprinkle of invalid pixels
    rgb = np.stack([(3 * u + 2 * v) % 256, (5 * v + 11) % 256, np.where(t_sph < t_plane, 200, 40) + (u % 32)], axis=-1).astype(np.uint8)
    return depth, rgb


def frames():
    return [(*render(T), T) for T in POSES]


def touched_units(depth, T_cw):
    """-> sorted [U, 3] int64: the units a frame opens (vectorised over the strided samples; sdf_trunc < unit_length, so a
    sample's range has one or two units per axis)."""
    fx, fy, cx, cy = K
    T_wc = np.linalg.inv(T_cw)
    ii, jj = np.mgrid[0:H:STRIDE, 0:W:STRIDE]
    d = depth[ii, jj].astype(np.float64)
    ok = (d > 0) & (d < DEPTH_TRUNC)
    p_cam = np.stack([(jj - cx) * d / fx, (ii - cy) * d / fy, d], axis=-1)[ok]
    p = p_cam @ T_wc[:3, :3].T + T_wc[:3, 3]
    lo = np.floor((p - TRUNC) / UNIT).astype(np.int64)
    hi = np.floor((p + TRUNC) / UNIT).astype(np.int64)
    assert (hi - lo).max() <= 1
    out = []
    for corner in range(8):
        pick = np.array([(corner >> a) & 1 for a in range(3)], bool)
        out.append(np.where(pick, hi, lo))
    return np.unique(np.concatenate(out), axis=0)


def evaluate(frame_list, chunk=256):
    """-> keys [U,3] (sorted like the dumps), weight [U,4096] int, tsdf [U,4096] f64 (mean of the accepted frames' values),
    colour [U,4096,3] f64, fragile [U,4096] bool.  Voxel order inside a unit: x * 256 + y * 16 + z (the dumps')."""
    fx, fy, cx, cy = K
    per_frame = [touched_units(d, T) for d, _, T in frame_list]
    keys = np.unique(np.concatenate(per_frame), axis=0)
    order = np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))
    keys = keys[order]
    U = len(keys)
    weight = np.zeros((U, RES ** 3), np.int32)
    t_sum = np.zeros((U, RES ** 3), np.float64)
    c_sum = np.zeros((U, RES ** 3, 3), np.float64)
    fragile = np.zeros((U, RES ** 3), bool)
    lx, ly, lz = np.meshgrid(np.arange(RES), np.arange(RES), np.arange(RES), indexing="ij")  # flattened: x * 256 + y * 16 + z
    local = (np.stack([lx, ly, lz], axis=-1).reshape(-1, 3) + 0.5) * VOXEL
    key_id = {tuple(k): i for i, k in enumerate(keys.tolist())}
    for depth, rgb, T_cw in frame_list:
        rows = np.array(sorted(key_id[tuple(k)] for k in touched_units(depth, T_cw).tolist()))
        R, t = T_cw[:3, :3], T_cw[:3, 3]
        for lo in range(0, len(rows), chunk):
            sel = rows[lo:lo + chunk]
            c = keys[sel][:, None, :] * UNIT + local[None, :, :]            # [n, 4096, 3] voxel centres, float64
            pc = c @ R.T + t
            z = pc[..., 2]
            front = z > 0
            with np.errstate(divide="ignore", invalid="ignore"):
                uf = np.where(front, pc[..., 0] * fx / z + cx + 0.5, -1.0)
                vf = np.where(front, pc[..., 1] * fy / z + cy + 0.5, -1.0)
            inside = front & (uf >= 0.0001) & (uf < W - 0.0001) & (vf >= 0.0001) & (vf < H - 0.0001)
            ui = np.clip(np.floor(uf), 0, W - 1).astype(np.int64)
            vi = np.clip(np.floor(vf), 0, H - 1).astype(np.int64)
            dd = depth[vi, ui].astype(np.float64)
            valid = inside & (dd > 0) & (dd < DEPTH_TRUNC)
            mult = np.sqrt(((ui - cx) / fx) ** 2 + ((vi - cy) / fy) ** 2 + 1.0)
            sdf = (dd - z) * mult
            acc = valid & (sdf > -TRUNC)
            # decisions within rounding distance of a boundary (float32 walk vs float64 closed form: ~1e-6 relative)
            near_px = (np.abs(uf - np.round(uf)) < 2e-3) | (np.abs(vf - np.round(vf)) < 2e-3)
            near_border = front & ((np.abs(uf - 0.0001) < 2e-3) | (np.abs(uf - (W - 0.0001)) < 2e-3) | (np.abs(vf - 0.0001) < 2e-3) |
                                   (np.abs(vf - (H - 0.0001)) < 2e-3))
            near_band = valid & (np.abs(sdf + TRUNC) < 1e-4)
            near_plane = np.abs(z) < 1e-5
            # a neighbouring pixel with another validity / a depth step: the pixel choice then decides the value
            fragile[sel] |= (front & near_px & (inside | near_border)) | near_border | near_band | near_plane
            weight[sel] += acc
            t_sum[sel] += np.where(acc, np.minimum(1.0, sdf / TRUNC), 0.0)
            c_sum[sel] += np.where(acc[..., None], rgb[vi, ui].astype(np.float64), 0.0)
    wz = np.maximum(weight, 1)
    return keys, weight, t_sum / wz, c_sum / wz[..., None], fragile


def compare(dump, ref, what):
    """dump = (keys, tsdf, weight, colour) of an implementation, ref = evaluate(...).  -> dict of what was checked."""
    keys, tsdf, weight, colour = dump
    rkeys, rw, rt, rc, fragile = ref
    np.testing.assert_array_equal(np.asarray(keys, np.int64), rkeys, err_msg=f"{what}: unit set")
    ok = ~fragile
    w = np.asarray(weight)
    bad_w = ok & (w != rw)
    assert not bad_w.any(), (what, "weights differ on", int(bad_w.sum()), "non-fragile voxels; first:", np.argwhere(bad_w)[:3].tolist())
    upd = ok & (rw > 0)
    t_err = float(np.abs(np.asarray(tsdf, np.float64) - rt)[upd].max())
    c_err = float(np.abs(np.asarray(colour, np.float64) - rc)[upd].max())
    assert t_err <= 6e-5, (what, "tsdf", t_err)  # float32 z-walk (<= ~1.5e-6 m on the sdf at 1.5 m) / sdf_trunc 0.04 m; the contract is 1e-4
    assert c_err <= 1e-6 * 255.0 + 1e-9, (what, "colour", c_err)
    untouched = ok & (rw == 0)
    assert not np.asarray(tsdf)[untouched].any(), (what, "a voxel no frame accepts carries a value")
    # the fragile voxels are not a hiding place: their weights may differ by the frames that were on a boundary, never by more
    assert (np.abs(w.astype(np.int64) - rw)[fragile] <= 3).all()
    return {"units": int(len(rkeys)), "voxels": int(ok.size), "fragile_frac": float(fragile.mean()), "updated": int(upd.sum()),
            "max_weight": int(rw.max()), "tsdf_err": t_err, "colour_err": c_err}
