"""Deterministic synthetic RGB-D stream: no dataset is available offline.

Scene: an axis-aligned room 6 x 4 x 3 m seen from inside, two spheres (r = 0.5 m) and one box;
depth by analytic ray casting through a pinhole camera; colour is a procedural function of the
world hit point quantised to uint8; the camera moves on a closed circle (radius 1.2 m, height
1.5 m) looking at the room centre.  T_cw (world -> camera) is float64, depth float32 metres.
Sensor model: optional Gaussian depth noise sigma = 1 mm * z^2 (default_rng(seed)), `invalid_frac`
random pixels set to 0 (default_rng(seed + 1)); `distorted=True` renders through the config's lens model (plumb-bob
coefficients, TUM1's: the ray of sensor pixel (u, v) goes through its UNDISTORTED normalised position), so that the
frames need the reference's undistort / rectify step before they are fused.
"""
import numpy as np

ROOM_MIN = np.array([0.0, 0.0, 0.0])
ROOM_MAX = np.array([6.0, 4.0, 3.0])
SPHERES = [(np.array([1.9, 1.3, 0.5]), 0.5), (np.array([4.3, 2.7, 0.5]), 0.5)]
BOX = (np.array([2.6, 1.7, 0.0]), np.array([3.4, 2.3, 0.8]))
CENTRE = np.array([3.0, 2.0, 1.5])

CONFIGS = {
    # BASELINE.json configs[1]: synthetic 640x480 @ 30 Hz, 5 mm TSDF
    "synthetic_640x480_5mm": dict(width=640, height=480, fx=525.0, fy=525.0, cx=319.5, cy=239.5, voxel=0.005),
    # TUM fr1-shaped (settings/TUM1.yaml:27-39 intrinsics, DepthMapFactor 5000 -> u16 depth)
    "tum1_640x480_5mm": dict(width=640, height=480, fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989, voxel=0.005,
                             dist=(0.262383, -0.953104, -0.005358, 0.002628, 1.163314)),  # Camera.k1 k2 p1 p2 k3, TUM1.yaml:32-36
    # Replica-shaped (settings/REPLICA.yaml:27-39)
    "replica_1200x680_4mm": dict(width=1200, height=680, fx=600.0, fy=600.0, cx=599.5, cy=339.5, voxel=0.004),
    # EuRoC-shaped (settings/EuRoC_stereo.yaml:18-35)
    "euroc_752x480_10mm": dict(width=752, height=480, fx=435.2047, fy=435.2047, cx=367.4517, cy=252.2008, voxel=0.010),
    # ScanNet colour resolution (BASELINE.json configs[4])
    "scannet_1296x968_2mm": dict(width=1296, height=968, fx=1165.72, fy=1165.74, cx=649.09, cy=484.77, voxel=0.002),
    # tiny case for CPU-sized parity tests
    "tiny_160x120_2cm": dict(width=160, height=120, fx=131.25, fy=131.25, cx=79.5, cy=59.5, voxel=0.02),
}


def look_at_pose(eye, target, up=np.array([0.0, 0.0, 1.0])):
    """T_cw for a camera at `eye` looking at `target` (camera x right, y down, z forward)."""
    f = target - eye
    f = f / np.linalg.norm(f)
    r = np.cross(f, up)
    r = r / np.linalg.norm(r)
    d = np.cross(f, r)
    R_wc = np.stack([r, d, f], axis=1)  # columns = camera axes in world
    T_wc = np.eye(4)
    T_wc[:3, :3] = R_wc
    T_wc[:3, 3] = eye
    T_cw = np.eye(4)
    T_cw[:3, :3] = R_wc.T
    T_cw[:3, 3] = -R_wc.T @ eye
    return T_cw, T_wc


def trajectory_pose(i, n_poses=600, radius=1.2, height=1.5):
    th = 2.0 * np.pi * (i % n_poses) / n_poses
    eye = np.array([CENTRE[0] + radius * np.cos(th), CENTRE[1] + radius * np.sin(th), height])
    # look through the room centre region at the far walls: depths span ~0.7-4.6 m, so the
    # reference's depth_trunc = 4.0 m (config_parameters.py:350) really truncates some pixels
    target = eye + np.array([np.cos(th + np.pi + 0.35), np.sin(th + np.pi + 0.35), -0.15])
    return look_at_pose(eye, target)


def _colour(p, label):
    """Procedural colour of world points p [N,3] -> uint8 [N,3]."""
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    check = ((np.floor(x * 2.0) + np.floor(y * 2.0) + np.floor(z * 2.0)) % 2.0) * 40.0
    r = 110.0 + 70.0 * np.sin(2.1 * x + 0.3 * label) + check
    g = 120.0 + 70.0 * np.sin(1.7 * y + 1.1 * label) + check
    b = 130.0 + 70.0 * np.sin(2.9 * z + 2.3 * label) + check
    return np.clip(np.stack([r, g, b], axis=1), 0, 255).astype(np.uint8)


def render(T_wc, width, height, fx, fy, cx, cy, dist=None):
    """Analytic ray cast.  Returns depth [H,W] f64 (z-depth, 0 = miss), rgb [H,W,3] u8, label [H,W] i32.  dist: lens distortion
    coefficients (k1 k2 p1 p2 k3 ...) of the sensor - pixel (u, v) then looks along its undistorted normalised direction."""
    u, v = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    if dist is not None:
        from pyslam_amd.prep import undistort_points_normalized

        K = np.array([[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]])
        xn, yn = undistort_points_normalized(u, v, K, dist, iters=20)
        d_cam = np.stack([xn, yn, np.ones_like(u)], axis=-1).reshape(-1, 3)
    else:
        d_cam = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], axis=-1).reshape(-1, 3)
    R, o = T_wc[:3, :3], T_wc[:3, 3]
    d = d_cam @ R.T  # world direction per unit z-depth
    n = d.shape[0]
    best = np.full(n, np.inf)
    label = np.zeros(n, np.int32)
    # room: exit face of the enclosing box
    with np.errstate(divide="ignore", invalid="ignore"):
        t_hi = np.where(d > 0, (ROOM_MAX - o) / d, np.where(d < 0, (ROOM_MIN - o) / d, np.inf))
    t_room = t_hi.min(axis=1)
    face = t_hi.argmin(axis=1)
    best = t_room
    label = 1 + face.astype(np.int32) * 2 + (d[np.arange(n), face] > 0)
    # spheres
    for si, (c, rad) in enumerate(SPHERES):
        oc = o - c
        a = (d * d).sum(1)
        b = 2.0 * (d @ oc)
        cc = oc @ oc - rad * rad
        disc = b * b - 4 * a * cc
        ok = disc > 0
        t = np.where(ok, (-b - np.sqrt(np.where(ok, disc, 0.0))) / (2 * a), np.inf)
        hit = ok & (t > 1e-6) & (t < best)
        best = np.where(hit, t, best)
        label = np.where(hit, 10 + si, label)
    # box (slab test, entry face)
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (BOX[0] - o) / d
        t2 = (BOX[1] - o) / d
    tn = np.nanmax(np.minimum(t1, t2), axis=1)
    tf = np.nanmin(np.maximum(t1, t2), axis=1)
    hit = (tn < tf) & (tn > 1e-6) & (tn < best)
    best = np.where(hit, tn, best)
    label = np.where(hit, 20, label)
    depth = np.where(np.isfinite(best), best, 0.0)
    p = o + d * depth[:, None]
    rgb = _colour(p, label.astype(np.float64))
    return depth.reshape(height, width), rgb.reshape(height, width, 3), label.reshape(height, width)


class SyntheticRGBD:
    """Indexable stream: ``depth, rgb, T_cw = stream[i]``."""

    def __init__(self, config="synthetic_640x480_5mm", noise=True, invalid_frac=0.02, seed=0, n_poses=600,
                 depth_dtype="float32", depth_map_factor=5000.0, distorted=False):
        c = CONFIGS[config] if isinstance(config, str) else dict(config)
        self.dist = tuple(c["dist"]) if distorted and c.get("dist") else None  # lens model of the SENSOR frames (None: pinhole)
        self.width, self.height = c["width"], c["height"]
        self.fx, self.fy, self.cx, self.cy = c["fx"], c["fy"], c["cx"], c["cy"]
        self.voxel = c["voxel"]
        self.noise, self.invalid_frac, self.seed, self.n_poses = noise, invalid_frac, seed, n_poses
        self.depth_dtype = depth_dtype
        self.depth_map_factor = depth_map_factor

    @property
    def intrinsics(self):
        return self.fx, self.fy, self.cx, self.cy

    def pose(self, i):
        return trajectory_pose(i, self.n_poses)[0]

    def __getitem__(self, i):
        T_cw, T_wc = trajectory_pose(i, self.n_poses)
        depth, rgb, _ = render(T_wc, self.width, self.height, self.fx, self.fy, self.cx, self.cy, self.dist)
        if self.noise:
            rng = np.random.default_rng(self.seed + 7919 * i)
            depth = depth + rng.standard_normal(depth.shape) * 1e-3 * depth * depth
        if self.invalid_frac > 0:
            rng = np.random.default_rng(self.seed + 1 + 7919 * i)
            depth = np.where(rng.random(depth.shape) < self.invalid_frac, 0.0, depth)
        depth = np.maximum(depth, 0.0)
        if self.depth_dtype == "uint16":
            depth = np.clip(np.rint(depth * self.depth_map_factor), 0, 65535).astype(np.uint16)
        else:
            depth = depth.astype(np.float32)
        return depth, np.ascontiguousarray(rgb), T_cw

    def labels(self, i):
        return render(trajectory_pose(i, self.n_poses)[1], self.width, self.height, self.fx, self.fy, self.cx, self.cy, self.dist)[2]

    def _args(self):
        return dict(config=dict(width=self.width, height=self.height, fx=self.fx, fy=self.fy, cx=self.cx, cy=self.cy,
                                voxel=self.voxel, dist=self.dist), distorted=self.dist is not None, noise=self.noise, invalid_frac=self.invalid_frac, seed=self.seed,
                    n_poses=self.n_poses, depth_dtype=self.depth_dtype, depth_map_factor=self.depth_map_factor)

    def frames(self, start, count, workers=None):
        """[(depth, rgb, T_cw)] for frames start .. start+count-1.  Rendering is host-side numpy (~0.25 s per 640x480
        frame on one core): longer runs are spread over worker subprocesses (`python -m pyslam_amd.synthetic`, fresh
        interpreters rather than forks: the caller may hold a HIP context).  Frames are seeded per index, so the result
        does not depend on the worker count."""
        import os

        if workers is None:
            workers = min(os.cpu_count() or 1, 32, count // 4)
        if workers <= 1:
            return [self[i] for i in range(start, start + count)]
        import json
        import subprocess
        import sys
        import tempfile

        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        per = (count + workers - 1) // workers
        with tempfile.TemporaryDirectory(prefix="pyslam_amd_synth_") as tmp:
            jobs = []
            for w in range(workers):
                lo = start + w * per
                n = min(per, start + count - lo)
                if n <= 0:
                    break
                out = os.path.join(tmp, f"part{w}.npz")
                cmd = [sys.executable, "-m", "pyslam_amd.synthetic", json.dumps(self._args()), str(lo), str(n), out]
                jobs.append((subprocess.Popen(cmd, cwd=root, env=dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")), out))
            frames = []
            for proc, out in jobs:
                if proc.wait() != 0:
                    raise RuntimeError("synthetic frame worker failed")
                z = np.load(out)
                frames.extend((z["depth"][k], z["rgb"][k], z["T"][k]) for k in range(len(z["T"])))
        return frames

    def batch(self, start, count, workers=None):
        fr = self.frames(start, count, workers)
        return np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr]), np.stack([f[2] for f in fr])


if __name__ == "__main__":  # worker of SyntheticRGBD.frames: <json ctor args> <start> <count> <out.npz>
    import json
    import sys

    _s = SyntheticRGBD(**json.loads(sys.argv[1]))
    _d, _c, _T = _s.batch(int(sys.argv[2]), int(sys.argv[3]), workers=1)
    np.savez(sys.argv[4], depth=_d, rgb=_c, T=_T)
