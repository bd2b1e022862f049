"""Undistort / rectify preparation: host-side restatement of the two OpenCV calls the
reference makes once per camera (pyslam/dense/volumetric_integrator_base.py:758-786):

    new_K, _ = cv2.getOptimalNewCameraMatrix(K, D, (w, h), alpha, (w, h))
    map1, map2 = cv2.initUndistortRectifyMap(K, D, None, new_K, (w, h), cv2.CV_32FC1)

The per-frame `cv2.remap` (bilinear for colour, nearest for depth / labels, :1017-1043) runs on the
GPU (`hv_remap`).  PARITY UNPINNED: OpenCV is neither vendored by the reference nor installed here, so
these follow OpenCV's published algorithms (plumb-bob model k1,k2,p1,p2,k3[,k4,k5,k6]; 9x9 grid of
undistorted points for the inner/outer rectangles; 5 fixed-point iterations of undistortPoints) and
are validated geometrically (tests/test_prep_undistort.py), not bit-for-bit against cv2.
"""
import numpy as np


def _dist_coeffs(D):
    d = np.zeros(8, dtype=np.float64)
    D = np.asarray(D, dtype=np.float64).ravel()
    d[: min(8, D.size)] = D[:8]
    return d  # k1 k2 p1 p2 k3 k4 k5 k6


def distort_normalized(x, y, D):
    """Normalized undistorted (x, y) -> normalized distorted (xd, yd) (plumb-bob / rational model)."""
    k1, k2, p1, p2, k3, k4, k5, k6 = _dist_coeffs(D)
    r2 = x * x + y * y
    r4 = r2 * r2
    r6 = r4 * r2
    kr = (1 + k1 * r2 + k2 * r4 + k3 * r6) / (1 + k4 * r2 + k5 * r4 + k6 * r6)
    xd = x * kr + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * kr + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return xd, yd


def undistort_points_normalized(u, v, K, D, iters=5):
    """cv::undistortPoints (no R, no P): pixel -> normalized undistorted, fixed-point iteration."""
    k1, k2, p1, p2, k3, k4, k5, k6 = _dist_coeffs(D)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    x0 = (np.asarray(u, dtype=np.float64) - cx) / fx
    y0 = (np.asarray(v, dtype=np.float64) - cy) / fy
    x, y = x0.copy(), y0.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        icdist = (1 + ((k6 * r2 + k5) * r2 + k4) * r2) / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x = (x0 - dx) * icdist
        y = (y0 - dy) * icdist
    return x, y


def get_optimal_new_camera_matrix(K, D, image_size, alpha, new_image_size=None):
    """cv::getOptimalNewCameraMatrix (centerPrincipalPoint=False): interpolate between the projection
    that maps the inscribed rectangle of the undistorted image to the viewport (alpha=0) and the one
    that maps the circumscribed rectangle (alpha=1).  Returns (new_K, None)."""
    K = np.asarray(K, dtype=np.float64)
    w, h = image_size
    nw, nh = new_image_size if new_image_size else image_size
    N = 9
    jj, ii = np.meshgrid(np.arange(N), np.arange(N))
    u = (jj * w / (N - 1)).astype(np.float32).astype(np.float64)
    v = (ii * h / (N - 1)).astype(np.float32).astype(np.float64)
    x, y = undistort_points_normalized(u, v, K, D)
    o_x0, o_x1, o_y0, o_y1 = x.min(), x.max(), y.min(), y.max()
    i_x0, i_x1 = x[:, 0].max(), x[:, N - 1].min()
    i_y0, i_y1 = y[0, :].max(), y[N - 1, :].min()
    inner = (i_x0, i_y0, i_x1 - i_x0, i_y1 - i_y0)
    outer = (o_x0, o_y0, o_x1 - o_x0, o_y1 - o_y0)
    fx0, fy0 = (nw - 1) / inner[2], (nh - 1) / inner[3]
    cx0, cy0 = -fx0 * inner[0], -fy0 * inner[1]
    fx1, fy1 = (nw - 1) / outer[2], (nh - 1) / outer[3]
    cx1, cy1 = -fx1 * outer[0], -fy1 * outer[1]
    M = np.eye(3)
    M[0, 0] = fx0 * (1 - alpha) + fx1 * alpha
    M[1, 1] = fy0 * (1 - alpha) + fy1 * alpha
    M[0, 2] = cx0 * (1 - alpha) + cx1 * alpha
    M[1, 2] = cy0 * (1 - alpha) + cy1 * alpha
    return M, None


def init_undistort_rectify_map(K, D, new_K, image_size):
    """cv::initUndistortRectifyMap(K, D, R=I, new_K, size, CV_32FC1): for every pixel of the
    *undistorted* image, the (x, y) position to sample in the distorted source image."""
    K = np.asarray(K, dtype=np.float64)
    new_K = np.asarray(new_K, dtype=np.float64)
    w, h = image_size
    uu, vv = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    x = (uu - new_K[0, 2]) / new_K[0, 0]
    y = (vv - new_K[1, 2]) / new_K[1, 1]
    xd, yd = distort_normalized(x, y, D)
    map_x = (K[0, 0] * xd + K[0, 2]).astype(np.float32)
    map_y = (K[1, 1] * yd + K[1, 2]).astype(np.float32)
    return np.ascontiguousarray(map_x), np.ascontiguousarray(map_y)


class Undistorter:
    """Holds the maps on the GPU side of a volume and applies the reference's per-frame remaps:
    colour INTER_LINEAR, depth / labels INTER_NEAREST (volumetric_integrator_base.py:1017-1043)."""

    def __init__(self, volume, K, D, width, height, use_optimal_new_K=True, alpha=0.7):
        self.volume = volume
        K = np.asarray(K, dtype=np.float64)
        self.new_K = get_optimal_new_camera_matrix(K, D, (width, height), alpha, (width, height))[0] if use_optimal_new_K else K
        self.map_x, self.map_y = init_undistort_rectify_map(K, D, self.new_K, (width, height))

    @property
    def intrinsics(self):
        return float(self.new_K[0, 0]), float(self.new_K[1, 1]), float(self.new_K[0, 2]), float(self.new_K[1, 2])

    def color(self, img_u8):
        return self.volume.remap(img_u8, self.map_x, self.map_y, linear=True)

    def depth(self, depth_f32):
        return self.volume.remap(depth_f32, self.map_x, self.map_y, linear=False)

    def labels(self, label_i32):
        return self.volume.remap(label_i32, self.map_x, self.map_y, linear=False)
