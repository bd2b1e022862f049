"""``volumetric.BoundingBox2D`` / ``volumetric.OrientedBoundingBox2D`` (cpp/volumetric/bounding_boxes_2d.h / .cpp, bindings
bounding_boxes_module.h:165-258): the image-plane boxes of the module.  Host-side value classes as in the reference (nothing on the
dense path uses them; they complete the module's namespace)."""
import numpy as np

from .volumetric_semantic import OBBComputationMethod


def _rot2(angle):
    c, s = np.cos(angle), np.sin(angle)
    return np.array([[c, -s], [s, c]])


def _sat_intersects_2d(ca, Ra, ha, cb, Rb, hb):
    """bounding_boxes_2d.cpp:37-64: the four face axes, 1e-9 slack."""
    axes = [Ra[:, 0] / np.linalg.norm(Ra[:, 0]), Ra[:, 1] / np.linalg.norm(Ra[:, 1]), Rb[:, 0] / np.linalg.norm(Rb[:, 0]), Rb[:, 1] / np.linalg.norm(Rb[:, 1])]
    a0, a1, b0, b1 = axes
    t = cb - ca
    for L in axes:
        ra = ha[0] * abs(a0 @ L) + ha[1] * abs(a1 @ L)
        rb = hb[0] * abs(b0 @ L) + hb[1] * abs(b1 @ L)
        if abs(t @ L) > ra + rb + 1e-9:
            return False
    return True


class BoundingBox2D:
    """BoundingBox2D(), BoundingBox2D(min_point, max_point) or the four scalars; closed box."""

    def __init__(self, *args, **kw):
        if "min_point" in kw:
            args = (kw["min_point"], kw["max_point"])
        if len(args) == 0:
            vals = (0.0,) * 4
        elif len(args) == 2:
            vals = tuple(float(x) for x in args[0]) + tuple(float(x) for x in args[1])
        elif len(args) == 4:
            vals = tuple(float(x) for x in args)
        else:
            raise TypeError("BoundingBox2D(), BoundingBox2D(min_point, max_point) or BoundingBox2D(min_x, min_y, max_x, max_y)")
        self.min_x, self.min_y, self.max_x, self.max_y = vals

    def get_min_point(self):
        return np.array([self.min_x, self.min_y])

    def get_max_point(self):
        return np.array([self.max_x, self.max_y])

    def get_center(self):
        return np.array([(self.min_x + self.max_x) / 2.0, (self.min_y + self.max_y) / 2.0])

    def get_size(self):
        return np.array([self.max_x - self.min_x, self.max_y - self.min_y])

    def get_area(self):
        return float((self.max_x - self.min_x) * (self.max_y - self.min_y))

    def get_perimeter(self):
        return float(2.0 * ((self.max_x - self.min_x) + (self.max_y - self.min_y)))

    def get_diagonal_length(self):
        sx, sy = self.max_x - self.min_x, self.max_y - self.min_y
        return float(np.sqrt(sx * sx + sy * sy))

    def contains(self, points):
        """One point [2] -> bool; several [N,2] -> list of bool."""
        p = np.asarray(points, np.float64)
        m = (p[..., 0] >= self.min_x) & (p[..., 0] <= self.max_x) & (p[..., 1] >= self.min_y) & (p[..., 1] <= self.max_y)
        return bool(m) if p.ndim == 1 else [bool(x) for x in m]

    def intersects(self, other):
        return bool(self.min_x <= other.max_x and self.max_x >= other.min_x and self.min_y <= other.max_y and self.max_y >= other.min_y)

    @staticmethod
    def compute_from_points(points):
        p = np.asarray(points, np.float64).reshape(-1, 2)
        if len(p) == 0:
            return BoundingBox2D()
        return BoundingBox2D(p.min(axis=0), p.max(axis=0))

    def _frame(self):
        return self.get_center(), np.eye(2), self.get_size() * 0.5

    def __reduce__(self):
        return (BoundingBox2D, (self.min_x, self.min_y, self.max_x, self.max_y))


class OrientedBoundingBox2D:
    """center (2,), angle_rad (rotation about z, object -> world), size (2,)."""

    def __init__(self, center=(0.0, 0.0), angle_rad=0.0, size=(0.0, 0.0)):
        self.center = np.asarray(center, np.float64).copy()
        self.angle_rad = float(angle_rad)
        self.size = np.asarray(size, np.float64).copy()

    def get_volume(self):
        return float(self.size[0] * self.size[1])

    def get_area(self):
        return float(self.size[0] * self.size[1])

    def get_perimeter(self):
        return float(2.0 * (self.size[0] + self.size[1]))

    def get_diagonal_length(self):
        return float(np.sqrt(self.size[0] * self.size[0] + self.size[1] * self.size[1]))

    def get_corners(self):
        """bounding_boxes_2d.cpp:163-178 (same corner order)."""
        R, h = _rot2(self.angle_rad), self.size / 2.0
        return [self.center + R @ (h * np.array(s)) for s in ((1.0, 1.0), (-1.0, 1.0), (-1.0, -1.0), (1.0, -1.0))]

    def contains(self, points):
        """One point [2] -> bool; several [N,2] -> list of bool; closed box with the reference's 1e-10 slack (:185-215)."""
        p = np.asarray(points, np.float64)
        q = (p - self.center) @ _rot2(self.angle_rad)  # rows: R^T (p - c)
        h = self.size / 2.0 + 1e-10
        m = np.all((q >= -h) & (q <= h), axis=-1)
        return bool(m) if p.ndim == 1 else [bool(x) for x in m]

    def _frame(self):
        return self.center, _rot2(self.angle_rad), self.size / 2.0

    def intersects(self, other):
        """Against another OrientedBoundingBox2D or a BoundingBox2D (separating axes, :217-224)."""
        ca, Ra, ha = self._frame()
        cb, Rb, hb = other._frame()
        return _sat_intersects_2d(ca, Ra, ha, cb, Rb, hb)

    @staticmethod
    def compute_from_points(points_w, method=OBBComputationMethod.PCA):
        """compute_obb_pca_2d, bounding_boxes_2d.cpp:228-292 (the principal axis' sign is the eigen-solver's: the box is the
        reference's as a point set, its angle may differ by pi).  The convex-hull variant needs Qhull in the reference as well."""
        if method != OBBComputationMethod.PCA:
            raise NotImplementedError("OBBComputationMethod.CONVEX_HULL_MINIMAL is not provided (PCA is the reference's default)")
        p = np.asarray(points_w, np.float64).reshape(-1, 2)
        n = len(p)
        if n == 0:
            return OrientedBoundingBox2D()
        if n == 1:
            return OrientedBoundingBox2D(p[0], 0.0, (0.0, 0.0))
        centroid = p.sum(axis=0) / float(n)
        d = p - centroid
        cov = (d.T @ d) / float(n)
        w, v = np.linalg.eigh(cov)  # ascending eigenvalues, columns
        axis = v[:, 0] if w[0] >= w[1] else v[:, 1]  # (order = {0, 1} unless eigenvalue 1 is larger)
        angle = float(np.arctan2(axis[1], axis[0]))
        R = _rot2(angle)
        local = d @ R  # rows: R^T (p - centroid)
        lo, hi = local.min(axis=0), local.max(axis=0)
        return OrientedBoundingBox2D(centroid + R @ (0.5 * (hi + lo)), angle, hi - lo)

    def __reduce__(self):
        return (OrientedBoundingBox2D, (self.center, self.angle_rad, self.size))
