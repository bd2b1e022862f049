"""Binary little-endian PLY writers for save(): the on-disk format the reference produces through
o3d.io.write_triangle_mesh / write_point_cloud (volumetric_integrator_tsdf.py:233-247,
volumetric_integrator_voxel_grid.py:310-333): double vertices, uchar colours, int face indices."""
import numpy as np


def _colors_u8(colors):
    return np.clip(np.rint(np.asarray(colors, dtype=np.float64) * 255.0), 0, 255).astype(np.uint8)


def write_ply_points(path, points, colors=None, normals=None):
    """Property order as Open3D's WritePointCloudToPLY: x y z, then nx ny nz when the cloud has normals, then the colours."""
    points = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    has_c = colors is not None and len(colors) == len(points)
    has_n = normals is not None and len(normals) == len(points)
    header = ["ply", "format binary_little_endian 1.0", "comment Created by pyslam_amd", f"element vertex {len(points)}",
              "property double x", "property double y", "property double z"]
    fields = [("x", "<f8"), ("y", "<f8"), ("z", "<f8")]
    if has_n:
        header += ["property double nx", "property double ny", "property double nz"]
        fields += [("nx", "<f8"), ("ny", "<f8"), ("nz", "<f8")]
    if has_c:
        header += ["property uchar red", "property uchar green", "property uchar blue"]
        fields += [("r", "u1"), ("g", "u1"), ("b", "u1")]
    header.append("end_header")
    rec = np.zeros(len(points), dtype=fields)
    rec["x"], rec["y"], rec["z"] = points[:, 0], points[:, 1], points[:, 2]
    if has_n:
        nn = np.asarray(normals, dtype=np.float64).reshape(-1, 3)
        rec["nx"], rec["ny"], rec["nz"] = nn[:, 0], nn[:, 1], nn[:, 2]
    if has_c:
        c = _colors_u8(colors)
        rec["r"], rec["g"], rec["b"] = c[:, 0], c[:, 1], c[:, 2]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(rec.tobytes())


def write_ply_mesh(path, vertices, triangles, vertex_colors=None):
    vertices = np.asarray(vertices, dtype=np.float64).reshape(-1, 3)
    triangles = np.asarray(triangles, dtype=np.int32).reshape(-1, 3)
    has_c = vertex_colors is not None and len(vertex_colors) == len(vertices)
    header = ["ply", "format binary_little_endian 1.0", "comment Created by pyslam_amd", f"element vertex {len(vertices)}",
              "property double x", "property double y", "property double z"]
    fields = [("x", "<f8"), ("y", "<f8"), ("z", "<f8")]
    if has_c:
        header += ["property uchar red", "property uchar green", "property uchar blue"]
        fields += [("r", "u1"), ("g", "u1"), ("b", "u1")]
    header += [f"element face {len(triangles)}", "property list uchar uint vertex_indices", "end_header"]
    vrec = np.zeros(len(vertices), dtype=fields)
    vrec["x"], vrec["y"], vrec["z"] = vertices[:, 0], vertices[:, 1], vertices[:, 2]
    if has_c:
        c = _colors_u8(vertex_colors)
        vrec["r"], vrec["g"], vrec["b"] = c[:, 0], c[:, 1], c[:, 2]
    frec = np.zeros(len(triangles), dtype=[("n", "u1"), ("a", "<u4"), ("b", "<u4"), ("c", "<u4")])
    frec["n"] = 3
    frec["a"], frec["b"], frec["c"] = triangles[:, 0], triangles[:, 1], triangles[:, 2]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(vrec.tobytes())
        f.write(frec.tobytes())


def read_ply(path):
    """Minimal reader for the two layouts written above (tests / round trips)."""
    with open(path, "rb") as f:
        lines = []
        while True:
            line = f.readline().decode("ascii").strip()
            lines.append(line)
            if line == "end_header":
                break
        nv = int([l for l in lines if l.startswith("element vertex")][0].split()[-1])
        nf_l = [l for l in lines if l.startswith("element face")]
        nf = int(nf_l[0].split()[-1]) if nf_l else 0
        has_c = any("uchar red" in l for l in lines)
        has_n = any("double nx" in l for l in lines)
        fields = ([("x", "<f8"), ("y", "<f8"), ("z", "<f8")] + ([("nx", "<f8"), ("ny", "<f8"), ("nz", "<f8")] if has_n else [])
                  + ([("r", "u1"), ("g", "u1"), ("b", "u1")] if has_c else []))
        v = np.frombuffer(f.read(nv * np.dtype(fields).itemsize), dtype=fields)
        faces = None
        if nf:
            fdt = np.dtype([("n", "u1"), ("a", "<u4"), ("b", "<u4"), ("c", "<u4")])
            fr = np.frombuffer(f.read(nf * fdt.itemsize), dtype=fdt)
            faces = np.stack([fr["a"], fr["b"], fr["c"]], axis=1).astype(np.int32)
    pts = np.stack([v["x"], v["y"], v["z"]], axis=1)
    cols = np.stack([v["r"], v["g"], v["b"]], axis=1) if has_c else None
    return pts, cols, faces


def read_ply_normals(path):
    """-> [N,3] normals of a point-cloud PLY written above, or None."""
    with open(path, "rb") as f:
        lines = []
        while True:
            line = f.readline().decode("ascii").strip()
            lines.append(line)
            if line == "end_header":
                break
        if not any("double nx" in l for l in lines):
            return None
        nv = int([l for l in lines if l.startswith("element vertex")][0].split()[-1])
        has_c = any("uchar red" in l for l in lines)
        fields = [("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("nx", "<f8"), ("ny", "<f8"), ("nz", "<f8")] + ([("r", "u1"), ("g", "u1"), ("b", "u1")] if has_c else [])
        v = np.frombuffer(f.read(nv * np.dtype(fields).itemsize), dtype=fields)
    return np.stack([v["nx"], v["ny"], v["nz"]], axis=1)
