"""CPU: second set of hand-derived known-answer tests for the TSDF restatement (oracle/tsdf_oracle.c) - VERDICT r02 #2.

PARITY UNPINNED (Open3D is not installed): what anchors the restatement are evaluators written HERE from the algorithm as
SURVEY.md 8a T1-T6 states it (reference call sites pyslam/dense/volumetric_integrator_tsdf.py:104-108,215-223,239-267), not
from the oracle's source:
  * `expected_voxel` - one voxel of one frame for an ARBITRARY pose and depth / colour image, numpy float32;
  * `numpy_vertices` - the vertex half of marching cubes (which edges carry a vertex, where, with what colour) from a dumped
    volume; it uses no case table;
and analytic facts about a sphere (closed surface, Euler characteristic 2, area 4 pi r^2, vertices on grid edges).
"""
import numpy as np

import oracle
from pyslam_amd.synthetic import look_at_pose

f32 = np.float32
VL, TRUNC = 0.02, 0.08
UNIT = 16 * VL
TRUNC_INV = f32(1.0) / f32(TRUNC)


def tilted_plane_frame(W, H, K, T_wc, n, d, colour_fn):
    """z-depth image of the world plane n.p = d seen from camera T_wc, and a colour image that depends on the pixel."""
    fx, fy, cx, cy = K
    v, u = np.mgrid[0:H, 0:W].astype(np.float64)
    dirs = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], axis=-1) @ T_wc[:3, :3].T
    eye = T_wc[:3, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (d - eye @ n) / (dirs @ n)
    depth = np.where(np.isfinite(t) & (t > 0.05), t, 0.0).astype(np.float32)
    return depth, colour_fn(u, v).astype(np.uint8)


def expected_voxel(ix, iy, iz, depth, rgb, K, T_cw, depth_trunc=4.0):
    """Open3D's per-voxel update (T4) for global voxel index (ix, iy, iz), evaluated independently in numpy float32.
    -> None (not updated) or (tsdf contribution, rgb, fragile) where `fragile` marks a voxel whose pixel choice or truncation
    decision sits within rounding distance of a boundary (the closed form used here and the algorithm's z-walk differ by ~1e-6)."""
    H, W = depth.shape
    fx, fy, cx, cy = (f32(x) for x in K)
    E = T_cw.astype(np.float32)
    half = f32(VL) * f32(0.5)
    c = [f32(np.float64(half + f32(VL) * f32(i % 16)) + np.float64(i // 16) * UNIT) for i in (ix, iy, iz)]
    pc = [((E[r, 0] * c[0] + E[r, 1] * c[1]) + E[r, 2] * c[2]) + E[r, 3] for r in range(3)]
    if not pc[2] > 0:
        return None
    u_f = pc[0] * fx / pc[2] + cx + f32(0.5)
    v_f = pc[1] * fy / pc[2] + cy + f32(0.5)
    if not (u_f >= f32(0.0001) and u_f < f32(W) - f32(0.0001) and v_f >= f32(0.0001) and v_f < f32(H) - f32(0.0001)):
        return None
    u, v = int(u_f), int(v_f)
    dd = depth[v, u]
    if dd >= depth_trunc or not dd > 0:
        return None
    xx = (f32(u) - cx) * (f32(1.0) / fx)
    yy = (f32(v) - cy) * (f32(1.0) / fy)
    m = np.sqrt(xx * xx + yy * yy + f32(1.0), dtype=f32)
    sdf = (dd - pc[2]) * m
    fragile = (min(abs(float(u_f) - round(float(u_f))), abs(float(v_f) - round(float(v_f)))) < 2e-3 or abs(float(sdf) + TRUNC) < 1e-4
               or min(float(u_f), float(v_f), W - float(u_f), H - float(v_f)) < 2e-3)
    if not sdf > -f32(TRUNC):
        return (None, None, fragile)
    return (min(f32(1.0), sdf * TRUNC_INV), rgb[v, u].astype(np.float64), fragile)


def touched_units(depth, K, T_cw, depth_trunc=4.0, stride=4):
    """ScalableTSDFVolume::Integrate's unit touching (T3), independently: every `stride`-th pixel with a valid depth is
    back-projected in float64 and opens the units floor((p -/+ sdf_trunc) / unit_length).  A frame updates ONLY these units."""
    fx, fy, cx, cy = K
    T_wc = np.linalg.inv(T_cw)
    out = set()
    H, W = depth.shape
    for i in range(0, H, stride):
        for j in range(0, W, stride):
            d = float(depth[i, j])
            if not (d > 0 and d < depth_trunc):
                continue
            p = T_wc[:3, :3] @ np.array([(j - cx) * d / fx, (i - cy) * d / fy, d]) + T_wc[:3, 3]
            lo, hi = np.floor((p - TRUNC) / UNIT).astype(int), np.floor((p + TRUNC) / UNIT).astype(int)
            out.update((x, y, z) for x in range(lo[0], hi[0] + 1) for y in range(lo[1], hi[1] + 1) for z in range(lo[2], hi[2] + 1))
    return out


def check_against_evaluator(vol_dump, frames, K, sample=150, seed=0, exact=False):
    """Every sampled voxel of every unit: weight == number of frames that touch the unit AND whose evaluator accepts the voxel,
    tsdf / colour == the mean of those frames' contributions.  Also: the volume's units == the union of the frames' touched sets."""
    keys, tsdf, weight, color = vol_dump
    rng = np.random.default_rng(seed)
    n_checked = n_updated = n_fragile = 0
    touched = [touched_units(depth, K, T_cw) for depth, _, T_cw in frames]
    assert {tuple(k) for k in keys} == set().union(*touched)
    for ui in range(len(keys)):
        for lin in rng.choice(4096, size=sample, replace=False):
            x, y, z = lin // 256, (lin // 16) % 16, lin % 16  # Open3D IndexOf order
            g = [int(keys[ui][a]) * 16 + b for a, b in enumerate((x, y, z))]
            ts, cs, fragile = [], [], False
            for fi, (depth, rgb, T_cw) in enumerate(frames):
                if tuple(keys[ui]) not in touched[fi]:
                    continue
                e = expected_voxel(*g, depth, rgb, K, T_cw)
                if e is None:
                    continue
                fragile |= e[2]
                if e[0] is not None:
                    ts.append(e[0])
                    cs.append(e[1])
            if fragile and not exact:
                n_fragile += 1
                continue
            n_checked += 1
            assert weight[ui, lin] == len(ts), (keys[ui], lin, weight[ui, lin], len(ts))
            if ts:
                n_updated += 1
                assert abs(float(tsdf[ui, lin]) - float(np.mean(np.array(ts, np.float64)))) <= 2e-5, (keys[ui], lin)
                np.testing.assert_allclose(color[ui, lin], np.mean(np.array(cs), axis=0), rtol=0, atol=1e-9)
            else:
                assert tsdf[ui, lin] == 0
    return n_checked, n_updated, n_fragile


def test_rotated_and_translated_pose_values():
    """A tilted world plane seen from two rotated + translated cameras: VALUES (tsdf, weight, colour per voxel), not only where
    the surface lands.  Depth and colour vary from pixel to pixel, so picking the wrong pixel or the wrong multiplier shows."""
    W, H, K = 96, 72, np.array([90.0, 88.0, 47.3, 35.6])
    n = np.array([0.2, -0.3, 1.0])
    n = n / np.linalg.norm(n)
    frames = []
    for eye, target in ((np.array([0.3, -0.2, -0.1]), np.array([0.35, 0.1, 1.2])), (np.array([-0.25, 0.15, 0.05]), np.array([0.1, 0.0, 1.1]))):
        T_cw, T_wc = look_at_pose(eye, target, up=np.array([0.1, -1.0, 0.05]))
        depth, rgb = tilted_plane_frame(W, H, K, T_wc, n, 1.15, lambda u, v: np.stack([2 * u + 10, 3 * v + 5, (u + v) % 256], axis=-1) % 256)
        frames.append((depth, rgb, T_cw))
    vol = oracle.PortTsdf(VL, TRUNC)
    for depth, rgb, T_cw in frames:
        vol.integrate(depth, rgb, K, T_cw, 1.0, 4.0)
    checked, updated, fragile = check_against_evaluator(vol.dump(), frames, K, sample=120)
    assert checked > 5000 and updated > checked // 5 and fragile < checked // 20
    assert vol.dump()[2].max() == 2  # both cameras see part of the plane


def test_image_border_and_camera_plane_decisions_are_exact():
    """u_f just inside / just outside [0.0001, W - 0.0001) and voxels at or behind the camera plane: an intrinsic matrix chosen
    so that the voxel centres of the slab z_c = 1.01 project onto u_f = i + 1 + cx (fx = z_c / voxel: x fx / z_c = i + 0.5);
    cx = -1 + 5e-5 puts the first column at u_f = 5e-5 (rejected), cx = -1 + 2e-4 at 2e-4 (accepted, pixel 0)."""
    W, H = 40, 40
    for cx, first_col_in in ((-1.0 + 5e-5, False), (-1.0 + 2e-4, True)):
        K = np.array([1.01 / VL, 1.01 / VL, cx, 14.5])
        depth = np.full((H, W), 1.0, np.float32)
        rgb = np.full((H, W, 3), 77, np.uint8)
        vol = oracle.PortTsdf(VL, TRUNC)
        vol.integrate(depth, rgb, K, np.eye(4), 1.0, 4.0)
        # every voxel, no exclusions: on this lattice nothing but the designed column is near a decision boundary in u
        keys, tsdf, weight, color = vol.dump()
        idx = {tuple(k): i for i, k in enumerate(keys)}
        assert (0, 0, 3) in idx  # unit holding x in [0, 0.32), z in [0.96, 1.28): voxel z index 50 -> z_c = 1.01
        ui = idx[(0, 0, 3)]
        col0 = [weight[ui, 0 * 256 + y * 16 + 2] for y in range(16)]   # x index 0: x_c = 0.01 -> x fx = 0.5 -> u_f = 1 + cx
        col1 = [weight[ui, 1 * 256 + y * 16 + 2] for y in range(16)]   # x index 1: u_f = 2 + cx: always inside
        assert all(w == (1 if first_col_in else 0) for w in col0), (cx, col0)
        assert all(w == 1 for w in col1)
        e = expected_voxel(0, 3, 50, depth, rgb, K, np.eye(4))
        assert (e is not None) == first_col_in
    # a plane 5 cm in front of the camera: units straddle the camera plane; voxels with z_c <= 0 are never touched
    K = np.array([60.0, 60.0, 31.5, 23.5])
    depth = np.full((48, 64), 0.05, np.float32)
    rgb = np.full((48, 64, 3), 9, np.uint8)
    vol = oracle.PortTsdf(VL, TRUNC)
    vol.integrate(depth, rgb, K, np.eye(4), 1.0, 4.0)
    keys, tsdf, weight, _ = vol.dump()
    assert {tuple(k)[2] for k in keys} == {-1, 0}
    for ui, k in enumerate(keys):
        zc = (k[2] * 16 + np.arange(16) + 0.5) * VL
        w = weight[ui].reshape(16, 16, 16)  # [x, y, z]
        assert (w[:, :, zc <= 0] == 0).all()
    # every voxel of the eight units against the evaluator (the frustum is ~3 voxels wide this close to the camera)
    checked, updated, _ = check_against_evaluator(vol.dump(), [(depth, rgb, np.eye(4))], K, sample=4096, seed=3)
    assert checked > 30000 and updated >= 20


def test_stride_4_samples_on_the_last_row_and_column():
    """PointCloud::CreateFromDepthImage(stride 4) visits rows / columns 0, 4, 8, ...: in a 65 x 49 image the LAST row (48) and
    column (64) are sampled, in a 66 x 50 image they are not.  One valid depth pixel in that corner decides."""
    for (W, H), hit in (((65, 49), True), ((66, 50), False)):
        K = np.array([60.0, 60.0, (W - 1) / 2, (H - 1) / 2])
        depth = np.zeros((H, W), np.float32)
        depth[H - 1, W - 1] = 1.0
        rgb = np.zeros((H, W, 3), np.uint8)
        vol = oracle.PortTsdf(VL, TRUNC)
        vol.integrate(depth, rgb, K, np.eye(4), 1.0, 4.0)
        if not hit:
            assert vol.num_units() == 0
            continue
        p = np.array([(W - 1 - K[2]) / K[0], (H - 1 - K[3]) / K[1], 1.0])
        lo, hi = np.floor((p - TRUNC) / UNIT).astype(int), np.floor((p + TRUNC) / UNIT).astype(int)
        want = {(x, y, z) for x in range(lo[0], hi[0] + 1) for y in range(lo[1], hi[1] + 1) for z in range(lo[2], hi[2] + 1)}
        assert {tuple(k) for k in vol.touched_keys()} == want
        # only voxels that project into that one pixel are updated
        keys, tsdf, weight, _ = vol.dump()
        assert 0 < (weight > 0).sum() < 400


def test_three_observations_one_rejected_by_truncation():
    """Plane at 1.00 m twice, then at 0.90 m: a voxel deeper than 0.90 + sdf_trunc (along its ray) is 'behind' the third
    surface and must NOT be updated by it - weight 2 with the first two frames' mean - while nearer voxels have weight 3."""
    W, H, K = 64, 48, np.array([60.0, 60.0, 31.5, 23.5])
    rgb = np.full((H, W, 3), 120, np.uint8)
    frames = [(np.full((H, W), z, np.float32), rgb, np.eye(4)) for z in (1.0, 1.0, 0.9)]
    vol = oracle.PortTsdf(VL, TRUNC)
    for d, c, T in frames:
        vol.integrate(d, c, K, T, 1.0, 4.0)
    keys, tsdf, weight, _ = vol.dump()
    assert set(np.unique(weight)) >= {0.0, 2.0, 3.0}
    checked, updated, fragile = check_against_evaluator((keys, tsdf, weight, _), frames, K, sample=200, seed=1)
    assert updated > 1000
    # along the optical axis: z_c <= 0.98 -> 3 observations, z_c >= 0.99 -> 2 (sdf_3 = 0.90 - z_c <= -0.08)
    idx = {tuple(k): i for i, k in enumerate(keys)}
    ui = idx[(0, 0, 3)]  # z in [0.96, 1.28)
    w = weight[ui].reshape(16, 16, 16)[0, 0]
    assert w[0] == 3 and w[1] == 2 and w[2] == 2  # z_c = 0.97, 0.99, 1.01


def numpy_vertices(keys, tsdf, weight, color):
    """Marching-cubes VERTICES from a dumped volume, without a case table: an edge (voxel a -> its +axis neighbour b) carries a
    vertex iff both are observed, tsdf_a < 0 differs from tsdf_b < 0, and at least one of the four cubes around the edge has all
    eight corners observed; position = a + f_a / (f_a - f_b) along the axis (voxel CENTRES), colour lerped the same way / 255."""
    idx = {tuple(k): i for i, k in enumerate(keys)}
    lo, hi = keys.min(axis=0), keys.max(axis=0) + 1
    shape = tuple((hi - lo) * 16)
    T = np.zeros(shape, np.float32)
    Wt = np.zeros(shape, np.float32)
    C = np.zeros(shape + (3,), np.float64)
    for k, i in idx.items():
        o = (np.array(k) - lo) * 16
        sl = tuple(slice(o[a], o[a] + 16) for a in range(3))
        T[sl], Wt[sl], C[sl] = tsdf[i].reshape(16, 16, 16), weight[i].reshape(16, 16, 16), color[i].reshape(16, 16, 16, 3)
    obs = Wt != 0
    cube_ok = np.ones(tuple(s - 1 for s in shape), bool)  # cube at (x, y, z) spans corners x..x+1 etc.
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                cube_ok &= obs[dx:shape[0] - 1 + dx, dy:shape[1] - 1 + dy, dz:shape[2] - 1 + dz]
    out = []
    for axis in range(3):
        sa = [slice(0, s) for s in shape]
        sb = [slice(0, s) for s in shape]
        sa[axis], sb[axis] = slice(0, shape[axis] - 1), slice(1, shape[axis])
        fa, fb = T[tuple(sa)], T[tuple(sb)]
        crossing = obs[tuple(sa)] & obs[tuple(sb)] & ((fa < 0) != (fb < 0))
        # the four cubes that share the edge: offsets 0 / -1 in the two other axes
        any_cube = np.zeros_like(crossing)
        others = [a for a in range(3) if a != axis]
        padded = np.zeros(tuple(s + 1 for s in cube_ok.shape), bool)
        padded[1:, 1:, 1:] = cube_ok  # padded[x+1, y+1, z+1] = cube (x, y, z); index 0 = cube -1 (does not exist)
        ext = np.zeros(tuple(s + 2 for s in shape), bool)
        ext[1:shape[0], 1:shape[1], 1:shape[2]] = cube_ok
        for d0 in (0, -1):
            for d1 in (0, -1):
                off = [0, 0, 0]
                off[others[0]], off[others[1]] = d0, d1
                sl = tuple(slice(1 + off[a], 1 + off[a] + crossing.shape[a]) for a in range(3))
                any_cube |= ext[sl]
        sel = crossing & any_cube
        ia = np.argwhere(sel)
        t = (fa[sel].astype(np.float64)) / (fa[sel].astype(np.float64) - fb[sel].astype(np.float64))
        pos = (ia + lo * 16 + 0.5) * VL
        pos[:, axis] += t * VL
        ca, cb = C[tuple(sa)][sel], C[tuple(sb)][sel]
        out.append(np.concatenate([pos, (ca + t[:, None] * (cb - ca)) / 255.0], axis=1))
    return np.concatenate(out)


def test_mesh_vertices_and_colour_lerp_against_a_table_free_evaluator():
    """Vertex positions and LERPED colours of extract_triangle_mesh vs numpy_vertices on a tilted plane whose colour changes
    from pixel to pixel (neighbouring voxels hold different colours, so the lerp weight matters), two views (mixed weights)."""
    W, H, K = 96, 72, np.array([90.0, 88.0, 47.3, 35.6])
    n = np.array([0.25, 0.15, 1.0])
    n = n / np.linalg.norm(n)
    vol = oracle.PortTsdf(VL, TRUNC)
    for eye, target in ((np.array([0.1, 0.0, 0.0]), np.array([0.1, 0.05, 1.0])), (np.array([-0.3, 0.2, 0.1]), np.array([0.0, 0.0, 1.1]))):
        T_cw, T_wc = look_at_pose(eye, target, up=np.array([0.0, -1.0, 0.0]))
        depth, rgb = tilted_plane_frame(W, H, K, T_wc, n, 1.1, lambda u, v: np.stack([2.5 * u, 255 - 3 * v, 40 + u + v], axis=-1) % 256)
        vol.integrate(depth, rgb, K, T_cw, 1.0, 4.0)
    verts, tris, cols = vol.extract_triangle_mesh()
    want = numpy_vertices(*vol.dump())
    assert len(verts) == len(want) > 3000
    got = np.concatenate([verts, cols], axis=1)
    got = got[np.lexsort(np.round(got[:, :3], 9).T[::-1])]
    want = want[np.lexsort(np.round(want[:, :3], 9).T[::-1])]
    np.testing.assert_allclose(got[:, :3], want[:, :3], rtol=0, atol=1e-9)
    np.testing.assert_allclose(got[:, 3:], want[:, 3:], rtol=0, atol=1e-9)
    assert np.ptp(cols, axis=0).min() > 0.2  # the colours really vary over the surface
    assert tris.min() >= 0 and tris.max() < len(verts)


def sphere_frame(W, H, K, T_wc, centre, radius):
    fx, fy, cx, cy = K
    v, u = np.mgrid[0:H, 0:W].astype(np.float64)
    dirs = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], axis=-1) @ T_wc[:3, :3].T
    oc = T_wc[:3, 3] - centre
    a = (dirs * dirs).sum(-1)
    b = 2.0 * (dirs @ oc)
    c = oc @ oc - radius * radius
    disc = b * b - 4 * a * c
    t = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), 0.0)
    depth = np.where(t > 0.05, t, 0.0).astype(np.float32)
    rgb = np.zeros((H, W, 3), np.uint8)
    rgb[..., 0], rgb[..., 1], rgb[..., 2] = 200, 150, 100
    return depth, rgb


def test_sphere_is_a_closed_surface_of_the_right_size():
    """A sphere (r = 0.3 m) seen from 14 directions.  Analytic facts the fused volume and its mesh must satisfy: observed voxels
    well inside are negative, well outside positive; every mesh vertex lies ON a grid edge (two coordinates are voxel-centre
    coordinates); the mesh is closed and oriented (every edge shared by exactly two triangles, opposite directions), has Euler
    characteristic 2, its vertices sit on the sphere to a fraction of a voxel (0.6 max, 0.15 mean) and its area is 4 pi r^2 within 5 % (measured +3.5 %)."""
    W, H, K = 160, 120, np.array([150.0, 150.0, 79.5, 59.5])
    centre, radius = np.array([0.13, -0.07, 0.21]), 0.3
    dirs = [np.array(d, float) for d in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (1, 1, 1), (1, 1, -1), (1, -1, 1),
                                         (1, -1, -1), (-1, 1, 1), (-1, 1, -1), (-1, -1, 1), (-1, -1, -1))]
    vol = oracle.PortTsdf(VL, TRUNC, threads=4)
    for d in dirs:
        eye = centre + 1.0 * d / np.linalg.norm(d)
        up = np.array([0.0, 0.0, 1.0]) if abs(d[2]) < 0.9 * np.linalg.norm(d) else np.array([0.0, 1.0, 0.0])
        T_cw, T_wc = look_at_pose(eye, centre, up=up)
        depth, rgb = sphere_frame(W, H, K, T_wc, centre, radius)
        vol.integrate(depth, rgb, K, T_cw, 1.0, 4.0)
    keys, tsdf, weight, _ = vol.dump()
    assert 1 <= weight.max() <= len(dirs)
    # sign structure
    grid = (np.arange(16) + 0.5) * VL
    for ui, k in enumerate(keys):
        o = k * UNIT
        X, Y, Z = np.meshgrid(o[0] + grid, o[1] + grid, o[2] + grid, indexing="ij")
        r = np.sqrt((X - centre[0]) ** 2 + (Y - centre[1]) ** 2 + (Z - centre[2]) ** 2).reshape(-1)
        w, t = weight[ui], tsdf[ui]
        assert (t[(w > 0) & (r < radius - 1.5 * VL)] < 0).all()
        assert (t[(w > 0) & (r > radius + 1.5 * VL)] > 0).all()
    verts, tris, cols = vol.extract_triangle_mesh()
    assert len(tris) > 5000
    # on-edge property
    frac = np.abs((verts / VL - 0.5) - np.round(verts / VL - 0.5))
    assert ((frac < 1e-6).sum(axis=1) >= 2).all()
    # closed, oriented, genus 0
    e = np.concatenate([tris[:, [0, 1]], tris[:, [1, 2]], tris[:, [2, 0]]])
    und, inv, cnt = np.unique(np.sort(e, axis=1), axis=0, return_inverse=True, return_counts=True)
    assert (cnt == 2).all()
    tot = np.zeros(len(cnt), np.int64)
    np.add.at(tot, inv.ravel(), np.where(e[:, 0] < e[:, 1], 1, -1))
    assert (tot == 0).all()
    assert len(verts) - len(und) + len(tris) == 2
    # size and place
    rv = np.linalg.norm(verts - centre, axis=1)
    # (a projective TSDF measures along the viewing ray: near the silhouettes of a view the zero crossing sits up to ~half a voxel off)
    assert np.abs(rv - radius).max() < 0.6 * VL and np.abs(rv - radius).mean() < 0.15 * VL
    a, b, c = verts[tris[:, 0]], verts[tris[:, 1]], verts[tris[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1).sum()
    # measured +3.5 %: at 15 voxels of radius the faceted surface and the outward bias of the silhouette zones (above) add up
    assert 0.0 < area / (4 * np.pi * radius ** 2) - 1.0 < 0.05
    nrm = np.cross(b - a, c - a)
    assert (np.einsum("ij,ij->i", nrm, (a + b + c) / 3 - centre) > 0).mean() > 0.999  # outward orientation
    np.testing.assert_allclose(cols, np.broadcast_to(np.array([200, 150, 100]) / 255.0, cols.shape), atol=1e-9)


def test_case_table_matches_what_is_known_of_bourkes_listing():
    """Facts about Paul Bourke's public triTable that do not come from this repository's typed copy: 820 triangles in total,
    at most 5 per case, the distribution of triangles per case, and a few rows quoted in every description of the method."""
    from tools import gen_mc_tables

    rows = gen_mc_tables.parse_rows()
    n = [len(r) // 3 for r in rows]
    assert len(rows) == 256 and sum(n) == 820 and max(n) == 5
    assert [n.count(k) for k in range(6)] == [2, 16, 50, 80, 76, 32]
    known = {0: [], 1: [0, 8, 3], 2: [0, 1, 9], 3: [1, 8, 3, 9, 8, 1], 4: [1, 2, 10], 8: [3, 11, 2], 16: [4, 7, 8], 32: [9, 5, 4],
             64: [10, 6, 5], 127: [7, 11, 6], 128: [7, 6, 11], 254: [0, 3, 8], 255: []}
    for case, tri in known.items():
        assert list(rows[case]) == tri, case


def test_point_cloud_normals_on_a_plane_and_a_sphere():
    """GetNormalAt = normalised central difference (+/- 0.99 voxel) of the trilinearly interpolated tsdf.  tsdf grows towards
    the camera, so on a fronto-parallel plane every interior normal is (0, 0, -1); on the sphere seen from all around the
    normals are the outward radial directions."""
    W, H, K = 64, 48, np.array([60.0, 60.0, 31.5, 23.5])
    vol = oracle.PortTsdf(VL, TRUNC)
    vol.integrate(np.full((H, W), 1.0, np.float32), np.full((H, W, 3), 50, np.uint8), K, np.eye(4), 1.0, 4.0)
    p, _ = vol.extract_point_cloud()
    n = vol.point_normals(p)
    inner = (np.abs(p[:, 0]) < 0.3) & (np.abs(p[:, 1]) < 0.2)
    assert inner.sum() > 500
    np.testing.assert_allclose(np.linalg.norm(n[inner], axis=1), 1.0, atol=1e-12)
    assert (n[inner, 2] < -0.999).all() and np.abs(n[inner, :2]).max() < 0.05
    # sphere
    W, H, K = 160, 120, np.array([150.0, 150.0, 79.5, 59.5])
    centre, radius = np.array([0.13, -0.07, 0.21]), 0.3
    vol = oracle.PortTsdf(VL, TRUNC, threads=4)
    for d in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (1, 1, 1), (-1, -1, 1), (1, -1, -1), (-1, 1, -1)):
        d = np.array(d, float)
        up = np.array([0.0, 0.0, 1.0]) if abs(d[2]) < 0.9 * np.linalg.norm(d) else np.array([0.0, 1.0, 0.0])
        T_cw, T_wc = look_at_pose(centre + d / np.linalg.norm(d), centre, up=up)
        depth, rgb = sphere_frame(W, H, K, T_wc, centre, radius)
        vol.integrate(depth, rgb, K, T_cw, 1.0, 4.0)
    p, _ = vol.extract_point_cloud()
    n = vol.point_normals(p)
    radial = (p - centre) / np.linalg.norm(p - centre, axis=1, keepdims=True)
    cos = np.einsum("ij,ij->i", n, radial)
    assert len(p) > 3000 and cos.min() > 0.8 and cos.mean() > 0.98  # (measured: mean 0.986, min 0.85 where unobserved voxels - tsdf 0 - enter the stencil)


def test_extraction_decisions_on_loaded_voxel_states():
    """Hand-placed voxel states through the oracle's load hook (`to_load_unit`), answers from Open3D's rules as SURVEY 8a T5 / T6
    states them: ExtractPointCloud takes a voxel pair (v, v + e_axis) iff both are observed, both tsdf lie in [-0.98, 0.98) and
    f0 * f1 < 0, at p0 + |f0| / (|f0| + |f1|) * voxel along the axis; marching cubes puts a vertex on a crossed edge of a cube
    whose eight corners are all observed (negative means f < 0: an exact 0 is outside)."""
    R, vl = 16, 0.01
    vol = oracle.PortTsdf(vl, 0.04)
    t = np.ones((1, R, R, R), np.float32)
    w = np.ones((1, R, R, R), np.float32)
    c = np.zeros((1, R, R, R, 3))
    pairs = {  # x -> (f at (x, 4, 4), f at (x, 4, 5)): only the z edge of the pair can cross
        0: (-0.98, 0.5),                                   # lower bound is inside the range: point
        1: (0.5, -0.98),                                   # ... in either order
        2: (0.98, -0.5),                                   # upper bound is outside: no point
        3: (float(np.nextafter(f32(0.98), f32(0))), -0.5),  # just below it: point
        4: (float(np.nextafter(f32(-0.98), f32(-2))), 0.5),  # just below the lower bound: no point
        5: (0.0, -0.5),                                    # product is -0: no point
        6: (-0.0, 0.5),                                    # ... for either zero
        7: (-0.25, 0.75),                                  # a plain crossing: a quarter of the way along the edge
    }
    for x, (f0, f1) in pairs.items():
        t[0, x, 4, 4], t[0, x, 4, 5] = f0, f1
    # everything else holds tsdf = +1 (outside the point cloud's range): the only candidate crossings are between pair voxels,
    # along z inside a pair and along x between neighbouring pairs - the evaluator below enumerates all of them
    vol.load_units(np.array([[0, 0, 0]], np.int32), t, w, c)
    pts, _ = vol.extract_point_cloud()
    got = {tuple(np.round(p / vl - 0.5, 6)) for p in pts}
    want = set()
    f = t[0].astype(np.float64)
    inr = lambda v: -0.98 <= np.float32(v) < np.float32(0.98)
    for x in range(R):
        for y in range(R):
            for z in range(R):
                for axis, (dx, dy, dz) in enumerate(((1, 0, 0), (0, 1, 0), (0, 0, 1))):
                    x1, y1, z1 = x + dx, y + dy, z + dz
                    if x1 >= R or y1 >= R or z1 >= R:
                        continue  # the neighbouring unit does not exist
                    f0, f1 = np.float32(f[x, y, z]), np.float32(f[x1, y1, z1])
                    if inr(f0) and inr(f1) and np.float32(f0 * f1) < 0:
                        p = [float(x), float(y), float(z)]
                        r0, r1 = abs(float(f0)), abs(float(f1))
                        p[axis] = (p[axis] * r1 + (p[axis] + 1.0) * r0) / (r0 + r1)
                        want.add(tuple(np.round(p, 6)))
    assert got == want
    zedge = {x: any(abs(p[0] - x) < 1e-9 and abs(p[1] - 4) < 1e-9 and 4 < p[2] < 5 for p in got) for x in pairs}
    assert zedge == {0: True, 1: True, 2: False, 3: True, 4: False, 5: False, 6: False, 7: True}
    assert (7.0, 4.0, 4.25) in got
    # marching cubes: an unobserved corner removes every cube around it; a 0 corner is "outside"
    t2 = np.ones((1, R, R, R), np.float32)
    t2[0, :, :, :8] = -1.0  # surface between z = 7 and z = 8, everywhere
    w2 = np.ones((1, R, R, R), np.float32)
    w2[0, 5, 5, 7] = 0.0  # one unobserved voxel on the surface: the 8 cubes touching it are invalid
    t2[0, 5, 5, 7] = 0.0
    t2[0, 10, 10, 7] = 0.0  # an exact zero below the surface: its vertical edge crosses between z = 6 and the voxel itself
    vol = oracle.PortTsdf(vl, 0.04)
    vol.load_units(np.array([[0, 0, 0]], np.int32), t2, w2, np.zeros((1, R, R, R, 3)))
    v, tri, _ = vol.extract_triangle_mesh()
    g = np.round(v / vl - 0.5, 6)
    # full cube layer: 15 x 15 cubes, two triangles each, minus the four cubes of that layer around the unobserved voxel
    flat = 2 * (15 * 15 - 4)
    assert len(tri) >= flat
    on_edge_above = lambda x, y: any(abs(p[0] - x) < 1e-9 and abs(p[1] - y) < 1e-9 and 7 <= p[2] <= 8 for p in g)
    assert not on_edge_above(5, 5)  # no vertex on the edge that starts at the unobserved voxel
    assert on_edge_above(3, 3)
    assert any(np.allclose(p, (10, 10, 7.0)) for p in g)  # the zero voxel: crossing (z 6 -> 7) lands exactly on it
