"""CPU: oracle/tsdf_oracle.c held to the closed-form whole-frame evaluator of tests/tsdf_closed_form.py on
three full 640x480 frames at the headline's voxel size - a differently structured second opinion on the restatement the headline
is measured against.  PARITY UNPINNED all the same (no Open3D here): see the evaluator's header."""
import os

import numpy as np

import oracle
from tests import tsdf_closed_form as cf


def test_oracle_equals_the_closed_form_evaluator_on_full_frames():
    fr = cf.frames()
    assert all((d > 0).mean() > 0.9 for d, _, _ in fr) and len({d.tobytes() for d, _, _ in fr}) == 3
    ref = cf.evaluate(fr)
    vol = oracle.PortTsdf(cf.VOXEL, cf.TRUNC, threads=min(32, os.cpu_count() or 1))
    for d, c, T in fr:
        vol.integrate(d, c, cf.K, T, 1.0, cf.DEPTH_TRUNC)
    stats = cf.compare(vol.dump(), ref, "oracle/tsdf_oracle.c")
    assert stats["units"] > 3000 and stats["max_weight"] == 3 and stats["updated"] > 4_000_000
    assert stats["fragile_frac"] < 0.05  # the comparison covers > 95 % of the voxels of every touched unit
    # the scene is what it claims to be: the sphere is in front of the plane in the first view's centre
    d0 = fr[0][0]
    assert 0.5 < d0[240, 320] < 1.4 and d0.max() < 4.0


def test_touched_units_match_a_plain_loop_on_a_subsample():
    """The vectorised unit enumeration against the obvious triple loop (every 16th strided sample of one frame)."""
    d, _, T = cf.frames()[1]
    fx, fy, cx, cy = cf.K
    T_wc = np.linalg.inv(T)
    got = {tuple(k) for k in cf.touched_units(d, T).tolist()}
    want = set()
    for i in range(0, cf.H, cf.STRIDE * 4):
        for j in range(0, cf.W, cf.STRIDE * 4):
            z = float(d[i, j])
            if not (0 < z < cf.DEPTH_TRUNC):
                continue
            p = T_wc[:3, :3] @ np.array([(j - cx) * z / fx, (i - cy) * z / fy, z]) + T_wc[:3, 3]
            lo, hi = np.floor((p - cf.TRUNC) / cf.UNIT).astype(int), np.floor((p + cf.TRUNC) / cf.UNIT).astype(int)
            want.update((x, y, zz) for x in range(lo[0], hi[0] + 1) for y in range(lo[1], hi[1] + 1) for zz in range(lo[2], hi[2] + 1))
    assert want and want <= got
