"""GPU parity of the semantic keyframe flow at the two configurations bench.py TIMES it (`semantic`: 640x480 / 1 cm;
`semantic_scannet_2mm`: BASELINE configs[4]'s shape, 1296x968 / 2 mm, 1.25 M points and ~1e5 blocks per keyframe) against the
compiled reference (oracle/_ref) — VERDICT r03 Weak #1: every other semantic parity test runs at 320x240 / 2 cm or 160x120 / 4 cm,
while the per-workgroup LDS vote table, the self-cleaning global table, the pending list and the wave-level frustum cull are
exactly the size-sensitive parts.  The reference runs at 0.4-0.7 keyframes/s here: 3 keyframes each."""
import numpy as np
import pytest

import oracle
from tests.semantic_helpers import compare_keyframe_flow

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not oracle.ref_available(), reason="compiled reference not available")]


@pytest.fixture(autouse=True)
def restore_reference_statics():
    from oracle.semantic import RefSemGrid2

    yield
    for kind in (0, 2):
        RefSemGrid2(kind, 0.05).set_depth_threshold(10.0)
    for kind in (1, 3):
        g = RefSemGrid2(kind, 0.05)
        g.set_depth_threshold(5.0)
        g.set_depth_decay_rate(0.07)


@pytest.mark.parametrize("kind", [0, 1, 2, 3])  # 2, 3: the "*2" payloads (voxel_data_semantic2.h)
def test_keyframe_flow_at_the_640x480_1cm_bench_configuration(kind):
    r = compare_keyframe_flow(kind, "synthetic_640x480_5mm", 0.01, (0, 3, 6), max_blocks=1 << 17, max_points=1 << 20, full_dump=True)
    assert r["occupied_voxels"] > 100_000 and r["new_object_ids"] >= 1


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_keyframe_flow_at_the_scannet_1296x968_2mm_bench_configuration(kind):
    r = compare_keyframe_flow(kind, "scannet_1296x968_2mm", 0.002, (0, 3, 6), max_blocks=1 << 17, max_points=1296 * 968)
    assert r["blocks"] > 50_000 and r["occupied_voxels"] > 1_000_000 and r["new_object_ids"] >= 1


@pytest.mark.parametrize("kind", [0, 1])
def test_keyframe_flow_with_carving_and_host_images_at_1cm(kind):
    """The same flow with carving inside the association and host (numpy) images: the staging branch of every call."""
    compare_keyframe_flow(kind, "synthetic_640x480_5mm", 0.01, (0, 4, 8), max_blocks=1 << 16, max_points=1 << 20, do_carving=True,
                          depth_threshold=0.05, device_images=False)
