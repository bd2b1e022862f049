"""GPU: the drop-in front on the real HIP volumes, and the GPU shadow-point filter."""
import time

import numpy as np
import pytest

import oracle
from oracle import host_prep as hp
from tests import dense_helpers as dh
from tests.conftest import sort_rows

pytestmark = pytest.mark.gpu


def wait_until(cond, timeout=60.0):
    t0 = time.time()
    while time.time() - t0 < timeout:
        if cond():
            return True
        time.sleep(0.02)
    return False


@pytest.mark.parametrize("config", ["tiny_160x120_2cm", "synthetic_640x480_5mm"])
def test_shadow_filter_bit_exact(config):
    import torch
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import VoxelBlockGrid

    s = SyntheticRGBD(config)
    g = VoxelBlockGrid(0.02, 8, max_blocks=1 << 10, max_points=1 << 20)
    for i in (0, 13):
        depth = s[i][0]
        want = hp.filter_shadow_points(depth)
        got = g.filter_shadow_points(depth)
        np.testing.assert_array_equal(got, want)
        assert (want == -1).sum() > 0  # the filter really removes discontinuity pixels
        got_dev = g.filter_shadow_points(torch.from_numpy(depth).cuda()).cpu().numpy()
        np.testing.assert_array_equal(got_dev, want)
    flat = np.full((48, 64), 1.5, np.float32)  # no positive delta: median of nothing -> nothing masked
    np.testing.assert_array_equal(g.filter_shadow_points(flat), hp.filter_shadow_points(flat))


def test_shadow_filter_on_a_side_stream_equals_the_reference():
    """hv_filter_shadow_points_on_stream: the filter queued on a caller's stream with scratch of its own (the upload side of the
    keyframe pipeline) - five images in flight on two side streams while the volume integrates on its own, each bit-identical to the
    reference's depth.py."""
    import torch
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import VoxelBlockGrid

    s = SyntheticRGBD("synthetic_640x480_5mm")
    g = VoxelBlockGrid(0.02, 8, max_blocks=1 << 14, max_points=1 << 20)
    side = [torch.cuda.Stream(), torch.cuda.Stream()]
    depths = [s[i][0] for i in (0, 5, 9, 13, 21)]
    outs = []
    for k, depth in enumerate(depths):
        st = side[k % 2]
        with torch.cuda.stream(st):
            d = torch.from_numpy(depth).cuda()
            outs.append((st, g.filter_shadow_points(d, stream=st)))
        depth0, rgb0, T0 = s[k]
        g.integrate_rgbd(depth0, rgb0, *s.intrinsics, T0, max_depth=4.0)  # the volume's own stream is busy meanwhile
    for (st, out), depth in zip(outs, depths):
        st.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), hp.filter_shadow_points(depth))


def test_semantic_backlog_with_upload_side_filter_equals_keyframe_by_keyframe():
    """integrate_keyframes_on_device with a backlog (keyframe k + 1 uploaded and shadow-filtered on side streams while keyframe k
    is fused) leaves the volume of the same keyframes handed over one at a time."""
    from pyslam_amd.dense.volumetric_integrator_types import DatasetEnvironmentType, SensorType
    from pyslam_amd.dense.volumetric_integrator_voxel_semantic_grid import VolumetricIntegratorVoxelSemanticGrid
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric_semantic import set_next_object_id
    from tests.semantic_helpers import CFG, semantic_frame
    from tests.test_semantic_oracle import srt

    _params(0.02, 0.08)
    s = SyntheticRGBD(CFG, noise=True)
    cam = dh.FakeCamera(s)
    frames = [semantic_frame(s, i) for i in (0, 4, 8, 12, 16)]
    dumps = []
    for backlog in (True, False):
        integ = VolumetricIntegratorVoxelSemanticGrid.__new__(VolumetricIntegratorVoxelSemanticGrid)
        integ.init(cam, DatasetEnvironmentType.INDOOR, SensorType.RGBD, {}, dict(use_semantic_probabilistic=True))
        set_next_object_id(1)
        kfs = [(rgb, depth, T, cls_img, inst_img) for depth, rgb, T, cls_img, inst_img in frames]
        if backlog:
            integ.integrate_keyframes_on_device(kfs)
        else:
            for kf in kfs:
                integ.integrate_keyframes_on_device([kf])
        v = integ.volume.get_voxels(1, 0.0)
        dumps.append(srt((v.points, v.colors, v.class_ids, v.object_ids, v.confidences)))
    assert len(dumps[0][0]) > 1000
    for k, (a, b) in enumerate(zip(*dumps)):
        if k == 3:  # new object ids may be permuted between instances created in the same call: compare as a partition
            pairs = set(zip(a.tolist(), b.tolist()))
            assert len({p[0] for p in pairs}) == len(pairs) == len({p[1] for p in pairs})
        else:
            np.testing.assert_array_equal(a, b)


def test_shadow_filter_median_select_edge_cases():
    """The exact median comes from a three-pass radix select on the deltas' bit patterns: images built so that the two
    middle ranks sit in one histogram bin, in neighbouring bins, in bins that differ in the top digit, with even and odd
    counts, NaN / inf / zero deltas, non-default strides and one-row / one-column images - each against numpy's median."""
    from pyslam_amd.volumetric import VoxelBlockGrid

    g = VoxelBlockGrid(0.02, 8, max_blocks=1 << 10, max_points=1 << 16)
    rng = np.random.default_rng(11)
    cases = []
    for shape in ((5, 7), (1, 9), (9, 1), (33, 20), (64, 48)):
        cases.append(rng.uniform(0.3, 5.0, shape).astype(np.float32))
    a = rng.uniform(0.5, 4.0, (40, 40)).astype(np.float32)
    a[::3, ::5] = np.nan
    a[7, 9] = np.inf
    a[10:14, :] = 2.0  # zero deltas
    cases.append(a)
    b = np.full((16, 16), 1.0, np.float32)  # two populations of deltas, orders of magnitude apart: the middle ranks straddle them
    b[:, 8:] += np.float32(1e-4) * np.arange(8, dtype=np.float32)
    b[8:, :] += np.float32(3.0)
    cases.append(b)
    c = np.cumsum(np.full((30, 30), 2.0 ** -12, np.float32), axis=1).astype(np.float32) + 1.0  # many identical deltas
    cases.append(c)
    for depth in cases:
        for kw in ({}, {"delta_x": 1, "delta_y": 3}, {"delta_x": 0, "delta_y": 2}, {"delta_x": 3, "delta_y": 0}):
            if kw.get("delta_x", 2) >= depth.shape[1] or kw.get("delta_y", 2) >= depth.shape[0]:
                continue
            with np.errstate(invalid="ignore"):
                want = hp.filter_shadow_points(depth, **kw)
            got = g.filter_shadow_points(depth, **kw)
            np.testing.assert_array_equal(got, want, err_msg=f"{depth.shape} {kw}")


def _params(voxel, trunc):
    from pyslam_amd.dense.parameters import Parameters

    Parameters.kVolumetricIntegrationVoxelLength = voxel
    Parameters.kVolumetricIntegrationTSdfTrunc = trunc
    Parameters.kVolumetricIntegrationOutputTimeInterval = 0.0
    Parameters.kVolumetricIntegrationHipMaxBlocks = 1 << 13
    return Parameters


def test_voxel_grid_integrator_matches_reference_flow():
    from pyslam_amd.dense import VolumetricIntegratorType, volumetric_integrator_factory
    from pyslam_amd.dense.volumetric_integrator_types import DatasetEnvironmentType, SensorType
    from pyslam_amd.synthetic import SyntheticRGBD

    P = _params(0.02, 0.08)
    s = SyntheticRGBD("tiny_160x120_2cm")
    cam = dh.FakeCamera(s)
    integ = volumetric_integrator_factory(VolumetricIntegratorType.VOXEL_GRID, cam, DatasetEnvironmentType.INDOOR, SensorType.RGBD)
    ref = oracle.RefGrid(0.02, 8) if oracle.ref_available() else oracle.PortGrid(0.02, 8)
    try:
        assert wait_until(integ.is_ready), "worker did not start"
        last = None
        for i in range(3):
            kf = dh.FakeKeyFrame(i, s, cam)
            integ.add_keyframe(kf, kf.img, None, kf.depth_img)
            out = []
            assert wait_until(lambda: (out.append(integ.pop_output(timeout=0.2)) or True) and out[-1] is not None)
            last = out[-1]
            depth, rgb, T = s[i]
            pts, cols, _ = hp.frame_to_world_f32(hp.filter_shadow_points(depth), rgb, *s.intrinsics, T, 4.0)
            ref.integrate(pts, cols)
        pa, ca = sort_rows(last.point_cloud.points, last.point_cloud.colors)
        pb, cb = sort_rows(*ref.get_voxels(P.kVolumetricIntegrationVoxelGridMinCount))
        np.testing.assert_array_equal(pa, pb)
        np.testing.assert_array_equal(ca, cb)
        assert last.point_cloud.points.dtype == np.float32
    finally:
        integ.quit()


@pytest.mark.parametrize("out_f32", [False, True])
def test_tsdf_integrator_end_to_end(tmp_path, out_f32, monkeypatch):
    """out_f32: kVolumetricIntegrationTsdfOutputInDenseMappingDtype - the output ticks carry float32 vertices / colours (the float64
    ones rounded once on the device); the saved dense_map.ply still comes from float64."""
    from pyslam_amd.dense import VolumetricIntegrationTaskType, VolumetricIntegratorType, volumetric_integrator_factory
    from pyslam_amd.dense.ply_io import read_ply
    from pyslam_amd.dense.volumetric_integrator_types import DatasetEnvironmentType, SensorType
    from pyslam_amd.synthetic import SyntheticRGBD

    monkeypatch.setattr(_params(0.02, 0.08), "kVolumetricIntegrationTsdfOutputInDenseMappingDtype", out_f32)
    s = SyntheticRGBD("tiny_160x120_2cm")
    cam = dh.FakeCamera(s)
    integ = volumetric_integrator_factory(VolumetricIntegratorType.TSDF, cam, DatasetEnvironmentType.INDOOR, SensorType.RGBD)
    cpu = oracle.PortTsdf(0.02, 0.08)
    K = np.array(s.intrinsics)
    try:
        assert wait_until(integ.is_ready), "worker did not start"
        last = None
        for i in range(3):
            kf = dh.FakeKeyFrame(i, s, cam)
            integ.add_keyframe(kf, kf.img, None, kf.depth_img)
            out = []
            assert wait_until(lambda: (out.append(integ.pop_output(timeout=0.2)) or True) and out[-1] is not None)
            last = out[-1]
            depth, rgb, T = s[i]
            cpu.integrate(depth, rgb, K, T, 1.0, 4.0)
        v, t, c = cpu.extract_triangle_mesh()
        assert last.task_type == VolumetricIntegrationTaskType.INTEGRATE and last.id == 2
        assert last.mesh.vertices.shape == v.shape and last.mesh.triangles.shape == t.shape
        assert last.mesh.vertices.dtype == last.mesh.vertex_colors.dtype == (np.float32 if out_f32 else np.float64)
        if out_f32:
            np.testing.assert_allclose(np.sort(last.mesh.vertices, axis=0), np.sort(v, axis=0).astype(np.float32), atol=5e-7, rtol=0)
        else:
            np.testing.assert_allclose(np.sort(last.mesh.vertices, axis=0), np.sort(v, axis=0), atol=1e-9)
        integ.save(str(tmp_path))
        pts, cols, faces = read_ply(str(tmp_path / "dense_map.ply"))
        assert pts.shape == v.shape and faces.shape == t.shape
    finally:
        integ.quit()


@pytest.mark.parametrize("use_instance_ids", [True, False])
@pytest.mark.parametrize("probabilistic", [False, True])
def test_semantic_integrator_matches_reference_flow(probabilistic, use_instance_ids):
    """VolumetricIntegratorVoxelSemanticGrid on the real GPU volume vs the same integrator class driving the
    compiled reference through the stand-in volume: identical labelled voxels after 3 keyframes.  Carving is on: with
    instance ids it runs inside the association, without them as volume.carve() on the device-resident depth."""
    if not oracle.ref_available():
        pytest.skip("compiled reference not available")
    from pyslam_amd.dense.parameters import Parameters
    from pyslam_amd.dense.volumetric_integrator_types import DatasetEnvironmentType, SensorType
    from pyslam_amd.dense.volumetric_integrator_voxel_semantic_grid import VolumetricIntegratorVoxelSemanticGrid
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric_semantic import set_next_object_id
    from oracle.semantic import RefSemGrid2
    from tests.semantic_helpers import CFG, relabel, semantic_frame
    from tests.test_semantic_oracle import srt

    P = _params(0.02, 0.08)
    old = (P.kVolumetricIntegrationVoxelGridMinCount, P.kVolumetricIntegrationVoxelGridMinConfidence, P.kVolumetricIntegrationVoxelGridUseCarving,
           P.kVolumetricSemanticIntegrationUseInstanceIds)
    P.kVolumetricIntegrationVoxelGridMinCount, P.kVolumetricIntegrationVoxelGridMinConfidence = 1, 0.0
    P.kVolumetricIntegrationVoxelGridUseCarving = True
    P.kVolumetricSemanticIntegrationUseInstanceIds = use_instance_ids
    try:
        s = SyntheticRGBD(CFG, noise=True)
        cam = dh.FakeCamera(s)
        kw = dict(use_semantic_probabilistic=probabilistic)
        gpu = VolumetricIntegratorVoxelSemanticGrid.__new__(VolumetricIntegratorVoxelSemanticGrid)
        gpu.init(cam, DatasetEnvironmentType.INDOOR, SensorType.RGBD, {}, dict(kw))
        ref = VolumetricIntegratorVoxelSemanticGrid.__new__(VolumetricIntegratorVoxelSemanticGrid)
        ref.init(cam, DatasetEnvironmentType.INDOOR, SensorType.RGBD, {}, dict(kw, volume_factory=dh.oracle_semantic_factory))
        set_next_object_id(1)
        ref.volume.grid.set_next_object_id(1)
        for i in (0, 6, 12):
            depth, rgb, T, cls_img, inst_img = semantic_frame(s, i)
            for integ in (gpu, ref):
                integ.integrate_keyframe(rgb, depth, T, cls_img, inst_img)
        a = gpu.volume.get_voxels(1, 0.0)
        b = ref.volume.get_voxels(1, 0.0)
        ga = srt((a.points, a.colors, a.class_ids, a.object_ids, a.confidences))
        gb = srt((b.points, b.colors, b.class_ids, b.object_ids, b.confidences))
        assert len(ga[0]) == len(gb[0]) > 1000
        np.testing.assert_array_equal(ga[0], gb[0])
        np.testing.assert_array_equal(ga[1], gb[1])
        np.testing.assert_array_equal(ga[2], gb[2])
        # new object ids may be permuted between instances created in the same call: compare as a partition
        pairs = set(zip(ga[3].tolist(), gb[3].tolist()))
        assert len({p[0] for p in pairs}) == len(pairs) == len({p[1] for p in pairs})
        np.testing.assert_allclose(ga[4], gb[4], rtol=0, atol=0 if not probabilistic else 2e-6)
        out = gpu.make_output("INTEGRATE")
        if use_instance_ids:
            assert out.objects is not None and out.objects.num_objects >= 1
        else:
            assert out.point_cloud is not None and len(out.point_cloud.points) > 1000
    finally:
        (P.kVolumetricIntegrationVoxelGridMinCount, P.kVolumetricIntegrationVoxelGridMinConfidence, P.kVolumetricIntegrationVoxelGridUseCarving,
         P.kVolumetricSemanticIntegrationUseInstanceIds) = old
        RefSemGrid2(0, 0.05).set_depth_threshold(10.0)
        g = RefSemGrid2(1, 0.05)
        g.set_depth_threshold(5.0)
        g.set_depth_decay_rate(0.07)


def test_semantic_integrator_worker_process():
    from pyslam_amd.dense import VolumetricIntegrationObjectList, VolumetricIntegratorType, volumetric_integrator_factory
    from pyslam_amd.dense.volumetric_integrator_types import DatasetEnvironmentType, SensorType
    from pyslam_amd.synthetic import SyntheticRGBD
    from tests.semantic_helpers import CFG

    _params(0.02, 0.08)
    s = SyntheticRGBD(CFG, noise=False)
    cam = dh.FakeCamera(s)
    integ = volumetric_integrator_factory(VolumetricIntegratorType.VOXEL_SEMANTIC_PROBABILISTIC_GRID, cam, DatasetEnvironmentType.INDOOR,
                                          SensorType.RGBD)
    try:
        assert wait_until(integ.is_ready), "worker did not start"
        last = None
        for i in (0, 6, 12):
            kf = dh.FakeKeyFrame(i, s, cam, semantic=True)
            integ.add_keyframe(kf, kf.img, None, kf.depth_img)
            out = []
            assert wait_until(lambda: (out.append(integ.pop_output(timeout=0.2)) or True) and out[-1] is not None)
            last = out[-1]
        assert last.id == 12 and isinstance(last.objects, VolumetricIntegrationObjectList)
    finally:
        integ.quit()


def test_depth_estimator_handoff_stays_on_device():
    """N3: a stereo keyframe without depth -> torch stereo network -> depth as a CUDA tensor -> shadow filter -> TSDF
    fusion, no host round trip; identical volume to the same depth taken through numpy."""
    import torch
    from pyslam_amd.dense.parameters import Parameters
    from pyslam_amd.dense.volumetric_integrator_base import VolumetricIntegrationKeyframeData
    from pyslam_amd.dense.volumetric_integrator_tsdf import VolumetricIntegratorTsdf
    from pyslam_amd.dense.volumetric_integrator_types import DatasetEnvironmentType, SensorType
    from pyslam_amd.depth_estimation import DepthEstimatorStereoTorch, make_stub_stereo_net
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import RGBDImage

    _params(0.02, 0.08)
    old = Parameters.kVolumetricIntegrationUseDepthEstimator
    Parameters.kVolumetricIntegrationUseDepthEstimator = True
    try:
        s = SyntheticRGBD("tiny_160x120_2cm")
        cam = dh.FakeCamera(s)
        cam.bf = 40.0
        net = make_stub_stereo_net()
        made = []

        def factory(camera, keep=True):
            made.append(DepthEstimatorStereoTorch(net, camera, "cuda", keep_on_device=keep))
            return made[-1]

        dev = VolumetricIntegratorTsdf.__new__(VolumetricIntegratorTsdf)
        dev.init(cam, DatasetEnvironmentType.INDOOR, SensorType.STEREO, {}, dict(depth_estimator_factory=factory))
        host = VolumetricIntegratorTsdf.__new__(VolumetricIntegratorTsdf)
        host.init(cam, DatasetEnvironmentType.INDOOR, SensorType.STEREO, {}, dict(depth_estimator_factory=lambda c: factory(c, False)))
        for i in range(3):
            kf = dh.FakeKeyFrame(i, s, cam)
            kf.depth_img = None
            kf.img_right = np.ascontiguousarray(np.roll(kf.img, 3, axis=1))
            for integ in (dev, host):
                kd = VolumetricIntegrationKeyframeData(kf, kf.img, kf.img_right, None)
                color, depth, _, _, _ = integ.estimate_depth_if_needed_and_rectify(kd)
                assert (hasattr(depth, "is_cuda") and depth.is_cuda) == (integ is dev)
                if integ is dev:
                    assert isinstance(color, torch.Tensor) and color.is_cuda
                rgbd = RGBDImage.create_from_color_and_depth(color, depth, depth_scale=1.0, depth_trunc=4.0, convert_rgb_to_intensity=False)
                integ.volume.integrate(rgbd, integ.o3d_camera, kd.pose)
                assert kd.id in integ.img_id_to_depth  # cached like the reference (base.py:1050-1052)
        for a, b in zip(dev.volume.dump(), host.volume.dump()):
            np.testing.assert_array_equal(a, b)
        assert dev.volume.num_blocks() > 10
    finally:
        Parameters.kVolumetricIntegrationUseDepthEstimator = old


def test_dense_reconstruction_cli_from_saved_state(tmp_path):
    """N4: map.json written in pySLAM's layout -> headless main_map_dense_reconstruction -> dense_map.ply (TSDF mesh)."""
    from pyslam_amd.dense.ply_io import read_ply
    from pyslam_amd.dense.volumetric_integrator_types import DatasetEnvironmentType, SensorType
    from pyslam_amd.io import CameraRecord, KeyFrameRecord, save_system_state
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.tools import dense_reconstruction as cli

    _params(0.02, 0.08)
    s = SyntheticRGBD("tiny_160x120_2cm")
    cam = CameraRecord(s.width, s.height, *s.intrinsics)
    kfs = []
    for i in range(4):
        depth, rgb, T = s[i]
        kfs.append(KeyFrameRecord(i, T, cam, np.ascontiguousarray(rgb[..., ::-1]), depth, timestamp=i / 30.0))
    state = tmp_path / "state"
    save_system_state(str(state), kfs, SensorType.RGBD, DatasetEnvironmentType.INDOOR)
    out = tmp_path / "out"
    n = cli.main(["-p", str(state), "-o", str(out), "--type", "TSDF", "--voxel-length", "0.02"])
    assert n == 4
    pts, cols, faces = read_ply(str(out / "dense_map.ply"))
    cpu = oracle.PortTsdf(0.02, 0.08)
    for i in range(4):
        depth, rgb, T = s[i]
        cpu.integrate(depth, rgb, np.array(s.intrinsics), T, 1.0, 4.0)
    v, t, c = cpu.extract_triangle_mesh()
    assert pts.shape == v.shape and faces.shape == t.shape
    np.testing.assert_allclose(np.sort(pts, axis=0), np.sort(v, axis=0), atol=1e-6)


def test_prep_kernels_match_reference_depth_fixture():
    """The GPU prep kernels against tests/golden/prep_depth.npz, i.e. against outputs of the reference's own
    pyslam/utilities/depth.py (tools/make_golden_prep.py): the shadow-point filter bit for bit (MAD and fixed threshold), the
    fused depth2pointcloud + integrate through the voxel rows it produces."""
    import os

    import oracle
    from pyslam_amd.volumetric import VoxelBlockGrid

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "prep_depth.npz"))
    g = VoxelBlockGrid(0.02, 8, max_blocks=1 << 12, max_points=1 << 16)
    np.testing.assert_array_equal(g.filter_shadow_points(z["depth"]), z["shadow_mad"])
    # reference points (camera frame; identity pose) -> oracle grid; fused GPU path from the raw depth -> same grid
    fx, fy, cx, cy = z["intr"]
    g.integrate_rgbd(z["depth"], z["rgb"], fx, fy, cx, cy, np.eye(4), max_depth=float(z["max_depth"]), min_depth=float(z["min_depth"]))
    ref = oracle.RefGrid(0.02, 8) if oracle.ref_available() else oracle.PortGrid(0.02, 8)
    ref.integrate(z["points"].astype(np.float32), z["colors"].astype(np.float32))
    for a, b in zip(g.dump(), ref.dump()):
        np.testing.assert_array_equal(a, b)
