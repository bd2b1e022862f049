"""CPU: the data formats either side of the dense path — pySLAM's map.json keyframe / camera /
image encoding (pinned to a fixture produced by the reference's own NumpyB64Json, tools/make_golden_io.py) and the
TUM / Replica / ScanNet / EuRoC directory layouts, written synthetically and read back."""
import json
import os

import numpy as np
import pytest

from pyslam_amd.dense.volumetric_integrator_types import DatasetEnvironmentType, SensorType
from pyslam_amd.io import (CameraRecord, EurocDataset, KeyFrameRecord, ReplicaDataset, ScannetDataset, TumDataset,
                           dataset_factory, load_system_state, numpy_from_json, numpy_to_json, save_system_state)
from pyslam_amd.io.datasets import camera_from_settings, inv_T
from pyslam_amd.io.images import imread_color, imread_unchanged, imwrite
from pyslam_amd.synthetic import SyntheticRGBD
from tools.make_golden_io import arrays

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "numpy_b64.json")


def test_numpy_b64_matches_the_reference_encoding():
    with open(GOLD) as f:
        gold = json.load(f)
    for name, arr in arrays().items():
        ours = numpy_to_json(arr)
        ref = gold[name]
        assert ours["type"] == ref["type"] == "npB64"
        assert (ours["dtype"], list(ours["shape"]), ours["order"], ours["data"]) == (ref["dtype"], list(ref["shape"]), ref["order"], ref["data"])
        np.testing.assert_array_equal(numpy_from_json(ref), arr)
        assert numpy_from_json(ref).dtype == arr.dtype


def make_stream():
    s = SyntheticRGBD("tiny_160x120_2cm", noise=False)
    cam = CameraRecord(s.width, s.height, *s.intrinsics, D=[0, 0, 0, 0, 0], bf=40.0, depth_factor=1.0 / 5000.0)
    return s, cam


def test_system_state_roundtrip_and_keyframe_protocol(tmp_path):
    s, cam = make_stream()
    cam.depth_factor = 1.0
    kfs = []
    for i in range(3):
        depth, rgb, T = s[i]
        kfs.append(KeyFrameRecord(i, T, cam, np.ascontiguousarray(rgb[..., ::-1]), depth, None, np.full(depth.shape, i, np.int32), None,
                                  timestamp=i / 30.0, lba_count=2))
    kfs.append(KeyFrameRecord(3, np.eye(4), cam, kfs[0].img, kfs[0].depth_img, is_bad=True))
    save_system_state(str(tmp_path), kfs, SensorType.RGBD, DatasetEnvironmentType.INDOOR)
    raw = json.load(open(tmp_path / "map.json"))
    assert raw["sensor_type"] == "SensorType.RGBD" and raw["environment_type"] == "DatasetEnvironmentType.INDOOR"
    assert set(raw["map"]) >= {"frames", "keyframes", "points", "viewer_scale"}
    st = load_system_state(str(tmp_path))
    assert st.sensor_type is SensorType.RGBD and st.environment_type is DatasetEnvironmentType.INDOOR
    assert st.map.num_keyframes() == 3  # the bad keyframe is not saved (map.py:956)
    for a, b in zip(st.map.get_keyframes(), kfs):
        np.testing.assert_array_equal(a.pose(), b.pose())
        np.testing.assert_array_equal(a.img, b.img)
        np.testing.assert_array_equal(a.depth_img, b.depth_img)
        np.testing.assert_array_equal(a.semantic_img, b.semantic_img)
        assert a.depth_img.dtype == np.float32 and a.lba_count == 2 and not a.is_bad() and a.is_semantics_available()
        assert (a.camera.fx, a.camera.width, a.camera.bf) == (cam.fx, cam.width, cam.bf)
    with pytest.raises(FileNotFoundError):
        load_system_state(str(tmp_path / "nope"))


def write_tum(root, name, s, n=4):
    base = os.path.join(root, name)
    os.makedirs(os.path.join(base, "rgb"))
    os.makedirs(os.path.join(base, "depth"))
    assoc, gt = [], ["# ground truth trajectory", "# file: synthetic", "# timestamp tx ty tz qx qy qz qw"]
    from scipy.spatial.transform import Rotation

    for i in range(n):
        depth, rgb, T = s[i]
        t = 1000.0 + i / 30.0
        imwrite(os.path.join(base, "rgb", f"{t:.6f}.png"), rgb[..., ::-1])
        imwrite(os.path.join(base, "depth", f"{t:.6f}.png"), np.clip(np.rint(depth * 5000.0), 0, 65535).astype(np.uint16))
        assoc.append(f"{t:.6f} rgb/{t:.6f}.png {t:.6f} depth/{t:.6f}.png")
        Twc = inv_T(T)
        q = Rotation.from_matrix(Twc[:3, :3]).as_quat()  # x y z w
        gt.append(f"{t + 0.001:.6f} {Twc[0, 3]:.9f} {Twc[1, 3]:.9f} {Twc[2, 3]:.9f} {q[0]:.9f} {q[1]:.9f} {q[2]:.9f} {q[3]:.9f}")
    open(os.path.join(base, "associations.txt"), "w").write("\n".join(assoc) + "\n")
    open(os.path.join(base, "groundtruth.txt"), "w").write("\n".join(gt) + "\n")


def test_tum_layout(tmp_path):
    s, cam = make_stream()
    write_tum(str(tmp_path), "seq", s)
    ds = dataset_factory("tum", str(tmp_path), "seq", cam)
    assert isinstance(ds, TumDataset) and ds.num_frames == 4
    kfs = list(ds.keyframes())
    assert len(kfs) == 4
    for i, kf in enumerate(kfs):
        depth, rgb, T = s[i]
        np.testing.assert_array_equal(kf.img, rgb[..., ::-1])  # BGR like cv2.imread
        assert kf.depth_img.dtype == np.float32
        assert np.abs(kf.depth_img - depth).max() <= 0.5 / 5000.0 + 1e-6  # uint16 quantisation at DepthMapFactor 5000
        np.testing.assert_allclose(kf.pose(), T, atol=1e-6)
        assert kf.timestamp == pytest.approx(1000.0 + i / 30.0)
    raw = ds.getDepth(0)
    assert raw.dtype == np.uint16  # IMREAD_UNCHANGED keeps the bit depth (dataset.py:626-631)
    ds.max_pose_dt = 1e-5
    assert ds.keyframe(0) is None  # no ground-truth sample close enough -> frame skipped


def test_replica_scannet_euroc_layouts(tmp_path):
    s, cam = make_stream()
    # Replica: results/frame%06d.jpg + depth%06d.png + traj.txt (T_wc rows)
    base = tmp_path / "replica" / "office0" / "results"
    os.makedirs(base)
    rows = []
    for i in range(2):
        depth, rgb, T = s[i]
        imwrite(str(base / f"frame{i:06d}.jpg"), rgb[..., ::-1])
        imwrite(str(base / f"depth{i:06d}.png"), np.rint(depth * 5000).astype(np.uint16))
        rows.append(" ".join(f"{x:.12e}" for x in inv_T(T).ravel()))
    open(tmp_path / "replica" / "office0" / "traj.txt", "w").write("\n".join(rows) + "\n")
    ds = ReplicaDataset(str(tmp_path / "replica"), "office0", cam)
    kf = ds.keyframe(1)
    np.testing.assert_allclose(kf.pose(), s[1][2], atol=1e-9)
    assert kf.img.shape == (s.height, s.width, 3) and kf.depth_img.dtype == np.float32
    assert np.abs(kf.img.astype(int) - s[1][1][..., ::-1].astype(int)).mean() < 6  # JPEG
    # ScanNet: scans/<name>/{color,depth,pose,label-filt,instance-filt}
    base = tmp_path / "scannet" / "scans" / "scene0000_00"
    for sub in ("color", "depth", "pose", "label-filt", "instance-filt"):
        os.makedirs(base / sub)
    cam_mm = CameraRecord(cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy, depth_factor=1.0 / 1000.0)
    for i in (0, 10):
        depth, rgb, T = s[i]
        imwrite(str(base / "color" / f"{i}.jpg"), np.repeat(np.repeat(rgb[..., ::-1], 2, 0), 2, 1))  # colour at 2x the depth size
        imwrite(str(base / "depth" / f"{i}.png"), np.rint(depth * 1000).astype(np.uint16))
        np.savetxt(base / "pose" / f"{i}.txt", inv_T(T))
        imwrite(str(base / "label-filt" / f"{i}.png"), s.labels(i).astype(np.uint16))
        imwrite(str(base / "instance-filt" / f"{i}.png"), (s.labels(i) % 7).astype(np.uint8))
    ds = ScannetDataset(str(tmp_path / "scannet"), "scene0000_00", cam_mm)
    assert ds.ids == [0, 10]
    kf = ds.keyframe(1)
    assert kf.img.shape == (s.height, s.width, 3)  # resized to the camera (= depth) resolution
    np.testing.assert_allclose(kf.pose(), s[10][2], atol=1e-9)
    np.testing.assert_array_equal(kf.semantic_img, s.labels(10).astype(np.int32))
    np.testing.assert_array_equal(kf.semantic_instances_img, (s.labels(10) % 7).astype(np.int32))
    assert np.abs(kf.depth_img - s[10][0]).max() <= 0.5e-3 + 1e-6
    # EuRoC: stereo pair, no depth, body poses + T_BS
    base = tmp_path / "euroc" / "MH01" / "mav0"
    for cam_dir in ("cam0", "cam1"):
        os.makedirs(base / cam_dir / "data")
    os.makedirs(base / "state_groundtruth_estimate0")
    lines, gts = ["#timestamp [ns],filename"], ["#timestamp, p_RS_R_x [m], p_RS_R_y [m], p_RS_R_z [m], q_RS_w [], q_RS_x [], q_RS_y [], q_RS_z []"]
    from scipy.spatial.transform import Rotation

    for i in range(2):
        depth, rgb, T = s[i]
        ts = 1403636579763555584 + i * 50000000
        gray = rgb.mean(axis=2).astype(np.uint8)
        imwrite(str(base / "cam0" / "data" / f"{ts}.png"), gray)
        imwrite(str(base / "cam1" / "data" / f"{ts}.png"), np.roll(gray, 2, axis=1))
        lines.append(f"{ts},{ts}.png")
        Twc = inv_T(T)
        q = Rotation.from_matrix(Twc[:3, :3]).as_quat()
        gts.append(f"{ts},{Twc[0, 3]:.9f},{Twc[1, 3]:.9f},{Twc[2, 3]:.9f},{q[3]:.9f},{q[0]:.9f},{q[1]:.9f},{q[2]:.9f}")
    for cam_dir in ("cam0", "cam1"):
        open(base / cam_dir / "data.csv", "w").write("\n".join(lines) + "\n")
    open(base / "state_groundtruth_estimate0" / "data.csv", "w").write("\n".join(gts) + "\n")
    ds = EurocDataset(str(tmp_path / "euroc"), "MH01", cam)
    kf = ds.keyframe(1)
    assert kf.depth_img is None and kf.img_right is not None and kf.img.shape == (s.height, s.width, 3)
    np.testing.assert_allclose(kf.pose(), s[1][2], atol=1e-6)
    assert ds.sensor_type is SensorType.STEREO


def test_camera_from_reference_style_settings(tmp_path):
    p = tmp_path / "TUM1.yaml"
    p.write_text("%YAML:1.0\nCamera.width: 640\nCamera.height: 480\nCamera.fx: 517.306408\nCamera.fy: 516.469215\n"
                 "Camera.cx: 318.643040\nCamera.cy: 255.313989\nCamera.k1: 0.262383\nCamera.k2: -0.953104\nCamera.p1: -0.005358\n"
                 "Camera.p2: 0.002628\nCamera.k3: 1.163314\nCamera.fps: 30\nCamera.bf: 40.0\nDepthMapFactor: 5000.0\n")
    cam = camera_from_settings(str(p))
    assert (cam.width, cam.height) == (640, 480) and cam.is_distorted
    assert cam.depth_factor == pytest.approx(1 / 5000.0) and cam.D[4] == pytest.approx(1.163314)


def test_image_helpers(tmp_path):
    a = (np.arange(12 * 10).reshape(12, 10) * 500).astype(np.uint16)
    imwrite(str(tmp_path / "d.png"), a)
    np.testing.assert_array_equal(imread_unchanged(str(tmp_path / "d.png")), a)
    c = np.random.default_rng(0).integers(0, 255, (6, 7, 3)).astype(np.uint8)
    imwrite(str(tmp_path / "c.png"), c)
    np.testing.assert_array_equal(imread_color(str(tmp_path / "c.png")), c)
    assert imread_color(str(tmp_path / "missing.png")) is None
