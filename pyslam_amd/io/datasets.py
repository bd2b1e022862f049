"""Dataset readers for the sequences BASELINE.json names (reference: pyslam/io/dataset.py:576-660 TUM/ICL-NUIM,
:662-798 ScanNet, :800-988 EuRoC, :990-1048 Replica; ground truth: pyslam/io/ground_truth.py).  Same directory
layouts, file naming, depth scaling and pose conventions; every reader yields KeyFrameRecord objects with
T_cw poses from the dataset's ground truth, ready for VolumetricIntegratorBase.add_keyframe()."""
import glob
import os

import numpy as np

from ..dense.volumetric_integrator_types import DatasetEnvironmentType, SensorType
from .images import imread_color, imread_unchanged
from .system_state import CameraRecord, KeyFrameRecord


def quat_to_R(qx, qy, qz, qw):
    n = np.sqrt(qx * qx + qy * qy + qz * qz + qw * qw)
    x, y, z, w = qx / n, qy / n, qz / n, qw / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def inv_T(T):
    out = np.eye(4)
    out[:3, :3] = T[:3, :3].T
    out[:3, 3] = -T[:3, :3].T @ T[:3, 3]
    return out


class Dataset:
    environment_type = DatasetEnvironmentType.INDOOR
    fps = 30

    def __init__(self, path, name, camera, sensor_type=SensorType.RGBD, start_frame_id=0):
        self.path, self.name, self.camera = path, name, camera
        self.sensor_type = sensor_type
        self.start_frame_id = start_frame_id
        self.num_frames = 0

    # -- to implement: getImage / getImageRight / getDepth (raw file values) / getTimestamp / getPoseTwc ---------
    def getImageRight(self, frame_id):
        return None

    def getSemantic(self, frame_id):
        return None, None

    def depth_in_metres(self, raw):
        """Frame's conversion (pyslam/slam/frame.py:429-430): depth_img * camera.depth_factor, float32."""
        if raw is None:
            return None
        d = raw.astype(np.float32)
        f = np.float32(self.camera.depth_factor)
        return d * f if f != 1.0 else d

    def keyframe(self, frame_id):
        """-> KeyFrameRecord (BGR image, float32 metric depth, T_cw) or None when the frame has no pose / image."""
        img = self.getImage(frame_id)
        Twc = self.getPoseTwc(frame_id)
        if img is None or Twc is None or not np.isfinite(Twc).all():
            return None
        sem, inst = self.getSemantic(frame_id)
        return KeyFrameRecord(frame_id, inv_T(Twc), self.camera, img, self.depth_in_metres(self.getDepth(frame_id)),
                              self.getImageRight(frame_id), sem, inst, self.getTimestamp(frame_id), img_id=frame_id)

    def keyframes(self, step=1, max_frames=None):
        n = 0
        for i in range(self.start_frame_id, self.num_frames, step):
            kf = self.keyframe(i)
            if kf is None:
                continue
            yield kf
            n += 1
            if max_frames is not None and n >= max_frames:
                return


class TumDataset(Dataset):
    """<path>/<name>/{rgb/*.png, depth/*.png (uint16, DepthMapFactor 5000), associations.txt, groundtruth.txt}
    (dataset.py:576-643, ground_truth.py TumGroundTruth: rows `timestamp tx ty tz qx qy qz qw` = T_wc)."""

    def __init__(self, path, name, camera, associations="associations.txt", sensor_type=SensorType.RGBD, start_frame_id=0,
                 max_pose_dt=0.02):
        super().__init__(path, name, camera, sensor_type, start_frame_id)
        self.base_path = os.path.join(path, name)
        with open(os.path.join(self.base_path, associations)) as f:
            self.associations_data = [ln.strip().split() for ln in f if ln.strip() and not ln.startswith("#")]
        self.num_frames = len(self.associations_data)
        gt = []
        gt_file = os.path.join(self.base_path, "groundtruth.txt")
        if os.path.exists(gt_file):
            with open(gt_file) as f:
                for ln in f:
                    if ln.startswith("#") or not ln.strip():
                        continue
                    gt.append([float(x) for x in ln.split()[:8]])
        self.gt = np.array(gt, dtype=np.float64).reshape(-1, 8)
        self.max_pose_dt = max_pose_dt

    def getTimestamp(self, frame_id):
        return float(self.associations_data[frame_id][0])

    def getImage(self, frame_id):
        return imread_color(os.path.join(self.base_path, self.associations_data[frame_id][1])) if frame_id < self.num_frames else None

    def getDepth(self, frame_id):
        if self.sensor_type == SensorType.MONOCULAR or frame_id >= self.num_frames:
            return None
        return imread_unchanged(os.path.join(self.base_path, self.associations_data[frame_id][3]))

    def getPoseTwc(self, frame_id):
        if len(self.gt) == 0:
            return None
        t = self.getTimestamp(frame_id)
        k = int(np.argmin(np.abs(self.gt[:, 0] - t)))  # nearest ground-truth sample, as the reference's association
        if abs(self.gt[k, 0] - t) > self.max_pose_dt:
            return None
        T = np.eye(4)
        T[:3, :3] = quat_to_R(*self.gt[k, 4:8])
        T[:3, 3] = self.gt[k, 1:4]
        return T


class IclNuimDataset(TumDataset):
    """Same layout as TUM (dataset.py:646-659)."""


class ReplicaDataset(Dataset):
    """<path>/<name>/results/{frame%06d.jpg, depth%06d.png (uint16, DepthMapFactor 6553.5)} + traj.txt with one
    row-major T_wc per line (dataset.py:990-1048, ground_truth.py ReplicaGroundTruth)."""

    fps = 25

    def __init__(self, path, name, camera, sensor_type=SensorType.RGBD, start_frame_id=0):
        super().__init__(path, name, camera, sensor_type, start_frame_id)
        self.base_path = os.path.join(path, name)
        self.color_paths = sorted(glob.glob(os.path.join(self.base_path, "results", "frame*.jpg")))
        self.num_frames = len(self.color_paths)
        traj = os.path.join(self.base_path, "traj.txt")
        self.poses = np.loadtxt(traj).reshape(-1, 4, 4) if os.path.exists(traj) else np.zeros((0, 4, 4))

    def getTimestamp(self, frame_id):
        return frame_id / self.fps

    def getImage(self, frame_id):
        return imread_color(os.path.join(self.base_path, "results", f"frame{frame_id:06d}.jpg"))

    def getDepth(self, frame_id):
        return imread_unchanged(os.path.join(self.base_path, "results", f"depth{frame_id:06d}.png"))

    def getPoseTwc(self, frame_id):
        return self.poses[frame_id] if frame_id < len(self.poses) else None


class ScannetDataset(Dataset):
    """<path>/scans/<name>/{color/<i>.jpg, depth/<i>.png (uint16 mm), pose/<i>.txt (T_wc), label-filt/<i>.png,
    instance-filt/<i>.png} (dataset.py:662-798, ground_truth.py ScannetGroundTruth).  Colour images are resized to the
    camera size (the depth resolution) like the reference does; labels are read as class / instance id images."""

    def __init__(self, path, name, camera, sensor_type=SensorType.RGBD, start_frame_id=0, with_labels=True):
        super().__init__(path, name, camera, sensor_type, start_frame_id)
        self.base_path = os.path.join(path, "scans", name)
        ids = [int(os.path.splitext(os.path.basename(p))[0]) for p in glob.glob(os.path.join(self.base_path, "color", "*.jpg"))]
        self.ids = sorted(ids)
        self.num_frames = len(self.ids)
        self.with_labels = with_labels

    def _id(self, frame_id):
        return self.ids[frame_id]

    def getTimestamp(self, frame_id):
        return self._id(frame_id) / self.fps

    def getImage(self, frame_id):
        img = imread_color(os.path.join(self.base_path, "color", f"{self._id(frame_id)}.jpg"))
        if img is not None and (img.shape[1], img.shape[0]) != (self.camera.width, self.camera.height):
            from PIL import Image

            img = np.ascontiguousarray(np.asarray(Image.fromarray(img).resize((self.camera.width, self.camera.height), Image.BILINEAR)))
        return img

    def getDepth(self, frame_id):
        return imread_unchanged(os.path.join(self.base_path, "depth", f"{self._id(frame_id)}.png"))

    def getPoseTwc(self, frame_id):
        f = os.path.join(self.base_path, "pose", f"{self._id(frame_id)}.txt")
        return np.loadtxt(f).reshape(4, 4) if os.path.exists(f) else None

    def getSemantic(self, frame_id):
        if not self.with_labels:
            return None, None
        out = []
        for sub in ("label-filt", "instance-filt"):
            a = imread_unchanged(os.path.join(self.base_path, sub, f"{self._id(frame_id)}.png"))
            if a is not None and a.shape[:2] != (self.camera.height, self.camera.width):
                from PIL import Image

                a = np.asarray(Image.fromarray(a).resize((self.camera.width, self.camera.height), Image.NEAREST))
            out.append(None if a is None else np.ascontiguousarray(a, dtype=np.int32))
        return out[0], out[1]


class EurocDataset(Dataset):
    """<path>/<name>/mav0/{cam0,cam1}/{data/<ns>.png, data.csv} + state_groundtruth_estimate0/data.csv
    (dataset.py:800-988).  Stereo: keyframes carry img / img_right and no depth (the integrator's depth estimator
    supplies it).  Ground-truth rows are body poses `t[ns] p(3) q(w,x,y,z) ...`; T_BS of cam0 maps
    them to the camera when given."""

    fps = 20

    def __init__(self, path, name, camera, sensor_type=SensorType.STEREO, start_frame_id=0, T_BS=None, max_pose_dt=0.01):
        super().__init__(path, name, camera, sensor_type, start_frame_id)
        self.base_path = os.path.join(path, name, "mav0")
        rows = []
        with open(os.path.join(self.base_path, "cam0", "data.csv")) as f:
            for ln in f:
                if ln.startswith("#") or not ln.strip():
                    continue
                t, fn = ln.strip().split(",")[:2]
                rows.append((int(t), fn.strip()))
        self.rows = rows
        self.num_frames = len(rows)
        self.T_BS = np.eye(4) if T_BS is None else np.asarray(T_BS, dtype=np.float64).reshape(4, 4)
        gt_file = os.path.join(self.base_path, "state_groundtruth_estimate0", "data.csv")
        gt = []
        if os.path.exists(gt_file):
            with open(gt_file) as f:
                for ln in f:
                    if ln.startswith("#") or not ln.strip():
                        continue
                    gt.append([float(x) for x in ln.strip().split(",")[:8]])
        self.gt = np.array(gt, dtype=np.float64).reshape(-1, 8)
        self.max_pose_dt = max_pose_dt

    def getTimestamp(self, frame_id):
        return self.rows[frame_id][0] * 1e-9

    def _gray_as_bgr(self, cam, frame_id):
        img = imread_color(os.path.join(self.base_path, cam, "data", self.rows[frame_id][1]))
        return img

    def getImage(self, frame_id):
        return self._gray_as_bgr("cam0", frame_id)

    def getImageRight(self, frame_id):
        return self._gray_as_bgr("cam1", frame_id) if self.sensor_type == SensorType.STEREO else None

    def getDepth(self, frame_id):
        return None

    def getPoseTwc(self, frame_id):
        if len(self.gt) == 0:
            return None
        t = self.rows[frame_id][0]
        k = int(np.argmin(np.abs(self.gt[:, 0] - t)))
        if abs(self.gt[k, 0] - t) * 1e-9 > self.max_pose_dt:
            return None
        T_wb = np.eye(4)
        qw, qx, qy, qz = self.gt[k, 4:8]
        T_wb[:3, :3] = quat_to_R(qx, qy, qz, qw)
        T_wb[:3, 3] = self.gt[k, 1:4]
        return T_wb @ self.T_BS


def camera_from_settings(path):
    """Camera.* block of a pySLAM settings yaml (settings/TUM1.yaml etc.) -> CameraRecord."""
    import yaml

    with open(path) as f:
        txt = f.read()
    if txt.startswith("%YAML"):
        txt = "\n".join(txt.splitlines()[1:])  # OpenCV's %YAML:1.0 header
    s = yaml.safe_load(txt)
    D = [s.get(f"Camera.{k}", 0.0) for k in ("k1", "k2", "p1", "p2", "k3")]
    dmf = float(s.get("DepthMapFactor", 1.0) or 1.0)
    return CameraRecord(s["Camera.width"], s["Camera.height"], s["Camera.fx"], s["Camera.fy"], s["Camera.cx"], s["Camera.cy"], D,
                        s.get("Camera.bf"), None, 1.0 / dmf, s.get("Camera.fps", 30))


def dataset_factory(kind, path, name, camera, **kw):
    """kind: 'tum' | 'icl_nuim' | 'replica' | 'scannet' | 'euroc' (cf. pyslam/io/dataset_factory.py)."""
    table = {"tum": TumDataset, "icl_nuim": IclNuimDataset, "replica": ReplicaDataset, "scannet": ScannetDataset,
             "euroc": EurocDataset}
    k = kind.lower()
    if k not in table:
        raise ValueError(f"unknown dataset type: {kind}")
    return table[k](path, name, camera, **kw)
