"""pySLAM's saved system state, as far as the dense path needs it.

``Slam.save_system_state(path)`` (pyslam/slam/slam.py:335-398) writes ``path/map.json``:
    {"USE_CPP_CORE", "sensor_type": "SensorType.RGBD", "environment_type": "DatasetEnvironmentType.INDOOR",
     "map": {"frames": [...], "keyframes": [KeyFrame.to_json()...], "points": [...], "viewer_scale", ...}, ...}
and ``main_map_dense_reconstruction.py`` reloads it (slam.py:400-445) only to walk ``map.get_keyframes()`` and feed
``kf, kf.img, kf.img_right, kf.depth_img`` to the volumetric integrator.  This module reads exactly those fields
(Frame.to_json, pyslam/slam/frame.py:657-727; KeyFrame.to_json, keyframe.py:373-404; PinholeCamera.to_json,
camera.py:323-353; images as NumpyB64Json, pyslam/utilities/serialization.py:421-484) into light records that
satisfy the KeyFrame protocol the integrator consumes, and can write the same subset back (tests, dataset
conversion)."""
import base64
import json
import os

import numpy as np

from ..dense.volumetric_integrator_types import DatasetEnvironmentType, SensorType


# ---- NumpyB64Json (serialization.py:421-484) ---------------------------------------------------------
def numpy_to_json(arr, order=None):
    if not isinstance(arr, np.ndarray):
        raise TypeError(f"numpy_to_json: Expected np.ndarray, got {type(arr)}")
    if order is None:
        order = "F" if arr.flags["F_CONTIGUOUS"] and not arr.flags["C_CONTIGUOUS"] else "C"
    return {"type": "npB64", "dtype": arr.dtype.str, "shape": list(arr.shape), "order": order,
            "data": base64.b64encode(arr.tobytes(order=order)).decode("ascii")}


def numpy_from_json(data):
    """NumpyB64Json / NumpyJson / plain nested lists -> ndarray (or None)."""
    if data is None:
        return None
    if isinstance(data, dict) and data.get("type") == "npB64":
        a = np.frombuffer(base64.b64decode(data["data"]), dtype=np.dtype(data["dtype"]))
        return a.reshape(tuple(int(x) for x in data["shape"]), order=data.get("order", "C")).copy()
    if isinstance(data, dict) and data.get("type") == "np":
        return np.array(data["data"], dtype=np.dtype(data["dtype"])).reshape(data["shape"])
    if isinstance(data, str):
        return numpy_from_json(json.loads(data))
    return np.array(data)


def _pose_from_json(data):
    """extract_tcw_matrix_from_pose_data: a 4x4 list, or a dict holding 'Tcw'."""
    if data is None:
        return None
    if isinstance(data, dict):
        data = data.get("Tcw", data.get("pose"))
    a = np.asarray(numpy_from_json(data), dtype=np.float64)
    return a.reshape(4, 4) if a.size == 16 else None


def _enum_from_json(enum_cls, s, default):
    if s is None:
        return default
    name = str(s).split(".")[-1]
    return enum_cls[name] if name in enum_cls.__members__ else default


class CameraRecord:
    """The PinholeCamera fields the dense path reads (fx, fy, cx, cy, width, height, D, bf, depth_factor)."""

    def __init__(self, width, height, fx, fy, cx, cy, D=None, bf=None, b=None, depth_factor=1.0, fps=30, sensor_type=None):
        self.width, self.height = int(width), int(height)
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)
        self.D = np.zeros(5) if D is None else np.asarray(D, dtype=np.float64).ravel()
        self.bf, self.b = bf, b
        self.depth_factor = 1.0 if depth_factor is None else float(depth_factor)
        self.fps = fps
        self.sensor_type = sensor_type
        self.K = np.array([[self.fx, 0, self.cx], [0, self.fy, self.cy], [0, 0, 1.0]])
        self.is_distorted = bool(np.linalg.norm(self.D) > 1e-10)

    @staticmethod
    def from_json(j):
        if isinstance(j, str):
            j = json.loads(j)
        D = j.get("D")
        D = json.loads(D) if isinstance(D, str) else D
        return CameraRecord(j["width"], j["height"], j["fx"], j["fy"], j["cx"], j["cy"], D, j.get("bf"), j.get("b"),
                            j.get("depth_factor", 1.0), j.get("fps", 30), j.get("sensor_type"))

    def to_json(self):  # camera.py:323-353 (the fields read back above)
        return {"type": 0, "width": self.width, "height": self.height, "fx": self.fx, "fy": self.fy, "cx": self.cx, "cy": self.cy,
                "D": json.dumps(self.D.astype(float).tolist()), "fps": self.fps, "bf": self.bf, "b": self.b,
                "depth_factor": self.depth_factor, "is_distorted": self.is_distorted,
                "K": json.dumps(self.K.tolist()), "sensor_type": self.sensor_type}


class KeyFrameRecord:
    """What VolumetricIntegrationKeyframeData / add_keyframe read of a KeyFrame
    (volumetric_integrator_base.py:101-137, 1153-1161)."""

    def __init__(self, id, pose_Tcw, camera, img, depth_img=None, img_right=None, semantic_img=None, semantic_instances_img=None,
                 timestamp=0.0, img_id=None, kid=None, lba_count=1, is_bad=False):
        self.id, self.kid, self.img_id = int(id), int(id if kid is None else kid), int(id if img_id is None else img_id)
        self.timestamp = float(timestamp)
        self._pose = np.asarray(pose_Tcw, dtype=np.float64).reshape(4, 4)
        self.camera = camera
        self.img, self.img_right, self.depth_img = img, img_right, depth_img
        self.semantic_img, self.semantic_instances_img = semantic_img, semantic_instances_img
        self.lba_count = int(lba_count)
        self._is_bad = bool(is_bad)
        self.is_keyframe = True

    def pose(self):
        return self._pose

    Tcw = property(lambda self: self._pose)

    def is_bad(self):
        return self._is_bad

    def is_semantics_available(self):
        return self.semantic_img is not None

    @staticmethod
    def from_json(j, default_camera=None):
        cam = CameraRecord.from_json(j["camera"]) if j.get("camera") is not None else default_camera
        depth = numpy_from_json(j.get("depth_img"))
        return KeyFrameRecord(j["id"], _pose_from_json(j["pose"]), cam, numpy_from_json(j.get("img")), depth,
                              numpy_from_json(j.get("img_right")), numpy_from_json(j.get("semantic_img")),
                              numpy_from_json(j.get("semantic_instances_img")), j.get("timestamp", 0.0), j.get("img_id"),
                              j.get("kid"), j.get("lba_count", 0), j.get("_is_bad", False))

    def to_json(self):  # frame.py:657-727 + keyframe.py:373-404, the subset read back above
        enc = lambda a: None if a is None else numpy_to_json(np.ascontiguousarray(a))  # noqa: E731
        return {"id": self.id, "timestamp": self.timestamp, "img_id": self.img_id, "pose": self._pose.tolist(),
                "camera": self.camera.to_json(), "is_keyframe": True, "kid": self.kid, "_is_bad": self._is_bad,
                "lba_count": self.lba_count, "img": enc(self.img), "depth_img": enc(self.depth_img), "img_right": enc(self.img_right),
                "semantic_img": enc(self.semantic_img), "semantic_instances_img": enc(self.semantic_instances_img)}


class MapRecord:
    """map.get_keyframes() / map.num_keyframes() / map.keyframes of the reference Map (pyslam/slam/map.py)."""

    def __init__(self, keyframes, viewer_scale=-1.0):
        self.keyframes = list(keyframes)
        self.viewer_scale = viewer_scale

    def get_keyframes(self):
        return list(self.keyframes)

    def num_keyframes(self):
        return len(self.keyframes)


class SystemState:
    def __init__(self, map, camera, sensor_type, environment_type):
        self.map, self.camera = map, camera
        self.sensor_type, self.environment_type = sensor_type, environment_type


def load_system_state(path):
    """path: folder holding map.json (or the file itself) -> SystemState."""
    map_file = path if path.endswith(".json") else os.path.join(path, "map.json")
    if not os.path.exists(map_file):
        raise FileNotFoundError(f"SLAM: File does not exist: {map_file}")
    with open(map_file, "rb") as f:
        j = json.loads(f.read())
    mj = j["map"]
    if isinstance(mj, str):
        mj = json.loads(mj)
    kfs = [KeyFrameRecord.from_json(k) for k in mj.get("keyframes", [])]
    kfs = sorted((k for k in kfs if not k.is_bad()), key=lambda k: k.id)
    camera = kfs[0].camera if kfs else None
    return SystemState(MapRecord(kfs, mj.get("viewer_scale", -1.0)), camera,
                       _enum_from_json(SensorType, j.get("sensor_type"), SensorType.RGBD),
                       _enum_from_json(DatasetEnvironmentType, j.get("environment_type"), DatasetEnvironmentType.INDOOR))


def save_system_state(path, keyframes, sensor_type=SensorType.RGBD, environment_type=DatasetEnvironmentType.INDOOR):
    """Writes path/map.json with the layout of Slam.save_system_state restricted to what load_system_state reads."""
    os.makedirs(path, exist_ok=True)
    out = {"USE_CPP_CORE": False, "sensor_type": f"SensorType.{sensor_type.name}",
           "environment_type": f"DatasetEnvironmentType.{environment_type.name}",
           "map": {"frames": [], "keyframes": [k.to_json() for k in keyframes if not k.is_bad()], "points": [],
                   "keyframe_origins": [], "viewer_scale": -1.0}}
    with open(os.path.join(path, "map.json"), "w") as f:
        f.write(json.dumps(out))
