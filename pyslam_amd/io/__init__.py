"""On-disk formats either side of the dense path: dataset readers that yield posed RGB-D
keyframes (TUM / ICL-NUIM associations, Replica, ScanNet, EuRoC stereo) and the saved-system-state loader
(``map.json`` written by pySLAM's Slam.save_system_state) that main_map_dense_reconstruction.py replays."""
from .datasets import (  # noqa: F401
    EurocDataset,
    IclNuimDataset,
    ReplicaDataset,
    ScannetDataset,
    TumDataset,
    dataset_factory,
)
from .images import imread_color, imread_unchanged  # noqa: F401
from .system_state import (  # noqa: F401
    CameraRecord,
    KeyFrameRecord,
    MapRecord,
    SystemState,
    load_system_state,
    numpy_from_json,
    numpy_to_json,
    save_system_state,
)
