"""pyslam_amd — MI355X-native (gfx950) volumetric fusion for pySLAM's dense-mapping hot path.

Layout
  csrc/            hand-written HIP kernels + the C ABI (include/hipvol.h) -> lib/libpyslam_hipvol.so
  _lib.py          ctypes binding (no CPU fallback)
  volumetric.py    mirror of the `volumetric` pybind module (VoxelBlockGrid, CameraFrustrum) and of
                   the open3d slice pySLAM uses (ScalableTSDFVolume)
  dense/           mirror of pyslam/dense (VolumetricIntegrator* classes, factory, task protocol)
  synthetic.py     deterministic analytic RGB-D scene (stand-in for TUM/Replica streams)
  distributed.py   tile-sharded multi-GPU fusion over torch.distributed (RCCL)
"""
__version__ = "0.1.0"
