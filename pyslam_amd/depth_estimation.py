"""Depth-estimator hand-off without a host round trip (reference
pyslam/depth_estimation/depth_estimator_raft_stereo.py:143-177 and volumetric_integrator_base.py:989-1004).

pySLAM's stereo back-end runs a PyTorch network (RAFT-Stereo by default) on the rectified pair, copies the
disparity to the host, converts it with ``depth = bf / |disparity|`` in numpy and hands a numpy depth image to
the integrator, which uploads it again.  Here the network's output stays a CUDA tensor: the disparity -> depth
conversion is a couple of torch ops on the same device (PyTorch-ROCm as plumbing), and the fused integrate calls
of libpyslam_hipvol take the tensor's device pointer (HV_DEVICE).

No network weights are available offline, so the module is injected: any ``torch.nn.Module`` mapping
(left, right) float tensors [1,3,H,W] to a disparity [1,1,H,W] or [H,W] works; ``StubStereoNet`` is a tiny
deterministic stand-in used by the tests (RAFT-Stereo itself is out of this package's scope)."""
import numpy as np


class DepthEstimatorStereoTorch:
    """infer(image, image_right) -> (depth, None); depth is a float32 CUDA tensor when `keep_on_device`."""

    def __init__(self, module, camera, device="cuda", keep_on_device=True, min_depth=0.0, max_depth=np.inf):
        import torch

        self.torch = torch
        self.module = module.to(device).eval()
        self.camera = camera
        self.device = torch.device(device)
        self.keep_on_device = keep_on_device
        self.min_depth, self.max_depth = float(min_depth), float(max_depth)
        self.disparity_map = None

    def infer(self, image, image_right=None):
        if image_right is None:
            raise ValueError("Image right is None. Are you using a stereo dataset? If not, you cant use a stereo depth estimator here.")
        torch = self.torch
        with torch.no_grad():
            left = torch.from_numpy(np.ascontiguousarray(image)).to(self.device).permute(2, 0, 1).float()[None]
            right = torch.from_numpy(np.ascontiguousarray(image_right)).to(self.device).permute(2, 0, 1).float()[None]
            disparity = self.module(left, right)
            if isinstance(disparity, (tuple, list)):
                disparity = disparity[-1]  # RAFT-Stereo returns (low-res flow, up-sampled flow)
            disparity = disparity.squeeze()
            self.disparity_map = disparity
            bf = float(getattr(self.camera, "bf", 1.0) or 1.0)
            a = disparity.abs()
            depth = torch.where(a > 0, bf / a.clamp_min(1e-30), torch.zeros_like(a)).float()  # raft_stereo.py:170-174
            if np.isfinite(self.max_depth):
                depth = torch.where(depth > self.max_depth, torch.zeros_like(depth), depth)
            if self.min_depth > 0:
                depth = torch.where(depth < self.min_depth, torch.zeros_like(depth), depth)
            depth = depth.contiguous()
        if self.keep_on_device:
            torch.cuda.current_stream(self.device).synchronize()  # the volume's HIP stream consumes it next
            return depth, None
        return depth.cpu().numpy(), None


def make_stub_stereo_net(seed=0):
    """A tiny deterministic conv net with a strictly positive disparity output (tests / examples only)."""
    import torch

    class StubStereoNet(torch.nn.Module):
        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(seed)
            self.conv = torch.nn.Conv2d(6, 1, 5, padding=2, bias=True)
            with torch.no_grad():
                self.conv.weight.copy_(torch.randn(self.conv.weight.shape, generator=g) * 0.002)
                self.conv.bias.fill_(0.0)

        def forward(self, left, right):
            x = torch.cat([left, right], dim=1) / 255.0
            return 20.0 + 30.0 * torch.sigmoid(self.conv(x))  # disparity in (20, 50) px

    return StubStereoNet()
