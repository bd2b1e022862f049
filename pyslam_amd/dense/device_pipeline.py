"""Keyframe uploads beside the previous keyframe's kernels.

A semantic keyframe is 19 MB of host images at the ScanNet resolution (depth f32, colour u8x3, class and instance ids i32):
0.33 ms of PCIe even from page-locked memory, as long as the kernels of one keyframe.  On ONE stream the upload of keyframe k + 1
waits for the kernels of keyframe k; here the uploads go to a copy stream of their own and the compute stream (the volume's,
adopted from torch: _Volume.adopt_torch_stream) only waits for the event of the keyframe it is about to fuse."""
import numpy as np


class KeyframeUploader:
    def __init__(self, volume):
        import torch

        self.torch = torch
        self.volume = volume
        self.device = torch.device("cuda", int(volume._cfg.device))
        self.compute = volume.adopt_torch_stream()
        self.copy = torch.cuda.Stream(self.device)
        self.prep = torch.cuda.Stream(self.device)

    def upload(self, arrays, prep=None):
        """arrays: dict name -> (ndarray | None, dtype).  -> (dict name -> CUDA tensor | None, event recorded on the copy stream).
        prep(tensors, stream): work that depends on the keyframe's images ALONE (the shadow-point filter of its depth: seven launches,
        ~65 us at 1296x968, nothing of the volume read), queued on the prep stream behind the uploads - it runs beside the kernels of
        the keyframe being fused instead of in front of this keyframe's own; may add / replace entries of `tensors`."""
        torch = self.torch
        out = {}
        with torch.cuda.stream(self.copy):
            for name, (a, dtype) in arrays.items():
                if a is None:
                    out[name] = None
                    continue
                if hasattr(a, "data_ptr"):  # already a tensor (a depth estimator's output lives in HBM)
                    t = a.to(self.device, dtype=torch.from_numpy(np.zeros(0, dtype)).dtype, non_blocking=True).contiguous()
                else:
                    t = torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(self.device, non_blocking=True)
                t.record_stream(self.compute)  # allocated on the copy stream, consumed on the compute stream
                out[name] = t
            ev = torch.cuda.Event()
            ev.record(self.copy)
        if prep is not None:
            # a stream of its own: the copy stream goes on with the next keyframe's uploads
            with torch.cuda.stream(self.prep):
                self.prep.wait_event(ev)
                before = dict(out)
                for t in before.values():
                    if t is not None:
                        t.record_stream(self.prep)  # (allocated on the copy stream, read here)
                prep(out, self.prep)
                for name, t in out.items():
                    if t is not None and t is not before.get(name):
                        t.record_stream(self.compute)
                ev = torch.cuda.Event()
                ev.record(self.prep)
        return out, ev

    def run(self, items, to_arrays, body, prep=None):
        """For every item: to_arrays(item) -> dict for upload(); body(item, tensors) runs under the compute stream once the item's
        uploads (and prep, see upload()) have landed.  The uploads of item k + 1 are queued before body(item k) is."""
        torch = self.torch
        items = list(items)
        if not items:
            return
        nxt = self.upload(to_arrays(items[0]), prep)
        for k, item in enumerate(items):
            tensors, ev = nxt
            nxt = self.upload(to_arrays(items[k + 1]), prep) if k + 1 < len(items) else None
            with torch.cuda.stream(self.compute):
                self.compute.wait_event(ev)
                body(item, tensors)
        # the host images are handed back to their owner (ring slots) when this returns: every upload must have read them
        self.copy.synchronize()
        if prep is not None:
            self.prep.synchronize()  # (its scratch and inputs belong to tensors the caller may drop now)
