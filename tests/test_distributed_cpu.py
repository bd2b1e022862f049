"""CPU, world_size 2, gloo: the multi-GPU merge path of pyslam_amd.distributed (ragged key
all-gather, union, bucketed sum-reduce to the root, import on root, clear elsewhere).

The volume is duck-typed: here each rank wraps the CPU oracle (allowed in tests); on GPUs the same
code drives ScalableTSDFVolume over RCCL.  Identity checked: fusing frames {0,1} on rank 0 and {2,3}
on rank 1 and merging == fusing {0,1,2,3} in one volume (weights exact, tsdf/colour to 1e-5)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackedVolume:
    """Minimal volume protocol of TileShardedTSDF on top of oracle.PortTsdf."""

    res = 16

    def __init__(self):
        import oracle

        self.vol = oracle.PortTsdf(0.02, 0.08)
        self.imported = {}

    def set_tile(self, *a):
        pass

    def set_owner(self, *a):
        pass

    def integrate(self, depth, rgb, K, T):
        self.vol.integrate(depth, rgb, K, T, 1.0, 4.0)

    def unit_keys(self):
        return self.vol.dump()[0]

    def export_numerators(self, keys, out=None):
        k, tsdf, w, col = self.vol.dump()
        idx = {tuple(x): i for i, x in enumerate(k)}
        if out is None:
            out = np.zeros((len(keys), 4096, 5), np.float32)
        out[:] = 0
        for j, key in enumerate(keys):
            i = idx.get(tuple(key))
            if i is not None:
                out[j, :, 0] = tsdf[i] * w[i]
                out[j, :, 1] = w[i]
                out[j, :, 2:5] = col[i] * w[i][:, None]
        return out

    def import_numerators(self, keys, payload):
        for j, key in enumerate(keys):
            self.imported[tuple(key)] = np.array(payload[j])

    def reset(self):
        self.vol.reset()


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from pyslam_amd.distributed import TileShardedTSDF, tile_bounds
    from pyslam_amd.synthetic import SyntheticRGBD

    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = SyntheticRGBD("tiny_160x120_2cm")
    K = np.array(s.intrinsics)
    backend = OracleBackedVolume()
    fuser = TileShardedTSDF(0.02, 0.08, s.width, s.height, rank=rank, world_size=world, volume=backend)
    fuser.BUCKET_BYTES = 1 << 20  # force several buckets (12 units per bucket)
    assert fuser.tile == tile_bounds(rank, world, s.width, s.height)
    for i in (0, 1) if rank == 0 else (2, 3):
        d, c, T = s[i]
        backend.integrate(d, c, K, T)
    n = fuser.merge(root=0)
    if rank == 0:
        keys = np.array(sorted(backend.imported))
        payload = np.stack([backend.imported[tuple(k)] for k in keys])
        np.savez(os.path.join(tmpdir, "merged.npz"), keys=keys, payload=payload, n=n)
    else:
        assert backend.vol.num_units() == 0  # non-root ranks are cleared and keep fusing deltas
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_merge_equals_single_volume(tmp_path):
    import torch.multiprocessing as mp

    import oracle
    from pyslam_amd.synthetic import SyntheticRGBD

    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    z = np.load(tmp_path / "merged.npz")
    s = SyntheticRGBD("tiny_160x120_2cm")
    K = np.array(s.intrinsics)
    full = oracle.PortTsdf(0.02, 0.08)
    for i in range(4):
        d, c, T = s[i]
        full.integrate(d, c, K, T, 1.0, 4.0)
    k, tsdf, w, col = full.dump()
    np.testing.assert_array_equal(z["keys"], k)  # union of both ranks' units == all units
    assert int(z["n"]) == len(k)
    p = z["payload"]
    np.testing.assert_array_equal(p[..., 1], w)
    ww = np.maximum(w, 1)
    assert np.abs(p[..., 0] / ww - tsdf).max() < 1e-5
    assert np.abs(p[..., 2:5] / ww[..., None] - col).max() < 1e-3  # 0..255 scale, float32 numerators


def test_tile_bounds_partition_the_image():
    from pyslam_amd.distributed import tile_bounds, union_keys

    for world in (1, 2, 3, 4, 8):
        tiles = [tile_bounds(r, world, 640, 480) for r in range(world)]
        assert tiles[0][0] == 0 and tiles[-1][2] == 640
        for a, b in zip(tiles, tiles[1:]):
            assert a[2] == b[0]
    u = union_keys([np.array([[1, 2, 3], [0, 0, 0]]), np.zeros((0, 3), np.int32), np.array([[1, 2, 3], [-1, 5, 2]])])
    assert u.tolist() == [[-1, 5, 2], [0, 0, 0], [1, 2, 3]]
