"""GPU: the tile-sharded multi-GPU path on real HIP volumes.  A 1-GPU box cannot host two RCCL
ranks, so two processes share GPU 0 and talk over gloo; kernels, tiles, export/import and the merge
logic are exactly what the N-GPU RCCL run executes (only the transport differs)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmpdir, sharding):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from pyslam_amd.distributed import ShardedTSDF
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = SyntheticRGBD("tiny_160x120_2cm")
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    fuser = ShardedTSDF(0.02, 0.08, s.width, s.height, device=0, max_blocks=1 << 13, rank=rank, world_size=world,
                        sharding=sharding)
    frames = [s[i] for i in range(4)]
    for d, c, T in frames[:2]:  # online path
        fuser.integrate(RGBDImage(c, d, 1.0, 4.0), K, T)
    fuser.integrate_batch(np.stack([f[0] for f in frames[2:]]), np.stack([f[1] for f in frames[2:]]), K,
                          np.stack([f[2] for f in frames[2:]]), 1.0, 4.0)  # multi-frame sweep path
    local_units = fuser.volume.num_blocks()
    n = fuser.merge(root=0)
    if rank == 0:
        np.save(os.path.join(tmpdir, "local_units.npy"), np.array([local_units]))
        keys, tsdf, w, col = fuser.volume.dump()
        np.savez(os.path.join(tmpdir, "merged.npz"), keys=keys, tsdf=tsdf, w=w, col=col, n=n)
    else:
        assert fuser.volume.num_blocks() == 0
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sharding", ["owner", "tile"])
def test_two_ranks_equal_single_gpu(tmp_path, sharding):
    import torch.multiprocessing as mp

    import oracle
    from pyslam_amd.synthetic import SyntheticRGBD

    port = 29600 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path), sharding), nprocs=2, join=True)
    z = np.load(tmp_path / "merged.npz")
    s = SyntheticRGBD("tiny_160x120_2cm")
    K = np.array(s.intrinsics)
    full = oracle.PortTsdf(0.02, 0.08)
    for i in range(4):
        d, c, T = s[i]
        full.integrate(d, c, K, T, 1.0, 4.0)
    k, tsdf, w, col = full.dump()
    np.testing.assert_array_equal(z["keys"], k)      # the union of the ranks' units == all touched units
    np.testing.assert_array_equal(z["w"], w)         # each voxel update lands on exactly one tile per frame
    assert np.abs(z["tsdf"] - tsdf).max() <= 1e-4    # north-star tolerance (numerators are float32 sums)
    assert np.abs(z["col"] - col).max() / 255.0 <= 1e-4
    local = int(np.load(tmp_path / "local_units.npy")[0])
    if sharding == "owner":  # a rank stores only its share of the units (hash-balanced), and nothing is double counted
        assert 0.3 * len(k) < local < 0.7 * len(k)
        assert np.abs(z["tsdf"] - tsdf).max() <= 5e-6  # disjoint units: the sweep's batch fold and the export/import round trip round
    else:  # a rank allocates only the units that can project into its image tile
        assert 0.3 * len(k) < local < len(k)


def _halo_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from pyslam_amd.distributed import ShardedTSDF
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = SyntheticRGBD("tiny_160x120_2cm")
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    fuser = ShardedTSDF(0.02, 0.08, s.width, s.height, device=0, max_blocks=1 << 13, rank=rank, world_size=world, sharding="tile")
    frames = [s[i] for i in range(6)]
    stats = []
    for lo in (0, 3):  # two fuse + merge rounds: the second merge must again be a plain sum
        fuser.integrate(RGBDImage(frames[lo][1], frames[lo][0], 1.0, 4.0), K, frames[lo][2])  # online path
        fuser.integrate_batch(np.stack([f[0] for f in frames[lo + 1:lo + 3]]), np.stack([f[1] for f in frames[lo + 1:lo + 3]]), K,
                              np.stack([f[2] for f in frames[lo + 1:lo + 3]]), 1.0, 4.0)   # multi-frame sweep path
        dirty = len(fuser.volume.dirty_keys())
        n_shared, n_dirty = fuser.merge_halo()
        assert n_dirty == dirty and len(fuser.volume.dirty_keys()) == 0
        stats.append((n_shared, n_dirty, fuser.volume.num_blocks(), fuser.last_halo["payload_bytes"]))
    np.save(os.path.join(tmpdir, f"stats{rank}.npy"), np.array(stats))
    n = fuser.gather_to_root(root=0)
    if rank == 0:
        keys, tsdf, w, col = fuser.volume.dump()
        np.savez(os.path.join(tmpdir, "gathered.npz"), keys=keys, tsdf=tsdf, w=w, col=col, n=n)
    dist.barrier()
    dist.destroy_process_group()


def test_tile_halo_merge_on_hip_volumes(tmp_path):
    """Image-tile sharding on real HIP volumes (two ranks on GPU 0 over gloo): merge_halo reduces only the units both ranks
    stamped (payload = shared x 81 920 B, fewer than either rank's units), twice in a row, and the volume gathered
    afterwards matches the oracle's single volume: keys and weights exact, tsdf / colour <= 1e-4."""
    import torch.multiprocessing as mp

    import oracle
    from pyslam_amd.synthetic import SyntheticRGBD

    port = 29600 + ((os.getpid() + 555) % 2000)
    mp.spawn(_halo_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    s0, s1 = np.load(tmp_path / "stats0.npy"), np.load(tmp_path / "stats1.npy")
    for r in range(2):
        assert s0[r][0] == s1[r][0] > 0                                 # same plan on both ranks
        assert s0[r][3] == s0[r][0] * 4096 * 5 * 4                      # only shared units travelled
        assert s0[r][0] < min(s0[r][2], s1[r][2])                       # ... and they are fewer than either rank's units
    z = np.load(tmp_path / "gathered.npz")
    s = SyntheticRGBD("tiny_160x120_2cm")
    K = np.array(s.intrinsics)
    full = oracle.PortTsdf(0.02, 0.08)
    for i in range(6):
        d, c, T = s[i]
        full.integrate(d, c, K, T, 1.0, 4.0)
    k, tsdf, w, col = full.dump()
    np.testing.assert_array_equal(z["keys"], k)
    np.testing.assert_array_equal(z["w"], w)
    assert np.abs(z["tsdf"] - tsdf).max() <= 1e-4
    assert np.abs(z["col"] - col).max() / 255.0 <= 1e-4


@pytest.mark.parametrize("world", [2, 3, 8])
def test_coherent_ownership_partitions_every_batch_exactly(world):
    """hv_tsdf_set_sharding(COHERENT): `world` volumes in one process play the ranks.  Every rank plans each batch on its own -
    no communication - and must arrive at the same plan: per batch the ranks' touched-unit lists are disjoint, their union is the
    single volume's list, and the SUM of the ranks' additive numerators (what merge_halo / gather_to_root reduce) is the single
    volume's state: weights and colour sums exact, tsdf within the sweep's fold tolerance.  Three batches of a moving camera, so
    ownership migrates and units end up on several ranks."""
    import torch

    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, ScalableTSDFVolume

    s = SyntheticRGBD("tiny_160x120_2cm")
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    single = ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 13)
    ranks = [ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 13) for _ in range(world)]
    for r, v in enumerate(ranks):
        v.set_owner(r, world, coherent=True)
    multi = 0
    for lo in (0, 20, 60):
        frames = [s[lo + 2 * i] for i in range(12)]
        d = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
        c = torch.from_numpy(np.stack([f[1] for f in frames])).cuda()
        T = np.stack([f[2] for f in frames])
        single.mark_merged()  # dirty_keys() below = the units this batch stamped
        single.integrate_batch(d, c, K, T, 1.0, 4.0)
        want = {tuple(k) for k in single.dirty_keys().tolist()}
        seen = set()
        sizes = []
        for v in ranks:
            v.mark_merged()
            v.integrate_batch(d, c, K, T, 1.0, 4.0)
            mine = {tuple(k) for k in v.dirty_keys().tolist()}
            assert not (mine & seen)  # a unit of a batch has exactly one owner
            seen |= mine
            sizes.append(len(mine))
        assert seen == want
        assert min(sizes) > 0  # every rank got a share of the batch
        if lo == 20:  # an online frame between two planned batches (single frames use the hash ownership in both modes)
            from pyslam_amd.volumetric import RGBDImage

            dd, cc, TT = s[lo + 30]
            for v in [single] + ranks:
                v.integrate(RGBDImage(cc, dd, 1.0, 4.0), K, TT)
    keys = single.unit_keys()
    ref = single.export_numerators(keys)
    total = np.zeros_like(ref)
    held = np.zeros(len(keys), np.int32)
    for v in ranks:
        assert v.dropped_points() == 0
        total += v.export_numerators(keys)
        mine = {tuple(k) for k in v.unit_keys().tolist()}
        held += np.array([tuple(k) in mine for k in keys.tolist()], np.int32)
        assert mine <= {tuple(k) for k in keys.tolist()}
    assert (held >= 1).all() and (held > 1).sum() > 0  # ownership moved with the camera: some units live on two ranks
    np.testing.assert_array_equal(total[..., 1:], ref[..., 1:])  # weight, colour sums: exact integers
    w = np.maximum(ref[..., 1], 1.0)
    assert np.abs(total[..., 0] - ref[..., 0]).max() <= 2e-5 * w.max()
    assert (np.abs(total[..., 0] - ref[..., 0]) / w).max() <= 1e-4  # tsdf within the north-star tolerance
