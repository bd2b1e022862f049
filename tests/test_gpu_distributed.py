"""GPU: the tile-sharded multi-GPU path on real HIP volumes.  A 1-GPU box cannot host two RCCL
ranks, so two processes share GPU 0 and talk over gloo; kernels, tiles, export/import and the merge
logic are exactly what the N-GPU RCCL run executes (only the transport differs)."""
import os
import sys

import ctypes

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


# ---- the shape bench.py --gpus N times: 640x480 / 5 mm / B = 32 (two sliding batches) and B = 64 (one), every sharding, 2 and 8 ranks -----------------
_BENCH = {}


def _bench_case():
    """Frames, the oracle's volume after 64 frames and a single-GPU volume of the same stream (built once per session)."""
    if not _BENCH:
        import torch

        import oracle
        from pyslam_amd.volumetric import PinholeCameraIntrinsic
        from tests.conftest import synthetic_frames

        B = 32
        s, frames = synthetic_frames("synthetic_640x480_5mm", 0, 2 * B)
        K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
        cpu = oracle.PortTsdf(0.005, 0.04, threads=min(32, os.cpu_count() or 1))
        for d, c, T in frames:
            cpu.integrate(d, c, K.as_array(), T, 1.0, 4.0)
        batches = []
        for step in range(2):
            fs = frames[step * B:(step + 1) * B]
            batches.append((torch.from_numpy(np.stack([f[0] for f in fs])).cuda(), torch.from_numpy(np.stack([f[1] for f in fs])).cuda(),
                            np.stack([f[2] for f in fs])))
        _BENCH.update(s=s, K=K, batches=batches, dump=cpu.dump())
    return _BENCH


@pytest.mark.parametrize("world", [3, 8])
def test_device_halo_lists_and_plan_equal_the_host_plan_and_merge(world):
    """The halo merge with key lists and plan in device memory (hv_halo.hip: what ShardedTSDF.merge_halo runs over RCCL) against the host
    path (hv_tsdf_dirty_keys / hv_tsdf_unit_keys -> hv_merge_halo_plan_held -> hv_merge_halo_pack / _unpack): `world` tile-sharded
    volumes in one process play the ranks; two merge windows (the second with units that were merged before and units only one rank
    wrote to).  Per rank: same dirty and held key sets, same shared keys with the same action; and after pack -> sum over the ranks
    -> unpack through the planned entry points every rank's volume equals the one the host path leaves."""
    import torch

    from pyslam_amd import _lib as L
    from pyslam_amd.distributed import tile_bounds
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage, ScalableTSDFVolume

    s = SyntheticRGBD("tiny_160x120_2cm")
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    lib = L.load()

    def make():
        vols = [ScalableTSDFVolume(0.02, 0.08, max_blocks=1 << 12) for _ in range(world)]
        for r, v in enumerate(vols):
            v.set_tile(*tile_bounds(r, world, s.width, s.height))
        return vols

    dev_ranks, host_ranks = make(), make()
    for window, frame_ids in enumerate(((0, 3, 6), (9, 12))):
        for vols in (dev_ranks, host_ranks):
            for i in frame_ids:
                d, c, T = s[i]
                for v in vols:
                    v.integrate(RGBDImage(c, d, 1.0, 4.0), K, T)
        # ---- host path (the reference of this test)
        mine = [np.ascontiguousarray(v.dirty_keys(), dtype=np.int32).reshape(-1, 3) for v in host_ranks]
        held = [np.ascontiguousarray(v.unit_keys(), dtype=np.int32).reshape(-1, 3) for v in host_ranks]
        dk, dc = np.ascontiguousarray(np.concatenate(mine)), np.array([len(x) for x in mine], np.int64)
        hk, hc = np.ascontiguousarray(np.concatenate(held)), np.array([len(x) for x in held], np.int64)
        host_plans = []
        for r in range(world):
            n = ctypes.c_int64()
            L.check(lib.hv_merge_halo_plan_held(L.ptr(dk), L.ptr(dc), L.ptr(hk), L.ptr(hc), world, r, None, None, 0, ctypes.byref(n)))
            shared, action = np.zeros((n.value, 3), np.int32), np.zeros(n.value, np.uint8)
            if n.value:
                L.check(lib.hv_merge_halo_plan_held(L.ptr(dk), L.ptr(dc), L.ptr(hk), L.ptr(hc), world, r, L.ptr(shared), L.ptr(action), n.value, ctypes.byref(n)))
            host_plans.append((shared, action))
        assert len(host_plans[0][0]) > 0 and any((a == 1).any() for _, a in host_plans) and any((a == 2).any() for _, a in host_plans)
        # ---- device path: lists, "all-gather" (a concatenation here), plan
        lists = []
        for v in dev_ranks:
            cap = max(v.num_blocks(), 1)
            d_t, h_t = torch.empty(cap, dtype=torch.int64, device="cuda"), torch.empty(cap, dtype=torch.int64, device="cuda")
            nd, nh = v.halo_lists_device(d_t, h_t)
            lists.append((d_t[:nd].clone(), h_t[:nh].clone()))
        for r in range(world):
            assert len(lists[r][0]) == len(mine[r]) and len(lists[r][1]) == len(held[r])
        sd, sh = max(max(len(a) for a, _ in lists), 1), max(max(len(b) for _, b in lists), 1)
        dirty_all, held_all = torch.zeros((world, sd), dtype=torch.int64, device="cuda"), torch.zeros((world, sh), dtype=torch.int64, device="cuda")
        for r, (a, b) in enumerate(lists):
            dirty_all[r, :len(a)] = a
            held_all[r, :len(b)] = b
        torch.cuda.synchronize()
        for r, v in enumerate(dev_ranks):
            k = v.halo_plan_device(dirty_all, dc, held_all, hc, world, r)
            keys, action = v.halo_plan_fetch()
            assert k == len(keys) == len(host_plans[r][0])
            got = sorted(map(tuple, np.concatenate([keys, action[:, None].astype(np.int32)], axis=1).tolist()))
            want = sorted(map(tuple, np.concatenate([host_plans[r][0], host_plans[r][1][:, None].astype(np.int32)], axis=1).tolist()))
            assert got == want
        # ---- the merge itself through both paths: pack, sum over the ranks (the all-reduce), unpack
        k = len(host_plans[0][0])
        total = sum(v.export_numerators(host_plans[0][0]) for v in host_ranks)
        for r, v in enumerate(host_ranks):
            v.halo_unpack(host_plans[r][0], total, host_plans[r][1])
            v.mark_merged()
        payloads = []
        for v in dev_ranks:
            p = torch.empty((k, v.res ** 3, 5), dtype=torch.float32, device="cuda")
            v.halo_pack_planned(0, k, p)
            v.synchronize()
            payloads.append(p)
        total_d = payloads[0].clone()
        for p in payloads[1:]:
            total_d += p  # (the same order of additions as the host path's sum)
        torch.cuda.synchronize()
        for v in dev_ranks:
            half = k // 2  # (two buckets: the range arguments)
            v.halo_unpack_planned(0, half, total_d[:half].contiguous())
            v.halo_unpack_planned(half, k - half, total_d[half:].contiguous())
            v.synchronize()
            v.mark_merged()
        for a, b in zip(dev_ranks, host_ranks):
            for x, y in zip(a.dump(), b.dump()):
                np.testing.assert_array_equal(x, y)
            assert len(a.dirty_keys()) == 0


@pytest.mark.parametrize("batch", [32, 64])
@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("sharding", ["owner", "tile"])
def test_sharded_bench_step_sums_to_the_oracle(sharding, world, batch):
    """VERDICT r04 next #1a: the sharded forms had only been held to the oracle at 160x120 / 2 cm with a handful of frames.  Here
    `world` volumes in one process play the ranks of `bench.py --gpus world --sharding <sharding>` at ITS shape - two consecutive
    32-frame batches of synthetic_640x480_5mm (or the same 64 frames as ONE batch: the bench's step since round 6) through
    integrate_batch, device-resident frames - and the SUM over the ranks of the
    additive numerators (what gather_to_root / merge_halo reduce) is compared with oracle.PortTsdf fusing the same 64 frames one
    by one: unit set and weights exact, colour sums exact integers (<= 1e-4 of the mean), tsdf within the north star's 1e-4
    (owner sharding, whose unit sets are disjoint: within the batch fold's 5e-6)."""
    from pyslam_amd.distributed import tile_bounds
    from pyslam_amd.volumetric import ScalableTSDFVolume
    from tests.conftest import FOLD_TSDF_TOL

    c = _bench_case()
    s, K = c["s"], c["K"]
    ranks = [ScalableTSDFVolume(0.005, 0.04, max_blocks=1 << 14) for _ in range(world)]
    for r, v in enumerate(ranks):
        if sharding == "tile":
            v.set_tile(*tile_bounds(r, world, s.width, s.height))
        else:
            v.set_owner(r, world)
    batches = c["batches"]
    if batch == 64:  # bench.py's default step since round 6: the same 64 frames in ONE call
        import torch

        batches = [(torch.cat([b[0] for b in batches]), torch.cat([b[1] for b in batches]), np.concatenate([b[2] for b in batches]))]
    for d, col, T in batches:
        for v in ranks:
            v.integrate_batch(d, col, K, T, depth_scale=1.0, depth_trunc=4.0)
    keys, tsdf, w, colour = c["dump"]
    held = [{tuple(k) for k in v.unit_keys().tolist()} for v in ranks]
    assert set().union(*held) == {tuple(k) for k in keys.tolist()}  # the ranks' units together are exactly the reference's
    if sharding == "owner":
        assert sum(len(h) for h in held) == len(keys)  # ... and disjoint
        assert max(len(h) for h in held) < 1.25 * len(keys) / world  # hash-balanced
    worst_t = worst_c = 0.0
    for lo in range(0, len(keys), 256):
        sub = np.ascontiguousarray(keys[lo:lo + 256])
        total = np.zeros((len(sub), 4096, 5), np.float64)
        for v in ranks:
            assert v.dropped_points() == 0
            total += v.export_numerators(sub)
        # a unit's voxels are exported in pool order (word = z * 256 + x * 16 + y), the dumps list them as Open3D does (x * 256 + y * 16 + z)
        total = total.reshape(len(sub), 16, 16, 16, 5).transpose(0, 2, 3, 1, 4).reshape(len(sub), 4096, 5)
        np.testing.assert_array_equal(total[..., 1], w[lo:lo + 256])  # weights: every (voxel, frame) update landed on exactly one rank
        ww = np.maximum(w[lo:lo + 256], 1.0)
        worst_t = max(worst_t, float(np.abs(total[..., 0] / ww - tsdf[lo:lo + 256] * (w[lo:lo + 256] > 0)).max()))
        worst_c = max(worst_c, float(np.abs(total[..., 2:] / ww[..., None] - colour[lo:lo + 256] * (w[lo:lo + 256] > 0)[..., None]).max()))
    assert worst_c / 255.0 <= 1e-4
    assert worst_t <= (2 * FOLD_TSDF_TOL if sharding == "owner" else 1e-4)


def _nccl_world1_worker(rank, port, tmpdir):
    """One process, one GPU, backend nccl (= RCCL) with world_size 1: every `on_gpu` branch of pyslam_amd/distributed.py runs."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    from pyslam_amd.distributed import ShardedSemanticGrid, ShardedTSDF, ShardedVoxelGrid
    from pyslam_amd.synthetic import SyntheticRGBD
    from pyslam_amd.volumetric import PinholeCameraIntrinsic, RGBDImage, VoxelBlockGrid

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    out = {}
    s = SyntheticRGBD("tiny_160x120_2cm")
    K = PinholeCameraIntrinsic(s.width, s.height, *s.intrinsics)
    frames = [s[i] for i in range(6)]
    # -- TSDF, tile form: merge_halo (CUDA key buffers, CUDA payload into export_numerators / halo_unpack, RCCL all-reduce) ---------
    fuser = ShardedTSDF(0.02, 0.08, s.width, s.height, device=0, max_blocks=1 << 13, rank=0, world_size=1, sharding="tile",
                        force_collectives=True)
    plain = ShardedTSDF(0.02, 0.08, s.width, s.height, device=0, max_blocks=1 << 13)
    for f in (fuser, plain):
        for d, c, T in frames[:2]:
            f.integrate(RGBDImage(c, d, 1.0, 4.0), K, T)
        f.integrate_batch(np.stack([x[0] for x in frames[2:]]), np.stack([x[1] for x in frames[2:]]), K, np.stack([x[2] for x in frames[2:]]), 1.0, 4.0)
    dirty = len(fuser.volume.dirty_keys())
    n_shared, n_dirty = fuser.merge_halo()
    out["halo"] = (n_shared, n_dirty, dirty, fuser.last_halo["payload_bytes"], len(fuser.volume.dirty_keys()))
    a, b = fuser.volume.dump(), plain.volume.dump()
    out["halo_keys_equal"] = bool(np.array_equal(a[0], b[0]))
    out["halo_w_equal"] = bool(np.array_equal(a[2], b[2]))
    out["halo_tsdf_err"] = float(np.abs(a[1] - b[1]).max())
    out["halo_col_err"] = float(np.abs(a[3] - b[3]).max())
    # -- gather_to_root (export -> RCCL reduce -> import on the root) -------------------------------------------------------------
    n = fuser.gather_to_root(root=0)
    a = fuser.volume.dump()
    out["gather"] = (int(n), bool(np.array_equal(a[0], b[0])), bool(np.array_equal(a[2], b[2])), float(np.abs(a[1] - b[1]).max()))
    # -- VOXEL_GRID: gather_voxels --------------------------------------------------------------------------------------------------
    g = VoxelBlockGrid(0.02)
    sg = ShardedVoxelGrid(g, rank=0, world_size=1, force_collectives=True)
    d, c, T = frames[0]
    pts = np.random.default_rng(0).uniform(-1, 1, (5000, 3)).astype(np.float32)
    cols = np.random.default_rng(1).uniform(0, 1, (5000, 3)).astype(np.float32)
    sg.integrate(pts, cols)
    gp, gc = sg.gather_voxels(1)
    v = g.get_voxels(1)
    out["vg"] = (len(gp), bool(np.array_equal(np.sort(gp, axis=0), np.sort(np.asarray(v.points, np.float32), axis=0))))
    # -- semantic grids: the association's pair exchange on the device + gather_voxels ------------------------------------------------
    from pyslam_amd.volumetric import CameraFrustrum
    from pyslam_amd.volumetric_semantic import VoxelBlockSemanticGrid, remap_instance_ids, set_next_object_id
    from tests.semantic_helpers import CFG, DEPTH_MAX, DEPTH_MIN, frame_points, semantic_frame

    ss = SyntheticRGBD(CFG, noise=True, invalid_frac=0.02)
    res = {}
    for name in ("plain", "sharded"):
        grid = VoxelBlockSemanticGrid(CFG["voxel"], 8, max_blocks=1 << 14, max_points=1 << 18)
        sh = ShardedSemanticGrid(grid, rank=0, world_size=1, force_collectives=True) if name == "sharded" else None
        if sh is not None:
            assert grid._pair_exchange_device is not None  # nccl: the device-resident exchange is what runs
        grid.set_depth_threshold(2.0)
        fr = CameraFrustrum(*ss.intrinsics, ss.width, ss.height, np.eye(4), depth_max=DEPTH_MAX, depth_min=DEPTH_MIN)
        set_next_object_id(1)
        maps = []
        for k, i in enumerate((0, 6, 12, 18)):
            depth, rgb, T, cls_img, inst_img = semantic_frame(ss, i, shuffle=k)
            fr.set_T_cw(T)
            m = dict(grid.assign_object_ids_to_instance_ids(fr, cls_img, inst_img, depth, depth_threshold=0.05, do_carving=(k == 2),
                                                            min_vote_ratio=0.5, min_votes=3))
            maps.append(sorted(m.items()))
            obj_img = remap_instance_ids(inst_img, m, volume=grid)
            grid.integrate(*frame_points(depth, rgb, T, cls_img, obj_img, ss.intrinsics, 4.0))
        vox = grid.get_voxels(1, -1.0)
        order = np.lexsort(np.asarray(vox.points).T)
        rows = sh.gather_voxels(1, -1.0) if sh is not None else None
        res[name] = (maps, np.asarray(vox.points)[order], np.asarray(vox.object_ids)[order], rows, sh.last_exchange if sh is not None else None)
    out["sem_maps_equal"] = res["plain"][0] == res["sharded"][0] and any(v > 0 for _, v in res["plain"][0][-1])
    out["sem_points_equal"] = bool(np.array_equal(res["plain"][1], res["sharded"][1]))
    out["sem_objects_equal"] = bool(np.array_equal(res["plain"][2], res["sharded"][2]))
    out["sem_rows"] = int(len(res["sharded"][3][0]))
    out["sem_voxels"] = int(len(res["plain"][1]))
    out["sem_exchange"] = res["sharded"][4]
    import json

    with open(os.path.join(tmpdir, "nccl1.json"), "w") as f:
        json.dump(out, f)
    dist.barrier()
    dist.destroy_process_group()


def test_every_nccl_branch_runs_with_one_rank(tmp_path):
    """VERDICT r04 next #1b: every multi-rank test ran over gloo (host tensors); the `nccl` branches of pyslam_amd/distributed.py had
    never executed.  RCCL accepts a one-rank group, so this test initialises backend nccl with world_size 1 on the one GPU and - with
    force_collectives - drives merge_halo (CUDA key buffers, CUDA payload through hv_merge_halo_pack -> all_reduce -> _unpack),
    gather_to_root (export -> reduce -> import), ShardedVoxelGrid.gather_voxels and the semantic association's device-resident pair
    exchange (hv_assoc_pairs_export -> all_gather_into_tensor -> hv_assoc_pairs_import) + gather_voxels through them; the results
    must equal the same volumes without any collective."""
    import json

    import torch.multiprocessing as mp

    port = 29600 + ((os.getpid() + 1234) % 2000)
    mp.spawn(_nccl_world1_worker, args=(port, str(tmp_path)), nprocs=1, join=True)
    out = json.load(open(tmp_path / "nccl1.json"))
    n_shared, n_dirty, dirty, payload, dirty_after = out["halo"]
    assert n_shared == n_dirty == dirty > 0 and payload == n_shared * 4096 * 5 * 4 and dirty_after == 0
    assert out["halo_keys_equal"] and out["halo_w_equal"] and out["halo_tsdf_err"] <= 1e-5 and out["halo_col_err"] <= 1e-3
    n, keys_eq, w_eq, terr = out["gather"]
    assert n > 0 and keys_eq and w_eq and terr <= 1e-5
    assert out["vg"][0] > 0 and out["vg"][1]
    assert out["sem_maps_equal"] and out["sem_points_equal"] and out["sem_objects_equal"]
    assert out["sem_rows"] == out["sem_voxels"] > 0
    assert out["sem_exchange"]["device_resident"] is True
