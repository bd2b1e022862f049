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
        self.imported = {}   # key -> offset that, added to the oracle's own numerators, gives the unit's state (import / halo unpack)
        self.dirty = set()

    def set_tile(self, *a):
        pass

    def set_owner(self, *a):
        pass

    def integrate(self, depth, rgb, K, T):
        self.vol.integrate(depth, rgb, K, T, 1.0, 4.0)
        self.dirty.update(tuple(k) for k in self.vol.touched_keys())

    def unit_keys(self):
        return self.vol.dump()[0]

    def dirty_keys(self):
        return np.array(sorted(self.dirty), dtype=np.int32).reshape(-1, 3)

    def mark_merged(self):
        self.dirty.clear()

    def _numerators(self, keys):
        """The oracle volume's own additive numerators of `keys` (zeros where it has no such unit)."""
        k, tsdf, w, col = self.vol.dump()
        idx = {tuple(x): i for i, x in enumerate(k)}
        out = np.zeros((len(keys), 4096, 5), np.float32)
        for j, key in enumerate(keys):
            i = idx.get(tuple(key))
            if i is not None:
                out[j, :, 0] = tsdf[i] * w[i]
                out[j, :, 1] = w[i]
                out[j, :, 2:5] = col[i] * w[i][:, None]
        return out

    def _set_state(self, keys, values):
        """State of a unit := value.  Kept as an offset against the oracle volume, so that frames fused AFTERWARDS still add
        their deltas (a HIP volume overwrites the unit's planes and goes on fusing into them)."""
        own = self._numerators(keys)
        for j, key in enumerate(keys):
            self.imported[tuple(key)] = np.asarray(values[j], np.float32) - own[j]

    def halo_unpack(self, keys, payload, action):
        held = {tuple(k) for k in self.unit_keys()} | set(self.imported)
        sel = [j for j, key in enumerate(keys) if action[j] == 1 or (action[j] == 2 and tuple(key) in held)]
        if sel:
            self._set_state([keys[j] for j in sel],
                            [payload[j] if action[j] == 1 else np.zeros_like(np.asarray(payload[j])) for j in sel])

    def export_numerators(self, keys, out=None):
        own = self._numerators(keys)
        if out is None:
            out = np.zeros((len(keys), 4096, 5), np.float32)
        out[:] = own
        for j, key in enumerate(keys):
            if tuple(key) in self.imported:
                out[j] += self.imported[tuple(key)]
        return out

    def import_numerators(self, keys, payload):
        self._set_state(keys, payload)

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
        payload = backend.export_numerators(keys)
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


def _halo_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from pyslam_amd.distributed import TileShardedTSDF
    from pyslam_amd.synthetic import SyntheticRGBD

    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = SyntheticRGBD("tiny_160x120_2cm")
    K = np.array(s.intrinsics)
    backend = OracleBackedVolume()
    fuser = TileShardedTSDF(0.02, 0.08, s.width, s.height, rank=rank, world_size=world, volume=backend)
    fuser.BUCKET_BYTES = 1 << 20  # several buckets
    # overlapping but different views: rank 0 fuses frames 0, 1, 40; rank 1 fuses frames 2, 3, 80
    for i in (0, 1, 40) if rank == 0 else (2, 3, 80):
        d, c, T = s[i]
        backend.integrate(d, c, K, T)
    mine = {tuple(k) for k in backend.dirty_keys()}
    n_shared, n_dirty = fuser.merge_halo()
    h = fuser.last_halo
    np.savez(os.path.join(tmpdir, f"halo{rank}.npz"), mine=np.array(sorted(mine)), shared=h["shared_keys"], action=h["action"],
             payload_bytes=h["payload_bytes"], n_shared=n_shared, n_dirty=n_dirty, dirty_after=len(backend.dirty_keys()))
    # the volume is still complete as the sum over ranks: gather and save on the root
    n = fuser.gather_to_root(root=0)
    if rank == 0:
        keys = np.array(sorted(backend.imported))
        np.savez(os.path.join(tmpdir, "gathered.npz"), keys=keys, payload=backend.export_numerators(keys), n=n)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_halo_merge_reduces_only_shared_units(tmp_path):
    """merge_halo over gloo: the plan's shared set is exactly the intersection of the two ranks' dirty lists, only those
    units travel (payload bytes = shared x 81 920), the lowest rank keeps a shared unit and the other zeroes it, and the
    gathered volume afterwards equals fusing all six frames in one volume."""
    import torch.multiprocessing as mp

    import oracle
    from pyslam_amd.synthetic import SyntheticRGBD

    port = 29500 + ((os.getpid() + 777) % 2000)
    mp.spawn(_halo_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    h0, h1 = np.load(tmp_path / "halo0.npz"), np.load(tmp_path / "halo1.npz")
    a, b = {tuple(k) for k in h0["mine"]}, {tuple(k) for k in h1["mine"]}
    inter = sorted(a & b)
    assert 0 < len(inter) < min(len(a), len(b))  # a genuine halo: some, not all
    for h in (h0, h1):
        assert [tuple(k) for k in h["shared"]] == inter          # same plan on both ranks, sorted
        assert int(h["payload_bytes"]) == len(inter) * 4096 * 5 * 4  # nothing but shared units was reduced
        assert int(h["n_shared"]) == len(inter) and int(h["dirty_after"]) == 0
    assert (h0["action"] == 1).all() and (h1["action"] == 2).all()  # lowest listing rank keeps, the other zeroes
    z = np.load(tmp_path / "gathered.npz")
    s = SyntheticRGBD("tiny_160x120_2cm")
    K = np.array(s.intrinsics)
    full = oracle.PortTsdf(0.02, 0.08)
    for i in (0, 1, 40, 2, 3, 80):
        d, c, T = s[i]
        full.integrate(d, c, K, T, 1.0, 4.0)
    k, tsdf, w, col = full.dump()
    np.testing.assert_array_equal(z["keys"], k)
    p = z["payload"]
    np.testing.assert_array_equal(p[..., 1], w)
    ww = np.maximum(w, 1)
    assert np.abs(p[..., 0] / ww - tsdf).max() < 1e-4
    assert np.abs(p[..., 2:5] / ww[..., None] - col).max() < 1e-2  # 0..255 scale, float32 numerators


def test_merge_halo_plan_three_ranks():
    """hv_merge_halo_plan (host-only C): keys listed by >= 2 ranks, sorted; the lowest listing rank keeps, EVERY other rank
    zeroes (a rank that did not list the key may still hold a copy from an earlier window, and pack exports it)."""
    import ctypes

    from pyslam_amd import _lib as L

    lib = L.load()
    lists = [np.array([[0, 0, 0], [1, 0, 0], [5, 5, 5]], np.int32), np.array([[1, 0, 0], [2, 2, 2], [-3, 0, 1]], np.int32),
             np.array([[-3, 0, 1], [1, 0, 0], [9, 9, 9]], np.int32)]
    counts = np.array([len(x) for x in lists], np.int64)
    gathered = np.ascontiguousarray(np.concatenate(lists))
    want = {0: [2, 1], 1: [1, 2], 2: [2, 2]}  # shared (sorted): [-3,0,1] by ranks 1,2; [1,0,0] by ranks 0,1,2
    for rank in range(3):
        n = ctypes.c_int64()
        shared, action = np.zeros((4, 3), np.int32), np.zeros(4, np.uint8)
        L.check(lib.hv_merge_halo_plan(L.ptr(gathered), L.ptr(counts), 3, rank, L.ptr(shared), L.ptr(action), 4, ctypes.byref(n)))
        assert n.value == 2 and shared[:2].tolist() == [[-3, 0, 1], [1, 0, 0]]
        assert action[:2].tolist() == want[rank]


def test_merge_halo_plan_held_three_ranks():
    """hv_merge_halo_plan_held: merged = updated by some rank AND held by two or more; the lowest HOLDER keeps."""
    import ctypes

    from pyslam_amd import _lib as L

    lib = L.load()
    dirty = [np.array([[1, 0, 0]], np.int32), np.array([[1, 0, 0], [2, 2, 2], [7, 7, 7]], np.int32), np.zeros((0, 3), np.int32)]
    held = [np.array([[1, 0, 0], [4, 4, 4]], np.int32), np.array([[1, 0, 0], [2, 2, 2], [7, 7, 7]], np.int32),
            np.array([[1, 0, 0], [2, 2, 2], [4, 4, 4]], np.int32)]
    # [1,0,0]: dirty on 0 and 1, held by all -> rank 0 keeps, 1 and 2 zero (2 holds a copy it did not update)
    # [2,2,2]: dirty on 1 only, held by 1 and 2 -> merged, rank 1 keeps;  [4,4,4]: held by 0 and 2 but nobody updated it -> not merged
    # [7,7,7]: dirty on 1, held by 1 alone -> not merged
    dk, hk = np.ascontiguousarray(np.concatenate(dirty)), np.ascontiguousarray(np.concatenate(held))
    dc, hc = np.array([len(x) for x in dirty], np.int64), np.array([len(x) for x in held], np.int64)
    want = {0: [1, 2], 1: [2, 1], 2: [2, 2]}
    for rank in range(3):
        n = ctypes.c_int64()
        shared, action = np.zeros((4, 3), np.int32), np.zeros(4, np.uint8)
        L.check(lib.hv_merge_halo_plan_held(L.ptr(dk), L.ptr(dc), L.ptr(hk), L.ptr(hc), 3, rank, L.ptr(shared), L.ptr(action), 4,
                                            ctypes.byref(n)))
        assert n.value == 2 and shared[:2].tolist() == [[1, 0, 0], [2, 2, 2]]
        assert action[:2].tolist() == want[rank]


def _three_rank_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from pyslam_amd.distributed import TileShardedTSDF
    from pyslam_amd.synthetic import SyntheticRGBD

    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = SyntheticRGBD("tiny_160x120_2cm")
    K = np.array(s.intrinsics)
    backend = OracleBackedVolume()
    fuser = TileShardedTSDF(0.02, 0.08, s.width, s.height, rank=rank, world_size=world, volume=backend)
    # window 1: only rank 2 fuses (frames 0, 1): nothing is shared, nothing travels, rank 2 keeps its units
    if rank == 2:
        for i in (0, 1):
            backend.integrate(*s[i][:2], K, s[i][2])
    w1 = fuser.merge_halo()
    # window 2: ranks 0 and 1 fuse overlapping views of the same place; rank 2 holds those units from window 1 WITHOUT having
    # updated them now - its copies enter the reduced sum, so it must zero them too (A + B + 2C otherwise)
    if rank == 0:
        backend.integrate(*s[2][:2], K, s[2][2])
    if rank == 1:
        backend.integrate(*s[3][:2], K, s[3][2])
    w2 = fuser.merge_halo()
    # window 3: rank 1 alone revisits: units it shares with the keeper are consolidated although only one rank is dirty
    if rank == 1:
        backend.integrate(*s[4][:2], K, s[4][2])
    w3 = fuser.merge_halo()
    complete = None
    if rank == 0:  # after the merges rank 0 (lowest holder of everything it touches) holds COMPLETE units, without a gather
        keys = backend.unit_keys()
        complete = backend.export_numerators(keys)
        np.savez(os.path.join(tmpdir, "rank0_units.npz"), keys=keys, payload=complete)
    n = fuser.gather_to_root(root=0)
    if rank == 0:
        keys = np.array(sorted(backend.imported))
        np.savez(os.path.join(tmpdir, "gathered3.npz"), keys=keys, payload=backend.export_numerators(keys), n=n,
                 w=np.array([w1[0], w2[0], w3[0]]))
    dist.barrier()
    dist.destroy_process_group()


def test_three_rank_halo_merge_with_a_holder_that_is_not_dirty(tmp_path):
    """World size 3 (where the round-2 plan double-counted): a rank that holds a unit from an earlier window but did not
    update it in this one takes part in the merge as a zeroer; a unit updated by a single rank in a window is still
    consolidated when another rank holds it.  The gathered volume equals one volume fusing all five frames, and the
    units rank 0 holds after the merges are already complete there."""
    import torch.multiprocessing as mp

    import oracle
    from pyslam_amd.synthetic import SyntheticRGBD

    port = 29500 + ((os.getpid() + 1234) % 2000)
    mp.spawn(_three_rank_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    z = np.load(tmp_path / "gathered3.npz")
    assert z["w"][0] == 0 and z["w"][1] > 0 and z["w"][2] > 0
    s = SyntheticRGBD("tiny_160x120_2cm")
    K = np.array(s.intrinsics)
    full = oracle.PortTsdf(0.02, 0.08)
    for i in range(5):
        d, c, T = s[i]
        full.integrate(d, c, K, T, 1.0, 4.0)
    k, tsdf, w, col = full.dump()
    np.testing.assert_array_equal(z["keys"], k)
    p = z["payload"]
    np.testing.assert_array_equal(p[..., 1], w)  # nothing counted twice, nothing lost
    ww = np.maximum(w, 1)
    assert np.abs(p[..., 0] / ww - tsdf).max() < 1e-4
    r0 = np.load(tmp_path / "rank0_units.npz")
    index = {tuple(x): i for i, x in enumerate(k)}
    rows = [index[tuple(x)] for x in r0["keys"]]
    np.testing.assert_array_equal(r0["payload"][..., 1], w[rows])  # complete on rank 0 before any gather


def test_tile_bounds_partition_the_image():
    from pyslam_amd.distributed import tile_bounds, union_keys

    for world in (1, 2, 3, 4, 8):
        tiles = [tile_bounds(r, world, 640, 480) for r in range(world)]
        assert tiles[0][0] == 0 and tiles[-1][2] == 640
        for a, b in zip(tiles, tiles[1:]):
            assert a[2] == b[0]
    u = union_keys([np.array([[1, 2, 3], [0, 0, 0]]), np.zeros((0, 3), np.int32), np.array([[1, 2, 3], [-1, 5, 2]])])
    assert u.tolist() == [[-1, 5, 2], [0, 0, 0], [1, 2, 3]]


class OwnerFilteredOracleGrid:
    """VoxelBlockGrid protocol of ShardedVoxelGrid on top of oracle.PortGrid: set_owner keeps the rank / world, integrate drops the
    points whose block another rank owns - with the LIBRARY's ownership function (hv_block_owner, host code)."""

    def __init__(self, voxel, bs=8):
        import oracle

        self.voxel, self.bs = voxel, bs
        self.grid = oracle.PortGrid(voxel, bs)
        self.rank, self.world = 0, 1

    def set_owner(self, rank, world):
        self.rank, self.world = rank, world

    def integrate(self, points, colors=None):
        import oracle
        from pyslam_amd.distributed import block_owner

        pts = np.ascontiguousarray(points, np.float32)
        mine = block_owner(oracle.keys(pts, self.voxel, self.bs, which="port")[1], self.world) == self.rank
        self.grid.integrate(pts[mine], None if colors is None else np.asarray(colors)[mine])

    def get_voxels(self, min_count=1, min_confidence=0.0):
        import types

        p, c = self.grid.get_voxels(min_count, min_confidence)
        return types.SimpleNamespace(points=p, colors=c)


def _grid_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from oracle import host_prep as hp
    from pyslam_amd.distributed import ShardedVoxelGrid
    from pyslam_amd.synthetic import SyntheticRGBD

    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = SyntheticRGBD("tiny_160x120_2cm")
    sharded = ShardedVoxelGrid(OwnerFilteredOracleGrid(0.02), rank=rank, world_size=world)
    for i in (0, 7, 30):
        d, c, T = s[i]
        pts, cols, _ = hp.frame_to_world_f32(d, c, *s.intrinsics, T, 4.0)
        sharded.integrate(pts, cols)  # every rank sees every point
    local = len(sharded.grid.get_voxels(1).points)
    out = sharded.gather_voxels(min_count=1, root=0)
    np.save(os.path.join(tmpdir, f"grid_local{rank}.npy"), np.array([local, sharded.grid.grid.num_blocks()]))
    if rank == 0:
        np.savez(os.path.join(tmpdir, "grid_gathered.npz"), points=out[0], colors=out[1])
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_voxel_grid_ownership_equals_single_grid(tmp_path):
    """VOXEL_GRID over 2 ranks (gloo): owner(block) = hash(block key) % 2, each rank fuses only its blocks, nothing is reduced;
    the gathered get_voxels rows are EXACTLY the single grid's (sorted rows bit-identical), each rank holds about half the blocks."""
    import torch.multiprocessing as mp

    import oracle
    from oracle import host_prep as hp
    from pyslam_amd.synthetic import SyntheticRGBD
    from tests.conftest import sort_rows

    port = 29500 + ((os.getpid() + 333) % 2000)
    mp.spawn(_grid_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    s = SyntheticRGBD("tiny_160x120_2cm")
    full = oracle.PortGrid(0.02, 8)
    for i in (0, 7, 30):
        d, c, T = s[i]
        pts, cols, _ = hp.frame_to_world_f32(d, c, *s.intrinsics, T, 4.0)
        full.integrate(pts, cols)
    z = np.load(tmp_path / "grid_gathered.npz")
    pa, ca = sort_rows(z["points"], z["colors"])
    pb, cb = sort_rows(*full.get_voxels(1))
    np.testing.assert_array_equal(pa, pb)
    np.testing.assert_array_equal(ca, cb)
    l0, l1 = np.load(tmp_path / "grid_local0.npy"), np.load(tmp_path / "grid_local1.npy")
    assert l0[0] + l1[0] == len(pb) and l0[1] + l1[1] == full.num_blocks()
    assert 0.35 * full.num_blocks() < l0[1] < 0.65 * full.num_blocks()  # hash-balanced


# ---- semantic grids: the association's pair-list exchange ------------------------------------------------------------------------
class _FakeSemanticGrid:
    """What ShardedSemanticGrid needs of a grid: set_owner, the _pair_exchange hook, get_voxels."""

    def __init__(self, rank):
        self.rank, self.owner = rank, None
        self._pair_exchange = None
        self._pair_exchange_device = None

    def set_owner(self, rank, world):
        self.owner = (rank, world)

    def local_pairs(self):
        rng = np.random.default_rng(100 + self.rank)
        n = 0 if self.rank == 1 else 3 + 2 * self.rank  # one rank has nothing in view
        inst = rng.integers(0, 5, n).astype(np.uint64)
        obj = rng.integers(-3, 9, n).astype(np.int32)
        return (inst << np.uint64(32)) | obj.view(np.uint32).astype(np.uint64), rng.integers(1, 50, n).astype(np.int32)

    def get_voxels(self, min_count=1, min_confidence=0.0):
        import types

        n = 4 + self.rank
        return types.SimpleNamespace(points=np.full((n, 3), float(self.rank)), colors=np.full((n, 3), 0.5, np.float32),
                                     class_ids=np.full(n, self.rank, np.int32), object_ids=np.arange(n, dtype=np.int32),
                                     confidences=np.full(n, 0.25, np.float32))


def _sem_worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from pyslam_amd.distributed import ShardedSemanticGrid

    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = _FakeSemanticGrid(rank)
    sh = ShardedSemanticGrid(g, rank=rank, world_size=world)
    assert g.owner == (rank, world) and g._pair_exchange is not None
    keys, counts = g._pair_exchange(*g.local_pairs())  # what assign_object_ids_to_instance_ids calls between vote and decide
    assert keys.dtype == np.uint64 and counts.dtype == np.int32
    np.savez(os.path.join(tmpdir, f"pairs{rank}.npz"), keys=keys, counts=counts, sizes=np.array(sh.last_exchange["sizes"]))
    rows = sh.gather_voxels(1, 0.0, root=0)
    if rank == 0:
        np.savez(os.path.join(tmpdir, "sem_rows.npz"), points=rows[0], cls=rows[2], obj=rows[3], conf=rows[4])
    else:
        assert rows is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_semantic_pair_lists_are_all_gathered_in_rank_order(tmp_path, world):
    """The one collective of the semantic path: every rank ends up with the SAME merged list of the ranks' (instance << 32 | object,
    votes) pairs - equal pairs of different ranks added up (plain concatenation repeated the image's markers N times and
    could overflow the decide stage on 8 ranks), sorted by key, bit for bit (negative object markers included), also when a rank has
    no pairs; get_voxels rows are gathered on the root."""
    import torch.multiprocessing as mp

    port = 29500 + ((os.getpid() + 777 + world) % 2000)
    mp.spawn(_sem_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    cat_k = np.concatenate([_FakeSemanticGrid(r).local_pairs()[0] for r in range(world)])
    cat_c = np.concatenate([_FakeSemanticGrid(r).local_pairs()[1] for r in range(world)])
    merged = {}
    for k, c in zip(cat_k.tolist(), cat_c.tolist()):
        merged[k] = merged.get(k, 0) + c
    want_k = np.array(sorted(merged), np.uint64)
    want_c = np.array([merged[k] for k in sorted(merged)], np.int32)
    assert want_c.sum() == cat_c.sum() and len(want_k) <= len(cat_k)
    for r in range(world):
        z = np.load(tmp_path / f"pairs{r}.npz")
        np.testing.assert_array_equal(z["keys"], want_k)
        np.testing.assert_array_equal(z["counts"], want_c)
        assert z["sizes"].tolist() == [len(_FakeSemanticGrid(q).local_pairs()[0]) for q in range(world)]
    z = np.load(tmp_path / "sem_rows.npz")
    assert len(z["points"]) == sum(4 + r for r in range(world))
    assert sorted(z["cls"].tolist()) == sorted(sum(([r] * (4 + r) for r in range(world)), []))
