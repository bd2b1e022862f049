"""Multi-GPU TSDF fusion: one process per GPU over torch.distributed (backend "nccl" = RCCL on ROCm).

pySLAM itself has no multi-GPU path (no NCCL/MPI/torch.distributed call anywhere); this
is new design.  Every rank sees every posed frame (a 640x480 frame is 2 MB: replicating
it is free next to the ~0.5 GB of voxel traffic it causes).  Two ways to split the work:

``sharding="owner"`` (default, "zero reduce" form) — a unit belongs to rank ``hash(unit index) % N``
    (``hv_tsdf_set_owner``).  A rank claims, stores and sweeps only its own units: per-frame work
    and HBM footprint divide by N, the fused volume is *bit-identical* to a single GPU's and is
    simply distributed over the ranks — no collective while fusing.  ``gather_to_root()`` collects
    it on one rank when a mesh is wanted.

``sharding="tile"`` (north-star form) — rank r fuses only the voxels whose projection falls into its
    vertical image tile (``hv_tsdf_set_tile``); a unit that cannot project into a rank's tile is neither
    allocated nor swept there.  Units on tile borders (and revisits from other viewpoints) then hold
    *partial* running means on several ranks.  ``merge_halo()`` (BASELINE north star: "RCCL
    all-reduce of overlapping-block TSDF/weight") consolidates exactly those:
      all-gather of the key lists of the units each rank stamped since its last merge, and of the units it holds
      (12 B/key) -> ``hv_merge_halo_plan_held``: the keys some rank updated AND two ranks or more hold, identical order
      everywhere (the lowest holding rank keeps, every other holder zeroes - also one that did not update the unit in
      this window: its copy is part of the reduced sum) ->
      ``hv_merge_halo_pack``: additive numerators {sum w*tsdf, w, sum r, sum g, sum b} of THOSE units into one
      dense buffer -> ``all_reduce(SUM)`` in ~64 MB buckets (ring collectives over xGMI are per-link bound,
      ~153 GB/s: few, large messages) -> ``hv_merge_halo_unpack``: the keeper takes the reduced state, the others zero
      theirs and go on fusing deltas.
    The message is shared units x 81 920 B, not the volume.  The sum over ranks of a unit's numerators stays the
    single-GPU total at all times, so ``gather_to_root()`` (union of all keys, sum-reduce to one rank) yields the
    complete volume whenever a mesh is wanted.

Collectives used: all-gather of unit keys, all-reduce (merge_halo) / reduce (gather_to_root) of numerator
buffers — only at merge points, never per frame.  The volume object is duck-typed (unit_keys / dirty_keys /
export_numerators / import_numerators / halo_unpack / mark_merged / reset / set_tile / set_owner) so the
collective logic is exercised on CPU with the gloo backend in tests/.
"""
import numpy as np


def tile_bounds(rank, world_size, width, height):
    """Vertical strip of rank `rank`: (u0, v0, u1, v1)."""
    return (rank * width) // world_size, 0, ((rank + 1) * width) // world_size, height


def union_keys(key_sets):
    """Sorted unique [K,3] int32 union of several [n_i,3] key arrays."""
    allk = np.concatenate([np.asarray(k, dtype=np.int32).reshape(-1, 3) for k in key_sets], axis=0)
    if allk.shape[0] == 0:
        return allk
    return np.unique(allk, axis=0)


class ShardedTSDF:
    BUCKET_BYTES = 64 << 20

    def __init__(self, voxel_length, sdf_trunc, width, height, device=0, max_blocks=None, rank=0, world_size=1,
                 process_group=None, volume=None, group=None, sharding="owner", force_collectives=False):
        """force_collectives: run every collective (and the pack / reduce / unpack around it) even with world_size == 1 - RCCL
        accepts a one-rank group, so the `nccl` branches (CUDA key buffers, CUDA payloads into export_numerators / halo_unpack,
        the ordering of RCCL's stream against the volume's) can be executed on a one-GPU box (they had never run).
        With one rank merge_halo() has no shared unit to find: every dirty unit is then taken through pack -> all-reduce -> unpack
        as its own keeper, which leaves the volume as it was (up to the export / import round trip)."""
        assert sharding in ("owner", "tile")
        self.rank, self.world_size = int(rank), int(world_size)
        self.width, self.height = int(width), int(height)
        self.group = group
        self.sharding = sharding
        self.force_collectives = bool(force_collectives)
        self.distributed = self.world_size > 1 or self.force_collectives
        if volume is None:
            from .volumetric import ScalableTSDFVolume

            volume = ScalableTSDFVolume(voxel_length, sdf_trunc, device=device, max_blocks=max_blocks,
                                        max_points=max(width * height, 1 << 16))
        self.volume = volume
        self.tile = tile_bounds(self.rank, self.world_size, self.width, self.height)
        if self.world_size > 1:
            if sharding == "tile":
                self.volume.set_tile(*self.tile)
            else:
                self.volume.set_owner(self.rank, self.world_size)

    # -- fusion ----------------------------------------------------------------------------------
    def integrate(self, image, intrinsic, extrinsic):
        self.volume.integrate(image, intrinsic, extrinsic)

    def integrate_batch(self, depth, color, intrinsic, extrinsics, depth_scale=1.0, depth_trunc=4.0):
        self.volume.integrate_batch(depth, color, intrinsic, extrinsics, depth_scale, depth_trunc)

    # -- merge -----------------------------------------------------------------------------------
    def _gather_keys(self, keys, dist, torch, dev):
        """All ranks' key lists ([n_i, 3] int32) -> list of host arrays.  The plan that consumes them is host code
        (hv_merge_halo_plan_held), so this is where the keys are meant to end; over RCCL they make ONE device round trip:
        counts (one all-gather), then one [world, cap, 3] buffer gathered in place and downloaded once."""
        n_local = torch.tensor([keys.shape[0]], dtype=torch.int64, device=dev)
        counts_t = torch.zeros(self.world_size, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(counts_t, n_local, group=self.group)
        counts = [int(c) for c in counts_t.cpu().tolist()]
        cap = max(max(counts), 1)
        buf = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
        if keys.shape[0]:
            buf[: keys.shape[0]].copy_(torch.from_numpy(np.ascontiguousarray(keys)), non_blocking=False)
        gathered = torch.empty((self.world_size, cap, 3), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(gathered.view(-1), buf.view(-1), group=self.group)
        host = gathered.cpu().numpy()
        return [host[r, :c] for r, c in enumerate(counts)]

    def merge(self, root=0):
        """Sum-reduce all ranks' volumes into rank `root`; the other ranks are cleared.
        tile sharding: merges partial running means; owner sharding: the sum of disjoint unit sets,
        i.e. a gather.  Returns the number of units on the root afterwards."""
        if not self.distributed:
            return 0
        import torch
        import torch.distributed as dist

        on_gpu = dist.get_backend(self.group) == "nccl"
        dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
        keys = union_keys(self._gather_keys(self.volume.unit_keys(), dist, torch, dev))
        k = keys.shape[0]
        if k == 0:
            return 0
        res3 = self.volume.res ** 3
        units_per_bucket = max(1, self.BUCKET_BYTES // (res3 * 5 * 4))
        for b0 in range(0, k, units_per_bucket):
            sub = keys[b0 : b0 + units_per_bucket]
            # torch.empty: the export kernel (volume's own HIP stream, synchronised on return) writes
            # every element, so no fill kernel on torch's stream can race with it
            payload = torch.empty((sub.shape[0], res3, 5), dtype=torch.float32, device=dev)
            self.volume.export_numerators(sub, out=payload if on_gpu else payload.numpy())
            dist.reduce(payload, dst=root, op=dist.ReduceOp.SUM, group=self.group)
            if on_gpu:
                torch.cuda.current_stream().synchronize()  # RCCL result visible before the import kernel reads it
            if self.rank == root:
                # (owner sharding: the root now also *stores* foreign units; it keeps fusing only its own, the cleared ranks
                # keep fusing theirs, and a later gather sums the deltas onto these.  The first gather reproduces a single
                # GPU's volume up to the export / import round trip (weights and colours exact, tsdf to ~1e-6); later gathers
                # re-sum float32 numerators tsdf*w and are equal to the north-star tolerance 1e-4, not bit for bit.  The
                # import claims the units first and grows the root's pool when they do not fit.)
                self.volume.import_numerators(sub, payload if on_gpu else payload.numpy())
            del payload
        if self.rank != root:
            self.volume.reset()
        return k

    gather_to_root = merge

    def merge_halo(self):
        """All-reduce of the units that two or more ranks updated since their last merge (see the module docstring).
        Returns (shared units, units this rank listed as dirty).  ``last_halo`` keeps the keys / actions of the call
        for inspection (tests assert that nothing but shared units travelled)."""
        if not self.distributed:
            return 0, 0
        import ctypes

        import torch
        import torch.distributed as dist

        from . import _lib as L

        lib = L.load()
        on_gpu = dist.get_backend(self.group) == "nccl"
        dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
        if on_gpu and hasattr(self.volume, "halo_lists_device"):
            return self._merge_halo_device(dist, torch, dev)
        mine = np.ascontiguousarray(self.volume.dirty_keys(), dtype=np.int32).reshape(-1, 3)
        held = np.ascontiguousarray(self.volume.unit_keys(), dtype=np.int32).reshape(-1, 3)

        def gathered(keys):
            lists = self._gather_keys(keys, dist, torch, dev)
            return (np.ascontiguousarray(np.concatenate([x.reshape(-1, 3) for x in lists], axis=0), dtype=np.int32),
                    np.array([len(x) for x in lists], dtype=np.int64))

        (dk, dc), (hk, hc) = gathered(mine), gathered(held)
        n = ctypes.c_int64()
        L.check(lib.hv_merge_halo_plan_held(L.ptr(dk), L.ptr(dc), L.ptr(hk), L.ptr(hc), self.world_size, self.rank, None, None, 0,
                                            ctypes.byref(n)))
        k = n.value
        shared = np.zeros((k, 3), np.int32)
        action = np.zeros(k, np.uint8)
        if k:
            L.check(lib.hv_merge_halo_plan_held(L.ptr(dk), L.ptr(dc), L.ptr(hk), L.ptr(hc), self.world_size, self.rank, L.ptr(shared),
                                                L.ptr(action), k, ctypes.byref(n)))
        if self.force_collectives and self.world_size == 1 and len(mine):
            # one rank has nothing to share: take its dirty units through pack -> all-reduce -> unpack as their own keeper (see __init__)
            shared = np.ascontiguousarray(mine[np.lexsort((mine[:, 2], mine[:, 1], mine[:, 0]))])
            k = len(shared)
            action = np.ones(k, np.uint8)
        self.last_halo = {"shared_keys": shared, "action": action, "dirty": len(mine), "payload_bytes": 0}
        res3 = self.volume.res ** 3
        units_per_bucket = max(1, self.BUCKET_BYTES // (res3 * 5 * 4))
        for b0 in range(0, k, units_per_bucket):
            sub, act = shared[b0 : b0 + units_per_bucket], action[b0 : b0 + units_per_bucket]
            payload = torch.empty((sub.shape[0], res3, 5), dtype=torch.float32, device=dev)
            self.volume.export_numerators(sub, out=payload if on_gpu else payload.numpy())  # == hv_merge_halo_pack
            dist.all_reduce(payload, op=dist.ReduceOp.SUM, group=self.group)
            if on_gpu:
                torch.cuda.current_stream().synchronize()  # RCCL result visible before the unpack kernel reads it
            self.last_halo["payload_bytes"] += payload.numel() * 4
            self.volume.halo_unpack(sub, payload if on_gpu else payload.numpy(), act)
            del payload
        self.volume.mark_merged()
        return k, len(mine)


    def _merge_halo_device(self, dist, torch, dev):
        """merge_halo over RCCL with the key lists and the plan in device memory (hv_halo.hip): the lists are written into torch
        tensors, gathered where they lie, sorted and planned by the library on the device; the shared keys never exist on the host
        (``last_halo`` fetches them on demand).  Three integers cross to the host per merge: the two list lengths (they size the
        gather) and the number of shared units (it sizes the payload)."""
        vol = self.volume
        cap = max(int(vol.num_blocks()), 1)
        dirty = torch.empty(cap, dtype=torch.int64, device=dev)
        held = torch.empty(cap, dtype=torch.int64, device=dev)
        n_dirty, n_held = vol.halo_lists_device(dirty, held)
        counts = torch.tensor([n_dirty, n_held], dtype=torch.int64, device=dev)
        all_counts = torch.zeros((self.world_size, 2), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(all_counts.view(-1), counts, group=self.group)
        all_counts = all_counts.cpu().numpy()  # (2 x world integers: the gather below is sized by them)
        sd, sh = max(int(all_counts[:, 0].max()), 1), max(int(all_counts[:, 1].max()), 1)
        dirty_all = torch.empty((self.world_size, sd), dtype=torch.int64, device=dev)
        held_all = torch.empty((self.world_size, sh), dtype=torch.int64, device=dev)
        pad_d, pad_h = torch.zeros(sd, dtype=torch.int64, device=dev), torch.zeros(sh, dtype=torch.int64, device=dev)
        pad_d[:n_dirty].copy_(dirty[:n_dirty])
        pad_h[:n_held].copy_(held[:n_held])
        dist.all_gather_into_tensor(dirty_all.view(-1), pad_d, group=self.group)
        dist.all_gather_into_tensor(held_all.view(-1), pad_h, group=self.group)
        torch.cuda.current_stream().synchronize()  # RCCL results visible before the plan kernels (the volume's stream) read them
        own_all = self.force_collectives and self.world_size == 1  # one rank: its dirty units through the path as their own keeper
        k = vol.halo_plan_device(dirty_all, np.ascontiguousarray(all_counts[:, 0]), held_all, np.ascontiguousarray(all_counts[:, 1]),
                                 self.world_size, self.rank, all_dirty_kept=own_all)
        self.last_halo = _LazyHalo(vol, n_dirty)
        res3 = vol.res ** 3
        units_per_bucket = max(1, self.BUCKET_BYTES // (res3 * 5 * 4))
        for b0 in range(0, k, units_per_bucket):
            cnt = min(units_per_bucket, k - b0)
            payload = torch.empty((cnt, res3, 5), dtype=torch.float32, device=dev)
            vol.halo_pack_planned(b0, cnt, payload)
            vol.synchronize()  # the pack (volume's stream) before the all-reduce (torch's stream)
            dist.all_reduce(payload, op=dist.ReduceOp.SUM, group=self.group)
            torch.cuda.current_stream().synchronize()  # RCCL result visible before the unpack kernel reads it
            self.last_halo["payload_bytes"] += payload.numel() * 4
            vol.halo_unpack_planned(b0, cnt, payload)
            vol.synchronize()  # (the payload is dropped below)
            del payload
        vol.mark_merged()
        return k, n_dirty


class _LazyHalo(dict):
    """``last_halo`` of a device-planned merge: counters at once, the shared keys / actions fetched from the volume when asked for."""

    def __init__(self, volume, n_dirty):
        super().__init__(dirty=n_dirty, payload_bytes=0)
        self._volume = volume

    def __missing__(self, key):
        if key in ("shared_keys", "action"):
            keys, action = self._volume.halo_plan_fetch()
            self["shared_keys"], self["action"] = keys, action
            return self[key]
        raise KeyError(key)


class TileShardedTSDF(ShardedTSDF):
    """North-star form: image-tile sharding + numerator sum-reduce."""

    def __init__(self, *args, **kwargs):
        kwargs.setdefault("sharding", "tile")
        super().__init__(*args, **kwargs)


def block_owner(block_keys, world_size):
    """Owner rank of each block key [n,3] i32 under the library's ownership function (hv_block_owner, host code)."""
    from . import _lib as L

    keys = np.ascontiguousarray(block_keys, dtype=np.int32).reshape(-1, 3)
    out = np.zeros(len(keys), np.int32)
    L.check(L.load().hv_block_owner(L.ptr(keys), len(keys), int(world_size), L.ptr(out)))
    return out


class ShardedVoxelGrid:
    """VOXEL_GRID mode on N GPUs ("zero reduce" form): every rank sees every frame / point set and fuses only the blocks
    it owns (``hv_set_owner``: owner = hash(block key) % N).  The ranks' voxel sets are disjoint and their union is the single-GPU
    grid bit for bit, so there is no collective while fusing; ``gather_voxels()`` collects what ``get_voxels`` returns on one rank
    (all-gather of the row counts, then of the padded rows).  The grid object is duck-typed (set_owner / integrate* / get_voxels),
    so the gather runs on CPU over gloo in tests/."""

    def __init__(self, grid, rank=0, world_size=1, group=None, force_collectives=False):
        self.grid, self.rank, self.world_size, self.group = grid, int(rank), int(world_size), group
        self.force_collectives = bool(force_collectives)  # run the all-gathers with one rank too (ShardedTSDF.__init__)
        if self.world_size > 1:
            self.grid.set_owner(self.rank, self.world_size)

    def integrate(self, points, colors=None):
        self.grid.integrate(points, colors)

    def integrate_rgbd(self, *args, **kwargs):
        self.grid.integrate_rgbd(*args, **kwargs)

    def gather_voxels(self, min_count=1, min_confidence=0.0, root=0):
        """-> (points [M,3] f32, colors [M,3] f32) of the whole distributed grid on `root`, None elsewhere."""
        v = self.grid.get_voxels(min_count, min_confidence)
        pts, cols = np.ascontiguousarray(v.points, np.float32), np.ascontiguousarray(v.colors, np.float32)
        if self.world_size == 1 and not self.force_collectives:
            return pts, cols
        import torch
        import torch.distributed as dist

        on_gpu = dist.get_backend(self.group) == "nccl"
        dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
        n_local = torch.tensor([len(pts)], dtype=torch.int64, device=dev)
        counts = [torch.zeros_like(n_local) for _ in range(self.world_size)]
        dist.all_gather(counts, n_local, group=self.group)
        counts = [int(c.item()) for c in counts]
        cap = max(max(counts), 1)
        buf = torch.zeros((cap, 6), dtype=torch.float32, device=dev)
        if len(pts):
            buf[: len(pts), :3] = torch.from_numpy(pts).to(dev)
            buf[: len(pts), 3:] = torch.from_numpy(cols).to(dev)
        gathered = [torch.zeros_like(buf) for _ in range(self.world_size)]
        dist.all_gather(gathered, buf, group=self.group)
        if self.rank != root:
            return None
        rows = np.concatenate([g[:c].cpu().numpy() for g, c in zip(gathered, counts)], axis=0)
        return rows[:, :3], rows[:, 3:]


class ShardedSemanticGrid:
    """The semantic block grids on N GPUs (BASELINE configs[4]: "2 mm TSDF + semantic labels, 8 x MI355X").  Block ownership as for the
    VOXEL_GRID (``hv_set_owner``: a rank fuses and stores the blocks owner(block key) == rank; the union of the ranks' grids is the
    single grid bit for bit) - plus the ONE real exchange step of the semantic path: the per-keyframe association
    (assign_object_ids_to_instance_ids, voxel_semantic_data_association.h:70-373) is a global vote.  Every rank votes with the voxels
    it owns (``hv_assoc_vote``), the compacted (instance, object, votes) pair lists - a few hundred bytes - are all-gathered, every
    rank decides on the concatenation with the reference's rules (``hv_assoc_decide``: the rules kernel adds the counts of equal
    pairs) and so arrives at the same map and the same new object ids (the process-wide counters advance in lock step).
    The grid is duck-typed (set_owner / _pair_exchange / integrate* / get_voxels ...): the exchange runs on CPU over gloo in tests/."""

    def __init__(self, grid, rank=0, world_size=1, group=None, force_collectives=False):
        self.grid, self.rank, self.world_size, self.group = grid, int(rank), int(world_size), group
        self.force_collectives = bool(force_collectives)  # run the exchange with one rank too (ShardedTSDF.__init__)
        self.last_exchange = None
        if self.world_size > 1:
            self.grid.set_owner(self.rank, self.world_size)
        if self.world_size > 1 or self.force_collectives:
            import torch.distributed as dist

            if dist.is_initialized() and dist.get_backend(self.group) == "nccl":
                self.grid._pair_exchange_device = self.all_gather_pairs_device
            else:
                self.grid._pair_exchange = self.all_gather_pairs

    def all_gather_pairs_device(self, grid):
        """RCCL form of the exchange: export -> all_gather_into_tensor -> import, the lists never leave the GPU and the host never
        waits (round 4's form did device -> host -> device -> all-gather -> host -> device per keyframe).  One message per rank:
        int64 [1 + 2 cap] = [n, keys, votes]; equal pairs of different ranks are added up on the way in."""
        import torch
        import torch.distributed as dist

        dev = torch.device("cuda", torch.cuda.current_device())
        words = 1 + 2 * grid.ASSOC_PAIRS_CAP
        if getattr(self, "_msg", None) is None or self._msg.device != dev:
            self._msg = torch.empty(words, dtype=torch.int64, device=dev)
            self._msgs = torch.empty(words * self.world_size, dtype=torch.int64, device=dev)
        grid.assoc_pairs_export(self._msg)
        dist.all_gather_into_tensor(self._msgs, self._msg, group=self.group)  # (RCCL's stream is ordered against torch's current one)
        grid.assoc_pairs_import(self._msgs, self.world_size)
        self.last_exchange = {"sizes": None, "bytes": int(words * 8 * self.world_size), "device_resident": True}

    def all_gather_pairs(self, keys, counts):
        """-> the ranks' (keys u64, votes i32) lists merged - identical on every rank: equal pairs (the image's 'instance seen' markers
        are contributed by every rank, and an object's voxels live on several) are added up, so the merged list is no longer than a
        single GPU's would be and the decide stage's 4096-pair limit is not reached by repetition."""
        import torch
        import torch.distributed as dist

        on_gpu = dist.get_backend(self.group) == "nccl"
        dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
        keys = np.ascontiguousarray(keys, np.uint64)
        counts = np.ascontiguousarray(counts, np.int32)
        n_local = torch.tensor([len(keys)], dtype=torch.int64, device=dev)
        sizes = [torch.zeros_like(n_local) for _ in range(self.world_size)]
        dist.all_gather(sizes, n_local, group=self.group)
        sizes = [int(x.item()) for x in sizes]
        cap = max(max(sizes), 1)
        # one message per rank: [cap] int64 keys followed by [cap] votes widened to int64
        buf = torch.zeros(2 * cap, dtype=torch.int64, device=dev)
        if len(keys):
            buf[: len(keys)] = torch.from_numpy(keys.view(np.int64)).to(dev)
            buf[cap : cap + len(keys)] = torch.from_numpy(counts.astype(np.int64)).to(dev)
        gathered = [torch.zeros_like(buf) for _ in range(self.world_size)]
        dist.all_gather(gathered, buf, group=self.group)
        ks, cs = [], []
        for g, m in zip(gathered, sizes):
            g = g.cpu().numpy()
            ks.append(g[:m].view(np.uint64))
            cs.append(g[cap : cap + m].astype(np.int32))
        all_k, all_c = np.concatenate(ks), np.concatenate(cs)
        uk, inv = np.unique(all_k, return_inverse=True)  # (sorted: the same order on every rank)
        uc = np.zeros(len(uk), np.int64)
        np.add.at(uc, inv, all_c.astype(np.int64))
        self.last_exchange = {"sizes": sizes, "bytes": int(2 * cap * 8 * self.world_size), "merged": int(len(uk))}
        return uk, uc.astype(np.int32)

    def __getattr__(self, name):  # integrate, integrate_rgbd, assign_object_ids_to_instance_ids, remap_instance_ids, get_voxels, ...
        return getattr(self.grid, name)

    def gather_voxels(self, min_count=1, min_confidence=0.0, root=0):
        """-> (points f64 [M,3], colors f32, class_ids, object_ids, confidences) of the whole distributed grid on `root`, None elsewhere."""
        v = self.grid.get_voxels(min_count, min_confidence)
        rows = np.concatenate([np.asarray(v.points, np.float64), np.asarray(v.colors, np.float64), np.asarray(v.class_ids, np.float64)[:, None],
                               np.asarray(v.object_ids, np.float64)[:, None], np.asarray(v.confidences, np.float64)[:, None]], axis=1) \
            if len(v.points) else np.zeros((0, 9), np.float64)
        if self.world_size > 1 or self.force_collectives:
            import torch
            import torch.distributed as dist

            on_gpu = dist.get_backend(self.group) == "nccl"
            dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
            n_local = torch.tensor([len(rows)], dtype=torch.int64, device=dev)
            sizes = [torch.zeros_like(n_local) for _ in range(self.world_size)]
            dist.all_gather(sizes, n_local, group=self.group)
            sizes = [int(x.item()) for x in sizes]
            cap = max(max(sizes), 1)
            buf = torch.zeros((cap, 9), dtype=torch.float64, device=dev)
            if len(rows):
                buf[: len(rows)] = torch.from_numpy(rows).to(dev)
            gathered = [torch.zeros_like(buf) for _ in range(self.world_size)]
            dist.all_gather(gathered, buf, group=self.group)
            if self.rank != root:
                return None
            rows = np.concatenate([g[:m].cpu().numpy() for g, m in zip(gathered, sizes)], axis=0)
        return (rows[:, 0:3], rows[:, 3:6].astype(np.float32), rows[:, 6].astype(np.int32), rows[:, 7].astype(np.int32),
                rows[:, 8].astype(np.float32))
