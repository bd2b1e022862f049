"""Shared-memory transport of the dense front (VERDICT r03 Weak #7).

The reference moves every keyframe (2.1 MB at 640x480) and every output (a mesh of 100s of MB) through multiprocessing
queues: two pickles and a socket hop per item through the Manager proxy (volumetric_integrator_base.py:401-410,1216-1232,
1320-1342), and `push_to_front` drains and refills the WHOLE queue for every keyframe it adds — the front delivered 58
frames/s in round 3 whatever stood behind it.  Here q_in / q_out / q_management keep their names and semantics, but they only
carry CONTROL: a task's images travel in a ring of fixed-size slots in one POSIX shared-memory segment (`FrameRing`: one
memcpy in add_keyframe, zero-copy numpy views in the worker, which hands the slot pointers straight to
hv_tsdf_integrate_frames; the segment is page-locked with hipHostRegister where a GPU is present so that the H2D DMA reads
the slots in place), and an output's arrays travel in a segment of their own (`export_arrays` / `import_arrays`).
`ControlQueue` is the manager-side queue.Queue of the reference with two more operations that run INSIDE the manager
process — put_front (what push_to_front does with n gets and n + 1 puts) and get_batch (what the TSDF worker's backlog
drain does with one get per task)."""
import mmap
import os
import queue
from multiprocessing import shared_memory
from multiprocessing.managers import SyncManager

import numpy as np

_ALIGN = 256


def _aligned(n):
    return (int(n) + _ALIGN - 1) & ~(_ALIGN - 1)


class ArrayRef:
    """Where an ndarray lives in a shared segment."""

    __slots__ = ("segment", "offset", "shape", "dtype")

    def __init__(self, segment, offset, shape, dtype):
        self.segment, self.offset, self.shape, self.dtype = segment, int(offset), tuple(shape), np.dtype(dtype).str

    def __getstate__(self):
        return (self.segment, self.offset, self.shape, self.dtype)

    def __setstate__(self, s):
        self.segment, self.offset, self.shape, self.dtype = s

    @property
    def nbytes(self):
        return int(np.prod(self.shape, dtype=np.int64)) * np.dtype(self.dtype).itemsize


class FrameRing:
    """n_slots x slot_bytes of POSIX shared memory + one flag per slot (0 free, 1 held).  The producer (SLAM side) acquires a
    slot, copies a keyframe's images into it and sends ArrayRefs; the consumer (worker) builds numpy views, fuses, releases.
    Either process may release (a drained queue drops its tasks' slots)."""

    def __init__(self, ctx, slot_bytes, n_slots):
        self.slot_bytes, self.n_slots = _aligned(slot_bytes), int(n_slots)
        self._shm = shared_memory.SharedMemory(create=True, size=self.slot_bytes * self.n_slots)
        self.name = self._shm.name
        self._flags = ctx.Array("b", self.n_slots, lock=True)
        self._next = ctx.Value("i", 0, lock=False)  # round-robin cursor, guarded by the flags' lock
        self._owner = True
        self._registered = False

    # -- pickling: the child attaches to the same segment ---------------------------------------
    def __getstate__(self):
        return dict(slot_bytes=self.slot_bytes, n_slots=self.n_slots, name=self.name, _flags=self._flags, _next=self._next)

    def __setstate__(self, s):
        self.__dict__.update(s)
        self._shm = shared_memory.SharedMemory(name=self.name)
        self._owner = False
        self._registered = False

    @property
    def buf(self):
        return self._shm.buf

    def base_address(self):
        return np.frombuffer(self._shm.buf, dtype=np.uint8).ctypes.data

    def acquire(self):
        """-> a free slot index, or None.  Slots are handed out ROUND-ROBIN: the slot released last is reused last (with
        lowest-index-first a just-released slot was refilled at once - defence in depth behind the library's own rule that a
        page-locked source has been read when an HV_HOST call returns, hv_h2d)."""
        with self._flags.get_lock():
            start = self._next.value % self.n_slots
            for k in range(self.n_slots):
                i = (start + k) % self.n_slots
                if self._flags[i] == 0:
                    self._flags[i] = 1
                    self._next.value = (i + 1) % self.n_slots
                    return i
        return None

    def release(self, slot):
        if slot is not None and 0 <= int(slot) < self.n_slots:
            self._flags[int(slot)] = 0

    def held(self):
        return sum(1 for i in range(self.n_slots) if self._flags[i] != 0)

    def write(self, slot, arrays):
        """arrays: dict name -> ndarray.  -> dict name -> ArrayRef, or None when they do not fit one slot."""
        need = sum(_aligned(a.nbytes) for a in arrays.values())
        if need > self.slot_bytes:
            return None
        refs, at = {}, slot * self.slot_bytes
        for name, a in arrays.items():
            dst = np.ndarray(a.shape, a.dtype, buffer=self._shm.buf, offset=at)
            np.copyto(dst, a)
            refs[name] = ArrayRef(self.name, at, a.shape, a.dtype)
            at += _aligned(a.nbytes)
        return refs

    def view(self, ref):
        return np.ndarray(ref.shape, np.dtype(ref.dtype), buffer=self._shm.buf, offset=ref.offset)

    def close(self):
        try:
            self._shm.close()
        except Exception:
            pass
        if self._owner:
            try:
                self._shm.unlink()
            except Exception:
                pass


_KEYFRAME_ARRAYS = ("img", "img_right", "depth", "semantic_img", "semantic_instances_img")


def keyframe_to_ring(ring, keyframe_data):
    """Move a keyframe snapshot's images into a ring slot (in place: the fields become ArrayRefs, `_ring_slot` names the slot).
    -> True when moved; False (the snapshot is left as it is and travels pickled, as in the reference) when there is no ring, no
    free slot, or the images do not fit."""
    if ring is None or keyframe_data is None:
        return False
    arrays = {}
    for f in _KEYFRAME_ARRAYS:
        a = getattr(keyframe_data, f, None)
        if isinstance(a, np.ndarray) and a.size > 0:
            arrays[f] = np.ascontiguousarray(a)
    if not arrays:
        return False
    slot = ring.acquire()
    if slot is None:
        return False
    refs = ring.write(slot, arrays)
    if refs is None:
        ring.release(slot)
        return False
    for f, r in refs.items():
        setattr(keyframe_data, f, r)
    keyframe_data._ring_slot = slot
    return True


def keyframe_from_ring(ring, keyframe_data):
    """Worker side: ArrayRef fields become numpy views of the slot (no copy).  -> the slot to release when the keyframe has been
    consumed, or None."""
    slot = getattr(keyframe_data, "_ring_slot", None)
    if slot is None or ring is None:
        return None
    for f in _KEYFRAME_ARRAYS:
        a = getattr(keyframe_data, f, None)
        if isinstance(a, ArrayRef):
            setattr(keyframe_data, f, ring.view(a))
    keyframe_data._ring_slot = None
    return slot


def drop_task(ring, task):
    """A task that leaves a queue without being consumed gives its slot back."""
    kd = getattr(task, "keyframe_data", None)
    slot = getattr(kd, "_ring_slot", None) if kd is not None else None
    if slot is not None and ring is not None:
        ring.release(slot)
        kd._ring_slot = None


# ---- outputs ---------------------------------------------------------------------------------------
_EXPORT_MIN_BYTES = 1 << 16


def _walk(obj, fn, depth=0):
    """Apply fn(container, key, value) to every attribute / list item reachable from obj (outputs are shallow: output ->
    mesh | point cloud | object list -> objects -> box)."""
    if depth > 4 or obj is None:
        return
    if isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            if isinstance(obj, list):
                fn(obj, i, v)
            _walk(obj[i] if isinstance(obj, list) else v, fn, depth + 1)
        return
    d = getattr(obj, "__dict__", None)
    if not isinstance(d, dict):
        return
    for k in list(d.keys()):
        fn(d, k, d[k])
        v = d[k]
        if not isinstance(v, (np.ndarray, ArrayRef, str, bytes, int, float, bool)) and v is not None:
            _walk(v, fn, depth + 1)


def export_arrays(output):
    """Worker side, before q_out.put: every ndarray of at least 64 KB reachable from `output` moves into ONE new shared segment
    (a single memcpy each) and is replaced by an ArrayRef; the segment's name is recorded in output._shm_segment.  The consumer
    calls import_arrays (pop_output does)."""
    found = []

    def collect(container, key, v):
        if isinstance(v, np.ndarray) and v.nbytes >= _EXPORT_MIN_BYTES and v.dtype != object:
            found.append((container, key, v))

    _walk(output, collect)
    if not found:
        return output
    total = sum(_aligned(v.nbytes) for _, _, v in found)
    try:  # a tmpfs that is too small only says so with SIGBUS on the first write
        vfs = os.statvfs("/dev/shm")
        if total > vfs.f_bavail * vfs.f_frsize // 2:
            return output  # travels pickled through the queue, like the reference's
    except OSError:
        pass
    seg = shared_memory.SharedMemory(create=True, size=total)
    at = 0
    for container, key, v in found:
        v = np.ascontiguousarray(v)
        np.copyto(np.ndarray(v.shape, v.dtype, buffer=seg.buf, offset=at), v)
        container[key] = ArrayRef(seg.name, at, v.shape, v.dtype)
        at += _aligned(v.nbytes)
    output._shm_segment = seg.name
    seg.close()
    return output


def import_arrays(output, copy=False):
    """Consumer side: ArrayRefs become numpy arrays over the segment, which is unlinked at once.  copy=False (default): the
    arrays are views of a private mapping of the segment - no memcpy of a mesh of 100s of MB; the mapping lives exactly as long
    as any of the arrays does (each array's base is the mmap).  copy=True: private copies."""
    name = getattr(output, "_shm_segment", None)
    if not name:
        return output
    path = "/dev/shm/" + name.lstrip("/")
    mm = None
    try:
        fd = os.open(path, os.O_RDWR)
        try:
            mm = mmap.mmap(fd, os.fstat(fd).st_size)
        finally:
            os.close(fd)
    except OSError as e:
        # never hand ArrayRefs downstream where ndarrays are expected: the output is unusable without its segment
        raise RuntimeError(f"shared output segment {name!r} cannot be opened ({e}): the output's arrays are lost") from e

    def resolve(container, key, v):
        if isinstance(v, ArrayRef) and v.segment == name:
            a = np.frombuffer(mm, dtype=np.dtype(v.dtype), count=int(np.prod(v.shape, dtype=np.int64)), offset=v.offset).reshape(v.shape)
            container[key] = np.array(a, copy=True) if copy else a

    _walk(output, resolve)
    output._shm_segment = None
    _unlink_segment(name)
    return output


def _unlink_segment(name):
    try:
        os.unlink("/dev/shm/" + name.lstrip("/"))
    except OSError:
        pass
    try:  # the creating process registered the segment with the (shared) resource tracker: it is gone now, on purpose
        from multiprocessing import resource_tracker

        resource_tracker.unregister("/" + name.lstrip("/"), "shared_memory")
    except Exception:
        pass


def drop_output(output):
    """An output that leaves q_out unseen (rebuild / reset drain it) frees its segment."""
    name = getattr(output, "_shm_segment", None)
    if name:
        _unlink_segment(name)
        output._shm_segment = None


# ---- control queue ---------------------------------------------------------------------------------
class ControlQueue(queue.Queue):
    """The manager-side queue of the reference (MultiprocessingManager.Queue = queue.Queue behind a proxy) plus two
    operations that run inside the manager process, one round trip each."""

    def put_front(self, item):
        """data_management.py:94-114 push_to_front: `item` first, then everything that was queued, order kept."""
        with self.not_empty:
            self.queue.appendleft(item)
            self.unfinished_tasks += 1
            self.not_empty.notify()

    def get_batch(self, limit, task_type_name):
        """Pop items from the front while they are tasks of the named type (at most `limit`).  Stops at, and leaves queued,
        the first item that is not (the caller used to take it out and push it back to the front)."""
        out = []
        with self.not_empty:
            while self.queue and len(out) < limit:
                head = self.queue[0]
                if head is None or getattr(getattr(head, "task_type", None), "name", None) != task_type_name:
                    break
                out.append(self.queue.popleft())
            if out:
                self.not_full.notify_all()
        return out

    def drain(self):
        """Everything that is queued, in order (empty_queue in one round trip)."""
        with self.not_empty:
            out = list(self.queue)
            self.queue.clear()
            self.not_full.notify_all()
        return out


class FrontManager(SyncManager):
    pass


FrontManager.register("ControlQueue", ControlQueue)


def start_manager(ctx):
    m = FrontManager(ctx=ctx)
    m.start()
    return m

