"""Mirror of pyslam/dense/volumetric_integrator_base.py: task/queue protocol, worker process loop,
keyframe gating, output wrappers (reference lines cited per item).

Differences that are deliberate (MI355X-first):
* the worker is always a *spawned* process (the reference already spawns for TSDF, base.py:348-362)
  and owns the HIP context; the volume stays resident in HBM for the life of the worker;
* q_in / q_out / q_management keep their names and semantics (push-to-front for regular tasks, RESET on
  q_management; q_in and q_management are manager queues like the reference's, because drain-and-refill needs put()
  to be synchronous), but they carry CONTROL only: a keyframe's images travel in a shared-memory ring of slots
  (one memcpy in add_keyframe, zero-copy views in the worker, page-locked for the H2D DMA), an output's arrays in a
  shared segment of their own, and push_to_front / the backlog drain are ONE manager round trip each
  (shared_transport.py; `kVolumetricIntegrationUseSharedMemory = False` restores pickled images);
* undistortion: the maps are computed by a restatement of OpenCV's getOptimalNewCameraMatrix /
  initUndistortRectifyMap (pyslam_amd/prep.py) and cv2.remap runs on the GPU (hv_remap); cv2 is not
  available here, so parity with OpenCV at that step is unpinned (validated geometrically).
"""
import multiprocessing as std_mp
import os
import signal
import threading
import time
import traceback
from collections import deque
from enum import Enum

import numpy as np

from . import shared_transport as st
from .parameters import get_parameters, static_fields_to_dict
from .ply_io import write_ply_mesh, write_ply_points

Parameters = get_parameters()

kVerbose = False
kVolumetricIntegratorProcessName = "VolumetricIntegratorProcess"  # base.py:88
kLogFile = "logs/volumetric_integrator.log"  # base.py:462


class VolumetricIntegrationTaskType(Enum):  # base.py:91-97
    NONE = 0
    INTEGRATE = 1
    SAVE = 2
    LOAD = 3
    RESET = 4
    UPDATE_OUTPUT = 5


class VolumetricIntegrationKeyframeData:  # base.py:101-137
    """Picklable snapshot of the keyframe fields the dense path consumes."""

    def __init__(self, keyframe, img=None, img_right=None, depth=None, semantic_img=None, semantic_instances_img=None):
        self.id = keyframe.id if keyframe is not None else -1
        self.kid = keyframe.kid if keyframe is not None else -1
        self.img_id = keyframe.img_id if keyframe is not None else -1
        self.timestamp = keyframe.timestamp if keyframe is not None else -1
        self.pose = keyframe.pose() if keyframe is not None else None  # Tcw
        self.camera = keyframe.camera if keyframe is not None else None
        self.img = img if img is not None else (keyframe.img if keyframe is not None else None)
        self.img_right = img_right if img_right is not None else (getattr(keyframe, "img_right", None) if keyframe is not None else None)
        self.depth = depth if depth is not None else (keyframe.depth_img if keyframe is not None else None)
        self.semantic_img = semantic_img if semantic_img is not None else (getattr(keyframe, "semantic_img", None) if keyframe is not None else None)
        self.semantic_instances_img = (
            semantic_instances_img if semantic_instances_img is not None
            else (getattr(keyframe, "semantic_instances_img", None) if keyframe is not None else None)
        )


class VolumetricIntegrationTask:  # base.py:140-156
    def __init__(self, keyframe=None, img=None, img_right=None, depth=None, semantic_img=None,
                 task_type=VolumetricIntegrationTaskType.NONE, load_save_path=None):
        self.task_type = task_type
        self.keyframe_data = VolumetricIntegrationKeyframeData(keyframe, img, img_right, depth, semantic_img)
        self.load_save_path = load_save_path


class VolumetricIntegrationPointCloud:  # base.py:159-206
    def __init__(self, point_cloud=None, points=None, colors=None, semantics=None, object_ids=None,
                 semantic_colors=None, object_colors=None):
        if point_cloud is not None:
            self.points = np.asarray(point_cloud.points)
            self.colors = np.asarray(point_cloud.colors)
            self.semantics = self.object_ids = self.semantic_colors = self.object_colors = None
        else:
            self.points = np.asarray(points) if points is not None else None
            self.colors = np.asarray(colors) if colors is not None else None
            self.semantics = np.asarray(semantics) if semantics is not None else None
            self.object_ids = np.asarray(object_ids) if object_ids is not None else None
            self.semantic_colors = np.asarray(semantic_colors) if semantic_colors is not None else None
            self.object_colors = np.asarray(object_colors) if object_colors is not None else None

    def to_o3d(self):
        import open3d as o3d

        pc = o3d.geometry.PointCloud()
        pc.points = o3d.utility.Vector3dVector(self.points)
        pc.colors = o3d.utility.Vector3dVector(self.colors)
        return pc


class VolumetricIntegrationMesh:  # base.py:209-226
    def __init__(self, mesh):
        self.vertices = np.asarray(mesh.vertices)
        self.triangles = np.asarray(mesh.triangles)
        self.vertex_colors = np.asarray(mesh.vertex_colors)
        self.vertex_normals = np.asarray(mesh.vertex_normals)

    def to_o3d(self):
        import open3d as o3d

        mesh = o3d.geometry.TriangleMesh()
        mesh.vertices = o3d.utility.Vector3dVector(self.vertices)
        mesh.triangles = o3d.utility.Vector3iVector(self.triangles)
        mesh.vertex_colors = o3d.utility.Vector3dVector(self.vertex_colors)
        mesh.vertex_normals = o3d.utility.Vector3dVector(self.vertex_normals)
        return mesh


class VolumetricIntegrationOutput:  # base.py:308-322
    def __init__(self, task_type, id=-1, point_cloud=None, mesh=None, objects=None):
        self.task_type = task_type
        self.id = id
        self.point_cloud = point_cloud
        self.mesh = mesh
        self.objects = objects
        self.timestamp = time.perf_counter()


def push_to_front(queue, item):
    """pyslam/utilities/data_management.py:94-114: drain, then refill with `item` first.  A ControlQueue does exactly that
    inside the manager process (one round trip instead of 2 n + 1)."""
    if hasattr(queue, "put_front"):
        queue.put_front(item)
        return
    items = [item]
    while True:
        try:
            items.append(queue.get(block=False))
        except Exception:
            break
    for i in items:
        try:
            queue.put(i, block=False)
        except Exception:
            try:
                queue.put(i, timeout=1.0)
            except Exception:
                pass


def empty_queue(queue, on_drop=None):
    """Drop everything that is queued; on_drop(item) sees every dropped item (a task gives its ring slot back, an output its
    shared segment)."""
    if hasattr(queue, "drain"):
        try:
            items = queue.drain()
        except Exception:
            items = []
        if on_drop is not None:
            for i in items:
                on_drop(i)
        return
    try:
        while True:
            item = queue.get(block=False)
            if on_drop is not None:
                on_drop(item)
    except Exception:
        pass


def take_integrate_backlog(q_in, limit):
    """The queued INTEGRATE tasks at the front of q_in (at most `limit`), in queue order; the first task of another kind stays
    queued where it was.  One manager round trip on a ControlQueue."""
    if hasattr(q_in, "get_batch"):
        return q_in.get_batch(limit, VolumetricIntegrationTaskType.INTEGRATE.name)
    out = []
    while len(out) < limit:
        try:
            nxt = q_in.get_nowait()
        except Exception:
            break
        if nxt is not None and nxt.task_type == VolumetricIntegrationTaskType.INTEGRATE:
            out.append(nxt)
        else:
            push_to_front(q_in, nxt)  # not ours: put it back where it was
            break
    return out


class _WorkerQueue:
    """q_in as the worker's volume_integration() sees it: the same queue, but a task that comes out has its images resolved
    from the shared ring (numpy views, no copy) and its slot noted; the worker loop releases the noted slots when the call that
    took them has returned."""

    def __init__(self, q, ring):
        self._q, self._ring, self._slots = q, ring, []

    def _take(self, task):
        kd = getattr(task, "keyframe_data", None)
        if kd is not None:
            slot = st.keyframe_from_ring(self._ring, kd)
            if slot is not None:
                self._slots.append(slot)
        return task

    def get(self, *a, **k):
        return self._take(self._q.get(*a, **k))

    def get_nowait(self):
        return self._take(self._q.get_nowait())

    def get_batch(self, limit, task_type_name):
        """-> list of queued tasks of that type from the front of the queue (possibly empty)."""
        if hasattr(self._q, "get_batch"):
            return [self._take(t) for t in self._q.get_batch(limit, task_type_name)]
        out = []
        while len(out) < limit:
            try:
                nxt = self._q.get_nowait()
            except Exception:
                break
            if nxt is not None and getattr(getattr(nxt, "task_type", None), "name", None) == task_type_name:
                out.append(self._take(nxt))
            else:
                push_to_front(self._q, nxt)  # not ours: put it back where it was
                break
        return out

    def release_consumed(self):
        for slot in self._slots:
            self._ring.release(slot)
        self._slots = []

    def __getattr__(self, name):  # put, put_front, empty, qsize, drain, ...
        return getattr(self._q, name)


def _shallow_output_copy(output):
    """A copy of an output whose containers (output, mesh / point cloud / object list, objects) are new objects sharing the
    arrays: export_arrays() rewrites the COPY's fields, the worker's last_output stays whole."""
    import copy

    def cp(o, depth=0):
        if o is None or depth > 4 or isinstance(o, (np.ndarray, str, bytes, int, float, bool, Enum)):
            return o
        if isinstance(o, list):
            return [cp(x, depth + 1) for x in o]
        if hasattr(o, "__dict__"):
            c = copy.copy(o)
            for k, v in list(c.__dict__.items()):
                c.__dict__[k] = cp(v, depth + 1)
            return c
        return o

    return cp(output)


class TimerFps:
    """pyslam/utilities/timer.py:75-94: moving average over the last 10 intervals."""

    def __init__(self, name="", average_width=10):
        self.name = name
        self.times = deque(maxlen=average_width)
        self.last = None
        self.last_elapsed = 0.0

    def start(self):
        self.last = time.perf_counter()

    def refresh(self):
        now = time.perf_counter()
        if self.last is not None:
            self.last_elapsed = now - self.last
            self.times.append(self.last_elapsed)
        self.last = now

    def get_fps(self):
        if not self.times:
            return 0.0
        mean = sum(self.times) / len(self.times)
        return 1.0 / mean if mean > 0 else 0.0


class _RepeatingTimer:
    """SimpleTaskTimer(interval, callback, single_shot=False) of pyslam/utilities/timer.py."""

    def __init__(self, interval, callback, name="timer"):
        self.interval, self.callback, self.name = interval, callback, name
        self._stop = threading.Event()
        self._thread = None

    def start(self):
        self._thread = threading.Thread(target=self._loop, name=self.name, daemon=True)
        self._thread.start()

    def _loop(self):
        while not self._stop.wait(self.interval):
            try:
                self.callback()
            except Exception:
                traceback.print_exc()

    def stop(self):
        self._stop.set()


class VolumetricIntegratorBase:
    """pyslam/dense/volumetric_integrator_base.py:328-1394 (public protocol)."""

    print = staticmethod(lambda *a, **k: print(*a, **k) if kVerbose else None)

    def __init__(self, camera, environment_type, sensor_type, volumetric_integrator_type, viewer_queue=None, **kwargs):
        self.volumetric_integrator_type = volumetric_integrator_type
        self.constructor_kwargs = kwargs
        self.mp = std_mp.get_context("spawn")  # base.py:348-362: spawn is required once a GPU context is involved
        mp = self.mp
        self.camera = camera
        self.environment_type = environment_type
        self.sensor_type = sensor_type
        self.viewer_queue = viewer_queue

        self.keyframe_queue_timer = _RepeatingTimer(0.5, self.flush_keyframe_queue, "KeyframeQueueTimer")  # base.py:369-374
        self.keyframe_queue_lock = threading.Lock()
        self.keyframe_queue = deque()

        self.volume = None
        self.time_volumetric_integration = mp.Value("d", 0.0)  # base.py:383
        self.last_input_task = None
        self.last_output = None
        self.last_integrated_id = -1

        self.reset_mutex = mp.Lock()
        self.reset_requested = mp.Value("i", -1)
        self.load_request_completed = mp.Value("i", -1)
        self.load_request_condition = mp.Condition()
        self.save_request_completed = mp.Value("i", -1)
        self.save_request_condition = mp.Condition()

        # q_in / q_management are MANAGER queues as in the reference (base.py:401 MultiprocessingManager): push_to_front,
        # empty_queue, rebuild and reset_if_requested drain with get(block=False) and expect to see what this very process
        # just put - true for a manager queue (put returns when the item is in the queue), not for mp.Queue, whose feeder
        # thread delivers later (push_to_front then degrades to append, rebuild()'s drain misses in-flight INTEGRATE tasks
        # and their pre-loop-closure poses get fused into the fresh volume).  q_out only ever carries one consumer's
        # outputs (meshes of 100s of MB): it stays a plain mp.Queue to avoid a second pickle of those.
        self._mp_manager = st.start_manager(mp)
        self.q_in = self._mp_manager.ControlQueue()          # regular tasks (integrate, update output, save)
        self.q_out = mp.Queue()                              # outputs (visualise, save)
        self.q_management = self._mp_manager.ControlQueue()  # management tasks (reset / rebuild)
        # keyframe images travel in shared memory, not through the queue: slot = colour + right image + depth + two label images
        self.frame_ring = None
        if getattr(Parameters, "kVolumetricIntegrationUseSharedMemory", True) and camera is not None:
            try:
                px = int(camera.width) * int(camera.height)
                slot_bytes = px * (3 + 3 + 4 + 4 + 4) + 8 * 256
                slots = int(getattr(Parameters, "kVolumetricIntegrationSharedMemorySlots", 96))
                try:  # a tmpfs that is too small only says so with SIGBUS on the first write: size the ring to what is free
                    vfs = os.statvfs("/dev/shm")
                    slots = min(slots, int(vfs.f_bavail * vfs.f_frsize // 4 // slot_bytes))
                except OSError:
                    pass
                if slots >= 2:
                    self.frame_ring = st.FrameRing(mp, slot_bytes, slots)
            except Exception:
                traceback.print_exc()
                self.frame_ring = None
        self.parameters_dict = static_fields_to_dict(Parameters)  # snapshot for the child, base.py:412-416
        self.q_in_condition = mp.Condition()
        self.q_out_condition = mp.Condition()
        self.is_running = mp.Value("i", 0)
        self.is_looping = mp.Value("i", 0)
        self.process = None
        self.start()

    def is_ready(self):  # base.py:451
        return self.is_running.value == 1 and self.is_looping.value == 1

    # -- pickling for the spawned child (base.py:495-526): drop what must not cross --------------
    def __getstate__(self):
        state = self.__dict__.copy()
        for k in ("keyframe_queue_timer", "keyframe_queue_lock", "keyframe_queue", "process", "mp", "volume", "_mp_manager"):
            state.pop(k, None)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.volume = None

    def start(self):  # base.py:536-568
        self.is_running.value = 1
        self.process = self.mp.Process(
            target=self.run,
            args=(self.camera, self.environment_type, self.sensor_type, self.viewer_queue, self.q_in, self.q_in_condition,
                  self.q_out, self.q_out_condition, self.q_management, self.is_running, self.is_looping, self.reset_mutex,
                  self.reset_requested, self.load_request_completed, self.load_request_condition,
                  self.save_request_completed, self.save_request_condition, self.time_volumetric_integration,
                  self.parameters_dict, self.constructor_kwargs),
            name=kVolumetricIntegratorProcessName,
        )
        self.process.daemon = True
        self.process.start()
        self.keyframe_queue_timer.start()

    def _stop_volume_integrator_implementation(self):  # base.py:571
        pass

    # -- caller-side API --------------------------------------------------------------------------
    def save(self, path):  # base.py:574-593
        if self.save_request_completed.value == 0:
            return
        filepath = path + "/dense_map.ply"
        task = VolumetricIntegrationTask(task_type=VolumetricIntegrationTaskType.SAVE, load_save_path=filepath)
        self.save_request_completed.value = 0
        self.add_task(task, front=True)
        with self.save_request_condition:
            while self.save_request_completed.value == 0 and self.is_running.value == 1:
                self.save_request_condition.wait(timeout=1.0)

    def load(self, path):  # base.py:595-604: a stub in the reference too
        return None

    def request_reset(self):  # base.py:606-631
        with self.reset_mutex:
            if self.reset_requested.value == 1:
                return
            self.reset_requested.value = 1
        while self.is_running.value == 1:
            with self.reset_mutex:
                with self.q_in_condition:
                    self.q_in_condition.notify_all()
                if self.reset_requested.value == 0:
                    break
            time.sleep(0.1)
        with self.keyframe_queue_lock:
            self.keyframe_queue.clear()

    def quit(self):  # base.py:652-700
        if self.is_running.value != 1:
            return
        self.is_running.value = 0
        self.keyframe_queue_timer.stop()
        with self.q_in_condition:
            self.q_in.put(None)
            self.q_in_condition.notify_all()
        with self.q_out_condition:
            self.q_out_condition.notify_all()
        self.process.join(timeout=2 * Parameters.kMultiprocessingProcessJoinDefaultTimeout)
        if self.process.is_alive():
            self.process.terminate()
        empty_queue(self.q_out, st.drop_output)
        try:
            self._mp_manager.shutdown()  # the queues' server process
        except Exception:
            pass
        if self.frame_ring is not None:
            self.frame_ring.close()

    def flush_keyframe_queue(self):  # base.py:1120-1188
        with self.keyframe_queue_lock:
            if len(self.keyframe_queue) == 0:
                return
            max_checks = len(self.keyframe_queue)
            processed = 0
            while len(self.keyframe_queue) > 0:
                kf = self.keyframe_queue[0]
                wait_for_semantics = bool(getattr(kf, "wait_for_semantics", False)) and not kf.is_semantics_available()
                if kf.lba_count >= Parameters.kVolumetricIntegrationMinNumLBATimes and not wait_for_semantics:
                    self.keyframe_queue.popleft()
                    processed += 1
                    self.add_task(VolumetricIntegrationTask(kf, task_type=VolumetricIntegrationTaskType.INTEGRATE))
                else:
                    if len(self.keyframe_queue) <= 1:
                        break
                    self.keyframe_queue.rotate(-1)
                    processed += 1
                if processed >= max_checks:
                    break

    def add_keyframe(self, keyframe, img, img_right, depth, print=print):  # base.py:1191-1214
        if (depth is None or depth.size == 0) and not Parameters.kVolumetricIntegrationUseDepthEstimator:
            return
        with self.keyframe_queue_lock:
            self.keyframe_queue.append(keyframe)
        self.flush_keyframe_queue()

    def add_task(self, task, front=True):  # base.py:1216-1232
        if self.is_running.value == 1:
            if task is not None and task.task_type == VolumetricIntegrationTaskType.INTEGRATE and self.frame_ring is not None:
                # one memcpy into a ring slot; the queue carries the slot's name.  A full ring waits briefly for the worker
                # (back-pressure), then falls back to pickling the images with the task like the reference does.
                t0 = time.perf_counter()
                while not st.keyframe_to_ring(self.frame_ring, task.keyframe_data):
                    if self.frame_ring.held() < self.frame_ring.n_slots or time.perf_counter() - t0 > 0.25 or self.is_running.value != 1:
                        break  # not a matter of waiting (nothing to move / does not fit), or waited long enough
                    time.sleep(0.0005)
            with self.q_in_condition:
                if front:
                    push_to_front(self.q_in, task)
                else:
                    self.q_in.put(task, timeout=1.0)
                self.q_in_condition.notify_all()

    def add_update_output_task(self):  # base.py:1234-1240
        if self.is_running.value == 1:
            with self.q_in_condition:
                self.q_in.put(VolumetricIntegrationTask(task_type=VolumetricIntegrationTaskType.UPDATE_OUTPUT))
                self.q_in_condition.notify_all()

    def rebuild(self, map):  # base.py:1242-1318
        if self.is_running.value != 1:
            return
        with self.q_in_condition:
            empty_queue(self.q_in, self._drop_task)
        with self.q_in_condition:
            self.q_management.put(VolumetricIntegrationTask(task_type=VolumetricIntegrationTaskType.RESET), timeout=1.0)
            self.q_in_condition.notify_all()
        wait_start = time.time()
        while self.is_running.value == 1 and not self.q_management.empty():
            if time.time() - wait_start > 5.0:
                break
            time.sleep(0.05)
        with self.q_out_condition:
            empty_queue(self.q_out, st.drop_output)
            self.q_out.put(VolumetricIntegrationOutput(VolumetricIntegrationTaskType.RESET))
            self.q_out_condition.notify_all()
        with self.keyframe_queue_lock:
            self.keyframe_queue.clear()
            for kf in map.keyframes:
                if not kf.is_bad() and kf.lba_count >= Parameters.kVolumetricIntegrationMinNumLBATimes:
                    if kf.depth_img is None:
                        continue
                    self.keyframe_queue.append(kf)
        self.flush_keyframe_queue()

    def pop_output(self, timeout=Parameters.kLoopDetectingTimeoutPopKeyframe):  # base.py:1320-1342
        if self.is_running.value == 0:
            return None
        # The reference waits on q_out_condition while q_out.empty() and then get()s (same outcome: the next output, or None after
        # `timeout`).  Its q_out is a manager queue, where an item is visible the moment put() returns; q_out here is an mp.Queue (a
        # second pickle of a mesh through the manager is what the shared segment avoids), whose feeder thread delivers AFTER put()
        # returns: a consumer woken by the worker's notify_all could find empty() still true, wait again and sleep out the whole
        # timeout with the output already on its way (seen as sporadic +0.5 s ticks in bench.py's front leg).  get(timeout) blocks on
        # the queue's pipe itself and wakes when the item is there.
        try:
            return st.import_arrays(self.q_out.get(timeout=timeout))
        except Exception:  # queue.Empty after `timeout`, or the queue closed by quit()
            return None

    def _drop_task(self, task):
        st.drop_task(self.frame_ring, task)

    def draw_output(self, output):  # base.py:1344: forwards to pySLAM's viewer when it exists
        if self.viewer_queue is not None and output is not None:
            self.viewer_queue.put(output)

    # -- worker side ------------------------------------------------------------------------------
    def init(self, camera, environment_type, sensor_type, parameters_dict, constructor_kwargs):  # base.py:703-786
        for k, v in (parameters_dict or {}).items():  # the snapshot taken in the parent wins
            setattr(Parameters, k, v)
        self.camera = camera
        self.environment_type = environment_type
        self.sensor_type = sensor_type
        self.depth_factor = 1.0  # base.py:713: the factor is already folded into the keyframe depth
        self.last_output = None
        self.last_integrated_id = -1
        # maps to undistort colour / depth / label images (base.py:758-786); the remaps themselves run on
        # the GPU (hv_remap).  OpenCV's getOptimalNewCameraMatrix / initUndistortRectifyMap are restated
        # in pyslam_amd/prep.py (cv2 is not a dependency of this package).
        D = np.asarray(getattr(camera, "D", np.zeros(5)), dtype=np.float64).ravel()
        K = np.array([[camera.fx, 0.0, camera.cx], [0.0, camera.fy, camera.cy], [0.0, 0.0, 1.0]])
        self.new_K, self.calib_map1, self.calib_map2 = K, None, None
        if np.linalg.norm(D) > 1e-10:
            from ..prep import get_optimal_new_camera_matrix, init_undistort_rectify_map

            if getattr(Parameters, "kDepthImageUndistortionUseOptimalNewCameraMatrixWithAlphaScale", True):
                alpha = getattr(Parameters, "kDepthImageUndistortionOptimalNewCameraMatrixWithAlphaScaleValue", 0.7)
                self.new_K, _ = get_optimal_new_camera_matrix(K, D, (camera.width, camera.height), alpha,
                                                              (camera.width, camera.height))
            self.calib_map1, self.calib_map2 = init_undistort_rectify_map(K, D, self.new_K, (camera.width, camera.height))
        self.rectified_fx, self.rectified_fy = float(self.new_K[0, 0]), float(self.new_K[1, 1])
        self.rectified_cx, self.rectified_cy = float(self.new_K[0, 2]), float(self.new_K[1, 2])
        self.dtype_vertices = np.dtype(Parameters.kDenseMappingDtypeVertices)
        self.dtype_colors = np.dtype(Parameters.kDenseMappingDtypeColors)
        self.dtype_depths = np.dtype(Parameters.kDenseMappingDtypeDepth)
        # depth estimator for keyframes without depth (stereo back-end), base.py:719-756.  The factory is injected
        # (constructor kwarg `depth_estimator_factory(camera) -> object with infer(img, img_right)`), or taken from
        # real pySLAM when it is importable; see pyslam_amd/depth_estimation.py for the device-resident wrapper.
        self.depth_estimator = None
        self.img_id_to_depth = {}
        if getattr(Parameters, "kVolumetricIntegrationUseDepthEstimator", False):
            make = constructor_kwargs.get("depth_estimator_factory")
            if make is not None:
                self.depth_estimator = make(camera)
            else:
                try:
                    from pyslam.depth_estimation.depth_estimator_factory import DepthEstimatorType, depth_estimator_factory  # type: ignore

                    self.depth_estimator = depth_estimator_factory(
                        depth_estimator_type=DepthEstimatorType.from_string(Parameters.kVolumetricIntegrationDepthEstimatorType),
                        camera=camera)
                except Exception:
                    self.depth_estimator = None

    def get_camera_intrinsics_for_depth(self):
        return self.rectified_fx, self.rectified_fy, self.rectified_cx, self.rectified_cy

    def estimate_depth_if_needed_and_rectify(self, keyframe_data):  # base.py:969-1062
        """-> (color RGB, depth f32 metres, pts3d, semantic, instances).  With a depth estimator (base.py:981-1004;
        SURVEY 8f N3) the depth may be a torch CUDA tensor that never visits the host: the shadow-point filter,
        the remap and the fusion all take device pointers."""
        color, depth = keyframe_data.img, keyframe_data.depth
        pts3d = None
        if depth is None or getattr(depth, "size", 1) == 0:
            if self.depth_estimator is None:
                return None, None, None, None, None  # skip this keyframe
            if keyframe_data.id in self.img_id_to_depth:
                depth = self.img_id_to_depth[keyframe_data.id]
            else:
                depth, pts3d = self.depth_estimator.infer(color, keyframe_data.img_right)
                if getattr(Parameters, "kVolumetricIntegrationDepthEstimationFilterShadowPoints", True):
                    depth = self.volume.filter_shadow_points(depth)
        is_dev = hasattr(depth, "data_ptr")
        if not is_dev and depth.dtype != np.float32:
            factor = getattr(self.camera, "depth_factor", 1.0) if getattr(self, "use_cpp_core", False) else 1.0
            depth = depth.astype(np.float32) * np.float32(factor) if factor != 1.0 else depth.astype(np.float32)
            keyframe_data.depth = depth
        semantic, instances = keyframe_data.semantic_img, keyframe_data.semantic_instances_img
        if self.calib_map1 is not None and getattr(self, "volume_rectifies", False):
            pass  # (TSDF mode: the volume remaps colour and depth of every frame it is handed on the device, hv_tsdf_set_rectify_maps)
        elif self.calib_map1 is not None:  # base.py:1017-1043: colour bilinear, depth and labels nearest
            m1, m2 = self.calib_map1, self.calib_map2
            color = self.volume.remap(np.ascontiguousarray(color), m1, m2, linear=True)
            depth = self.volume.remap(depth, m1, m2, linear=False)
            if semantic is not None:
                semantic = self.volume.remap(np.ascontiguousarray(semantic, dtype=np.int32), m1, m2, linear=False)
            if instances is not None:
                instances = self.volume.remap(np.ascontiguousarray(instances, dtype=np.int32), m1, m2, linear=False)
        if self.depth_estimator is not None and keyframe_data.id not in self.img_id_to_depth:
            # base.py:1050-1052 (a view of a ring slot is copied: the slot is reused after this call)
            self.img_id_to_depth[keyframe_data.id] = np.array(depth) if isinstance(depth, np.ndarray) and depth.base is not None else depth
        if getattr(self, "volume_takes_bgr", False):
            color_rgb = np.ascontiguousarray(color)  # the pack kernel swaps the channels (hv_tsdf_set_color_order): no host pass
        else:
            color_rgb = np.ascontiguousarray(color[..., ::-1])  # cv2.COLOR_BGR2RGB, base.py:1054
        if is_dev:  # the fused calls want colour and depth in the same place
            import torch

            color_rgb = torch.from_numpy(color_rgb).to(depth.device)
        return color_rgb, depth, pts3d, semantic, instances

    def volume_integration(self, *args, **kwargs):  # base.py:1100-1117
        raise NotImplementedError

    def reset_if_requested(self, reset_mutex, reset_requested, q_in, q_in_condition, q_out, q_out_condition):  # base.py:633-650
        with reset_mutex:
            if reset_requested.value == 1:
                with q_in_condition:
                    empty_queue(q_in, self._drop_task)
                    q_in_condition.notify_all()
                with q_out_condition:
                    empty_queue(q_out, st.drop_output)
                    q_out_condition.notify_all()
                try:
                    self.volume.reset()
                except Exception:
                    traceback.print_exc()
                reset_requested.value = 0

    def run(self, camera, environment_type, sensor_type, viewer_queue, q_in, q_in_condition, q_out, q_out_condition,
            q_management, is_running, is_looping, reset_mutex, reset_requested, load_request_completed,
            load_request_condition, save_request_completed, save_request_condition, time_volumetric_integration,
            parameters_dict, constructor_kwargs):  # base.py:789-967
        is_running.value = 1

        def signal_handler(signum, frame):
            is_running.value = 0
            with q_in_condition:
                q_in_condition.notify_all()

        try:
            signal.signal(signal.SIGTERM, signal_handler)
            signal.signal(signal.SIGINT, signal_handler)
        except ValueError:
            pass
        try:
            self.init(camera, environment_type, sensor_type, parameters_dict, constructor_kwargs)
        except Exception:
            traceback.print_exc()
            is_running.value = 0
            return
        ring = getattr(self, "frame_ring", None)
        if ring is not None and hasattr(self.volume, "register_host_memory"):
            try:  # page-lock the ring: the H2D DMA then reads a keyframe's slot in place (no staging copy)
                self.volume.register_host_memory(ring.base_address(), ring.slot_bytes * ring.n_slots)
            except Exception:
                traceback.print_exc()
        q_in_w = _WorkerQueue(q_in, ring)
        is_looping.value = 1
        timer_fps = TimerFps("VolumetricIntegratorBase")
        timer_fps.start()
        while is_running.value == 1:
            try:
                has_management_task = not q_management.empty()
                has_regular_task = not q_in.empty()
                with q_in_condition:
                    while (not has_management_task and not has_regular_task and is_running.value == 1
                           and reset_requested.value != 1):
                        q_in_condition.wait(timeout=1.0)
                        has_management_task = not q_management.empty()
                        has_regular_task = not q_in.empty()
                if is_running.value == 0:
                    break
                if has_regular_task or has_management_task:
                    q_in_size = q_in.qsize()
                    try:
                        self.volume_integration(q_in_w, q_out, q_out_condition, q_management, viewer_queue, is_running,
                                                load_request_completed, load_request_condition, save_request_completed,
                                                save_request_condition, time_volumetric_integration)
                    finally:
                        q_in_w.release_consumed()  # the keyframes this call took are fused (or dropped): their slots are free
                    timer_fps.refresh()
                    fps = timer_fps.get_fps()
                    if (Parameters.kVolumetricIntegrationFpsThrottleEnabled
                            and q_in_size > Parameters.kVolumetricIntegrationFpsThrottleMinQueueSize
                            and fps > Parameters.kVolumetricIntegrationFpsMaxThreshold > 0):  # base.py:923-938
                        time.sleep(Parameters.kVolumetricIntegrationFpsThrottleBaseDelay
                                   + (fps - Parameters.kVolumetricIntegrationFpsMaxThreshold) * Parameters.kVolumetricIntegrationFpsThrottleScale)
                else:
                    time.sleep(0.1)
                self.reset_if_requested(reset_mutex, reset_requested, q_in, q_in_condition, q_out, q_out_condition)
            except Exception:
                traceback.print_exc()  # base.py:952-954: log and keep going
                # ... but never leave save() waiting for a SAVE task that died (e.g. the volume reported an error)
                if save_request_completed.value == 0:
                    with save_request_condition:
                        save_request_completed.value = 1
                        save_request_condition.notify_all()
        is_looping.value = 0
        self._stop_volume_integrator_implementation()
        empty_queue(q_in, self._drop_task)
        empty_queue(q_out, st.drop_output)

    # -- shared tail of every volume_integration(): publish or acknowledge ------------------------
    def _publish(self, last_output, q_out, q_out_condition, is_running, save_request_completed, save_request_condition):
        if is_running.value == 1 and last_output is not None:
            if last_output.task_type in (VolumetricIntegrationTaskType.INTEGRATE, VolumetricIntegrationTaskType.UPDATE_OUTPUT):
                with q_out_condition:
                    last_output.timestamp = time.perf_counter()
                    if getattr(self, "frame_ring", None) is not None:
                        # big arrays leave through a shared segment (one memcpy here, one in pop_output) instead of a pickle
                        # through the queue's pipe; self.last_output keeps the arrays (the copy that travels is shallow)
                        q_out.put(st.export_arrays(_shallow_output_copy(last_output)))
                    else:
                        q_out.put(last_output)
                    q_out_condition.notify_all()
            elif last_output.task_type == VolumetricIntegrationTaskType.SAVE:
                with save_request_condition:
                    save_request_completed.value = 1
                    save_request_condition.notify_all()

    @staticmethod
    def _save_mesh(path, mesh):
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        write_ply_mesh(path, mesh.vertices, mesh.triangles, mesh.vertex_colors)

    @staticmethod
    def _save_points(path, points, colors, normals=None):
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        write_ply_points(path, points, colors, normals)
