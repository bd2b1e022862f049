"""CPU: the drop-in front (pyslam_amd.dense) speaks the reference's task/queue protocol —
factory, add_keyframe gating, UPDATE_OUTPUT, pop_output, save(dense_map.ply), rebuild, reset, quit
(volumetric_integrator_base.py:451-1342).  The worker process runs oracle-backed stand-in volumes
(injected through the constructor kwargs); the GPU tests run the same flow on the real volumes."""
import os
import time

import numpy as np
import pytest

from pyslam_amd.dense import (
    VolumetricIntegrationTaskType,
    VolumetricIntegratorType,
    volumetric_integrator_factory,
)
from pyslam_amd.dense.parameters import Parameters
from pyslam_amd.dense.ply_io import read_ply
from pyslam_amd.dense.volumetric_integrator_types import DatasetEnvironmentType, SensorType
from pyslam_amd.synthetic import SyntheticRGBD
from tests import dense_helpers as dh


def wait_until(cond, timeout=30.0):
    t0 = time.time()
    while time.time() - t0 < timeout:
        if cond():
            return True
        time.sleep(0.05)
    return False


@pytest.fixture
def small_params():
    old = (Parameters.kVolumetricIntegrationVoxelLength, Parameters.kVolumetricIntegrationTSdfTrunc,
           Parameters.kVolumetricIntegrationOutputTimeInterval, Parameters.kVolumetricIntegrationMinNumLBATimes)
    Parameters.kVolumetricIntegrationVoxelLength = 0.04
    Parameters.kVolumetricIntegrationTSdfTrunc = 0.12
    Parameters.kVolumetricIntegrationOutputTimeInterval = 0.0
    Parameters.kVolumetricIntegrationMinNumLBATimes = 1
    yield
    (Parameters.kVolumetricIntegrationVoxelLength, Parameters.kVolumetricIntegrationTSdfTrunc,
     Parameters.kVolumetricIntegrationOutputTimeInterval, Parameters.kVolumetricIntegrationMinNumLBATimes) = old


def test_types_and_factory_errors():
    assert VolumetricIntegratorType.from_string("TSDF") is VolumetricIntegratorType.TSDF
    assert [t.value for t in VolumetricIntegratorType] == [0, 1, 2, 3, 4]
    with pytest.raises(ValueError):
        VolumetricIntegratorType.from_string("NOPE")
    assert [t.name for t in VolumetricIntegrationTaskType] == ["NONE", "INTEGRATE", "SAVE", "LOAD", "RESET", "UPDATE_OUTPUT"]


@pytest.mark.parametrize("kind", ["TSDF", "VOXEL_GRID"])
def test_protocol_end_to_end(kind, small_params, tmp_path):
    s = SyntheticRGBD("tiny_160x120_2cm", noise=False)
    cam = dh.FakeCamera(s)
    factory = dh.oracle_tsdf_factory if kind == "TSDF" else dh.oracle_grid_factory
    integ = volumetric_integrator_factory(VolumetricIntegratorType.from_string(kind), cam, DatasetEnvironmentType.INDOOR,
                                          SensorType.RGBD, volume_factory=factory)
    try:
        assert wait_until(integ.is_ready), "worker did not start"
        kfs = [dh.FakeKeyFrame(i, s, cam) for i in range(3)]
        held = dh.FakeKeyFrame(3, s, cam, lba_count=0)  # gated until local BA has touched it (base.py:1153-1157)
        for kf in kfs + [held]:
            integ.add_keyframe(kf, kf.img, None, kf.depth_img)
        # one output per integrate call; queued keyframes may be fused by one multi-frame sweep (TSDF), so
        # between 1 and 3 outputs arrive for the 3 released keyframes
        outs = []
        assert wait_until(lambda: (outs.append(integ.pop_output(timeout=0.2)) or True) and any(o is not None for o in outs))
        time.sleep(1.0)
        while (o := integ.pop_output(timeout=0.1)) is not None:
            outs.append(o)
        outs = [o for o in outs if o is not None]
        assert 1 <= len(outs) <= 3 and {o.id for o in outs} <= {0, 1, 2}
        assert all(o.task_type == VolumetricIntegrationTaskType.INTEGRATE for o in outs)
        assert len(integ.keyframe_queue) == 1  # the gated keyframe is still waiting
        held.lba_count = 1
        assert wait_until(lambda: len(integ.keyframe_queue) == 0, 5.0)  # 0.5 s timer re-flushes (base.py:369-374)
        assert wait_until(lambda: (o := integ.pop_output(timeout=0.2)) is not None and o.id == 3)
        integ.add_update_output_task()
        o = None
        assert wait_until(lambda: (o := integ.pop_output(timeout=0.2)) is not None and o.task_type == VolumetricIntegrationTaskType.UPDATE_OUTPUT)
        assert integ.time_volumetric_integration.value > 0.0
        # save() blocks until dense_map.ply is on disk (base.py:574-593)
        integ.save(str(tmp_path))
        pts, cols, faces = read_ply(str(tmp_path / "dense_map.ply"))
        assert len(pts) > 100 and cols is not None
        if kind == "TSDF":
            assert faces is not None and faces.max() < len(pts)
        # rebuild(): RESET output first, then every good keyframe is replayed (base.py:1242-1318)
        while integ.pop_output(timeout=0.05) is not None:
            pass
        integ.rebuild(dh.FakeMap(kfs))
        seen = []
        assert wait_until(lambda: (seen.append(integ.pop_output(timeout=0.2)) or True)
                          and sum(1 for x in seen if x is not None and x.task_type == VolumetricIntegrationTaskType.INTEGRATE) >= 1, 20.0)
        assert any(x is not None and x.task_type == VolumetricIntegrationTaskType.RESET for x in seen)
        integ.request_reset()
        assert integ.reset_requested.value == 0
    finally:
        integ.quit()
    assert not integ.process.is_alive()
    assert integ.pop_output() is None  # after quit: None, not an exception


def test_outputs_have_reference_fields(small_params):
    from pyslam_amd.dense import VolumetricIntegrationMesh, VolumetricIntegrationOutput, VolumetricIntegrationPointCloud

    pc = VolumetricIntegrationPointCloud(points=np.zeros((2, 3)), colors=np.ones((2, 3)))
    for f in ("points", "colors", "semantics", "object_ids", "semantic_colors", "object_colors"):
        assert hasattr(pc, f)
    mesh = VolumetricIntegrationMesh(dh.OracleTsdfVolume(0.04, 0.12).extract_triangle_mesh())
    for f in ("vertices", "triangles", "vertex_colors", "vertex_normals"):
        assert hasattr(mesh, f)
    out = VolumetricIntegrationOutput(VolumetricIntegrationTaskType.INTEGRATE, 7, pc, mesh)
    assert (out.task_type, out.id, out.point_cloud, out.mesh, out.objects) == (VolumetricIntegrationTaskType.INTEGRATE, 7, pc, mesh, None)
    assert out.timestamp > 0


@pytest.mark.parametrize("kind", ["VOXEL_SEMANTIC_GRID", "VOXEL_SEMANTIC_PROBABILISTIC_GRID"])
def test_semantic_protocol_end_to_end(kind, small_params, tmp_path):
    """The semantic integrators behind the same protocol: labelled keyframes in, object lists / labelled point
    clouds out (volumetric_integrator_voxel_semantic_grid.py:208-748), on the compiled-reference stand-in."""
    import oracle

    if not oracle.ref_available():
        pytest.skip("compiled reference not available")
    from pyslam_amd.dense import VolumetricIntegrationObjectList
    from tests.semantic_helpers import CFG

    old = (Parameters.kVolumetricIntegrationVoxelGridMinCount, Parameters.kVolumetricIntegrationVoxelGridMinConfidence)
    Parameters.kVolumetricIntegrationVoxelGridMinCount, Parameters.kVolumetricIntegrationVoxelGridMinConfidence = 1, 0.0
    s = SyntheticRGBD(dict(CFG, width=160, height=120, fx=131.25, fy=131.25, cx=79.5, cy=59.5), noise=False)
    cam = dh.FakeCamera(s)
    integ = volumetric_integrator_factory(VolumetricIntegratorType.from_string(kind), cam, DatasetEnvironmentType.INDOOR,
                                          SensorType.RGBD, volume_factory=dh.oracle_semantic_factory)
    try:
        assert wait_until(integ.is_ready), "worker did not start"
        outs = []
        for i in (0, 6, 12):
            kf = dh.FakeKeyFrame(i, s, cam, semantic=True)
            integ.add_keyframe(kf, kf.img, None, kf.depth_img)
            got = []
            assert wait_until(lambda: (got.append(integ.pop_output(timeout=0.2)) or True) and got[-1] is not None)
            outs.append(got[-1])
        last = outs[-1]
        assert last.task_type == VolumetricIntegrationTaskType.INTEGRATE and last.id == 12
        assert isinstance(last.objects, VolumetricIntegrationObjectList) and last.point_cloud is None
        assert last.objects.num_objects == len(last.objects.object_list) >= 1
        o = last.objects.object_list[0]
        assert o.points.dtype == np.float32 and o.oriented_bounding_box.box_matrix.shape == (4, 4)
        assert last.objects.object_colors.shape == (last.objects.num_objects, 3)
        integ.save(str(tmp_path))
        pts, cols, faces = read_ply(str(tmp_path / "dense_map.ply"))
        assert len(pts) > 100 and faces is None
    finally:
        integ.quit()
        Parameters.kVolumetricIntegrationVoxelGridMinCount, Parameters.kVolumetricIntegrationVoxelGridMinConfidence = old


def test_semantic_point_cloud_output_without_instances(small_params):
    """No instance ids -> one labelled point cloud (reference :588-690), in-process on the stand-in volume."""
    import oracle

    if not oracle.ref_available():
        pytest.skip("compiled reference not available")
    from pyslam_amd.dense.volumetric_integrator_voxel_semantic_grid import VolumetricIntegratorVoxelSemanticGrid
    from tests.semantic_helpers import CFG, semantic_frame

    s = SyntheticRGBD(dict(CFG, width=160, height=120, fx=131.25, fy=131.25, cx=79.5, cy=59.5), noise=False)
    cam = dh.FakeCamera(s)
    integ = VolumetricIntegratorVoxelSemanticGrid.__new__(VolumetricIntegratorVoxelSemanticGrid)
    integ.init(cam, DatasetEnvironmentType.INDOOR, SensorType.RGBD, {}, dict(volume_factory=dh.oracle_semantic_factory))
    depth, rgb, T, cls_img, _ = semantic_frame(s, 0)
    integ.integrate_keyframe(rgb, depth, T, cls_img, None)
    integ.last_integrated_id = 0
    out = integ.make_output(VolumetricIntegrationTaskType.INTEGRATE)
    assert out.objects is None and out.point_cloud is not None
    pc = out.point_cloud
    assert pc.points.dtype == np.float32 and pc.semantics.dtype == np.int32 and len(pc.points) == len(pc.semantics) > 0
    assert pc.semantic_colors.shape == pc.points.shape and pc.object_colors.shape == pc.points.shape
    assert set(np.unique(pc.object_ids)) <= {0}  # no instance image: default object id 0


def test_push_to_front_is_newest_first_and_rebuild_drains():
    """the drain-and-refill idioms need queues whose put() is synchronous (manager queues, as in the reference).
    A backlog pushed with front=True must come out newest-first, and empty_queue must see everything this process put."""
    import multiprocessing as mp

    from pyslam_amd.dense.volumetric_integrator_base import empty_queue, push_to_front

    with mp.Manager() as m:
        q = m.Queue()
        for i in range(100):
            push_to_front(q, i)
        out = [q.get(block=False) for _ in range(100)]
        assert out == list(range(99, -1, -1))
        for i in range(50):
            q.put(i)
        empty_queue(q)
        assert q.empty()


def test_result_arrays_fall_back_to_plain_numpy(monkeypatch):
    """Extraction results go to page-locked arrays where torch can provide them; without a GPU, for small results, or with
    PYSLAM_AMD_PINNED_RESULTS=0 the caller gets a plain numpy array of the same shape and dtype."""
    import numpy as np

    from pyslam_amd.volumetric import _result_array

    for shape, dtype in (((0, 3), np.float64), ((5, 3), np.int32), ((200_000, 3), np.float64)):
        a = _result_array(shape, dtype)
        assert isinstance(a, np.ndarray) and a.shape == shape and a.dtype == dtype and a.flags.writeable
    monkeypatch.setenv("PYSLAM_AMD_PINNED_RESULTS", "0")
    a = _result_array((200_000, 3), np.float64)
    assert a.shape == (200_000, 3) and a.base is None  # an owning numpy array, not a view of a torch tensor


# ---- shared-memory transport of the front (pyslam_amd/dense/shared_transport.py) ------------------------------------------------
def _ring_child(ring, ref, q):
    """Runs in a spawned process: attach, read through the view, release the slot."""
    a = ring.view(ref)
    q.put((float(a.sum()), a.shape, str(a.dtype)))
    ring.release(0)


def test_frame_ring_round_trip_across_processes():
    import multiprocessing as mp

    from pyslam_amd.dense import shared_transport as st

    ctx = mp.get_context("spawn")
    ring = st.FrameRing(ctx, 1 << 20, 4)
    try:
        slot = ring.acquire()
        assert slot == 0 and ring.held() == 1
        a = np.arange(120 * 160, dtype=np.float32).reshape(120, 160)
        refs = ring.write(slot, {"depth": a, "img": np.full((120, 160, 3), 7, np.uint8)})
        assert set(refs) == {"depth", "img"} and refs["depth"].offset % 256 == 0 and refs["img"].offset % 256 == 0
        np.testing.assert_array_equal(ring.view(refs["depth"]), a)
        assert ring.write(slot, {"big": np.zeros(2 << 20, np.uint8)}) is None  # does not fit a slot
        q = ctx.Queue()
        p = ctx.Process(target=_ring_child, args=(ring, refs["depth"], q))
        p.start()
        got = q.get(timeout=60)
        p.join(30)
        assert got == (float(a.sum()), (120, 160), "float32")
        assert ring.held() == 0  # released by the other process
        slots = [ring.acquire() for _ in range(5)]
        # round-robin: the slot released last (0) is reused LAST, and a full ring says so
        assert slots[:4] == [1, 2, 3, 0] and slots[4] is None
        ring.release(2)
        ring.release(1)
        assert ring.acquire() == 1 and ring.acquire() == 2 and ring.acquire() is None
    finally:
        ring.close()


def test_keyframe_images_move_into_the_ring_and_back():
    import multiprocessing as mp

    from pyslam_amd.dense import shared_transport as st
    from pyslam_amd.dense.volumetric_integrator_base import VolumetricIntegrationTask

    s = SyntheticRGBD("tiny_160x120_2cm", noise=False)
    cam = dh.FakeCamera(s)
    kf = dh.FakeKeyFrame(0, s, cam, semantic=True)
    ring = st.FrameRing(mp.get_context("spawn"), 160 * 120 * 18 + 2048, 2)
    try:
        task = VolumetricIntegrationTask(kf, task_type=VolumetricIntegrationTaskType.INTEGRATE)
        task.keyframe_data.semantic_instances_img = kf.semantic_instances_img
        assert st.keyframe_to_ring(ring, task.keyframe_data)
        kd = task.keyframe_data
        assert all(isinstance(getattr(kd, f), st.ArrayRef) for f in ("img", "depth", "semantic_img", "semantic_instances_img"))
        assert kd.img_right is None and kd._ring_slot == 0
        assert kf.img.dtype == np.uint8  # the keyframe's own arrays are untouched
        import pickle

        assert len(pickle.dumps(task)) < 8192  # the queue item is control only
        kd2 = pickle.loads(pickle.dumps(task)).keyframe_data
        slot = st.keyframe_from_ring(ring, kd2)
        assert slot == 0
        np.testing.assert_array_equal(kd2.img, kf.img)
        np.testing.assert_array_equal(kd2.depth, kf.depth_img)
        np.testing.assert_array_equal(kd2.semantic_img, kf.semantic_img)
        assert not kd2.depth.flags.owndata  # a view of the slot, not a copy
        # a second keyframe takes slot 1, a third finds the ring full and stays as it is (travels pickled, like the reference)
        t2 = VolumetricIntegrationTask(kf, task_type=VolumetricIntegrationTaskType.INTEGRATE)
        t3 = VolumetricIntegrationTask(kf, task_type=VolumetricIntegrationTaskType.INTEGRATE)
        assert st.keyframe_to_ring(ring, t2.keyframe_data) and not st.keyframe_to_ring(ring, t3.keyframe_data)
        assert isinstance(t3.keyframe_data.depth, np.ndarray)
        st.drop_task(ring, t2)
        ring.release(slot)
        assert ring.held() == 0
    finally:
        ring.close()


def test_control_queue_put_front_get_batch_drain():
    import multiprocessing as mp

    from pyslam_amd.dense import shared_transport as st
    from pyslam_amd.dense.volumetric_integrator_base import (VolumetricIntegrationTask, empty_queue, push_to_front,
                                                             take_integrate_backlog)

    m = st.start_manager(mp.get_context("spawn"))
    try:
        q = m.ControlQueue()
        for i in range(100):
            push_to_front(q, i)  # -> put_front: one round trip each
        assert q.qsize() == 100
        assert [q.get(block=False) for _ in range(100)] == list(range(99, -1, -1))  # newest first, like drain-and-refill
        I, U = VolumetricIntegrationTaskType.INTEGRATE, VolumetricIntegrationTaskType.UPDATE_OUTPUT
        for t in (I, I, I, U, I):
            q.put(VolumetricIntegrationTask(task_type=t))
        got = take_integrate_backlog(q, 2)
        assert [t.task_type for t in got] == [I, I]
        got = take_integrate_backlog(q, 16)
        assert [t.task_type for t in got] == [I]  # stops at the UPDATE_OUTPUT task, which stays at the front
        assert q.get(block=False).task_type == U and q.qsize() == 1
        q.put(None)
        dropped = []
        empty_queue(q, dropped.append)
        assert q.empty() and len(dropped) == 2 and dropped[1] is None
        with pytest.raises(Exception):
            q.get(timeout=0.05)  # queue.Empty through the proxy
    finally:
        m.shutdown()


def test_output_arrays_travel_through_a_shared_segment():
    from multiprocessing import shared_memory

    from pyslam_amd.dense import VolumetricIntegrationOutput, VolumetricIntegrationPointCloud
    from pyslam_amd.dense import shared_transport as st
    from pyslam_amd.dense.volumetric_integrator_base import _shallow_output_copy

    pts = np.random.default_rng(0).random((50_000, 3)).astype(np.float32)
    cols = np.random.default_rng(1).random((50_000, 3)).astype(np.float32)
    out = VolumetricIntegrationOutput(VolumetricIntegrationTaskType.INTEGRATE, 5,
                                      VolumetricIntegrationPointCloud(points=pts, colors=cols, semantics=np.arange(10, dtype=np.int32)))
    import pickle

    travelling = st.export_arrays(_shallow_output_copy(out))
    assert isinstance(out.point_cloud.points, np.ndarray)  # the worker's own last_output is untouched
    assert isinstance(travelling.point_cloud.points, st.ArrayRef) and isinstance(travelling.point_cloud.semantics, np.ndarray)  # small: inline
    name = travelling._shm_segment
    wire = pickle.dumps(travelling)
    assert len(wire) < 4096
    got = st.import_arrays(pickle.loads(wire))
    np.testing.assert_array_equal(got.point_cloud.points, pts)
    np.testing.assert_array_equal(got.point_cloud.colors, cols)
    assert got.id == 5 and got.point_cloud.points.flags.writeable  # views of a private mapping of the (already unlinked) segment
    with pytest.raises(FileNotFoundError):
        shared_memory.SharedMemory(name=name)  # unlinked by the consumer
    dropped = st.export_arrays(_shallow_output_copy(out))
    name2 = dropped._shm_segment
    st.drop_output(dropped)
    with pytest.raises(FileNotFoundError):
        shared_memory.SharedMemory(name=name2)


def test_front_moves_keyframes_through_the_ring(small_params):
    """End to end on the stand-in volume: every keyframe goes through a ring slot, all slots come back, the mesh arrives whole."""
    s = SyntheticRGBD("tiny_160x120_2cm", noise=False)
    cam = dh.FakeCamera(s)
    integ = volumetric_integrator_factory(VolumetricIntegratorType.TSDF, cam, DatasetEnvironmentType.INDOOR, SensorType.RGBD,
                                          volume_factory=dh.oracle_tsdf_factory)
    try:
        assert wait_until(integ.is_ready)
        assert integ.frame_ring is not None and integ.frame_ring.held() == 0
        kfs = [dh.FakeKeyFrame(i, s, cam) for i in range(6)]
        for kf in kfs:
            integ.add_keyframe(kf, kf.img, None, kf.depth_img)
        outs = []
        # (tasks are pushed to the FRONT of q_in and a backlog is fused by one sweep: which keyframe id an output carries depends
        # on the interleaving; what is fixed is that every keyframe is consumed and every slot comes back)
        assert wait_until(lambda: integ.q_in.qsize() == 0 and integ.frame_ring.held() == 0, 60.0)
        assert wait_until(lambda: (outs.append(integ.pop_output(timeout=0.2)) or True) and any(o is not None for o in outs), 30.0)
        time.sleep(0.5)
        while (o := integ.pop_output(timeout=0.1)) is not None:
            outs.append(o)
        last = [o for o in outs if o is not None][-1]
        assert {o.id for o in outs if o is not None} <= set(range(6))
        assert isinstance(last.mesh.vertices, np.ndarray) and len(last.mesh.vertices) > 100 and last.mesh.triangles.max() < len(last.mesh.vertices)
        # rebuild(): the drained tasks give their slots back too
        integ.rebuild(dh.FakeMap(kfs))
        assert wait_until(lambda: integ.frame_ring.held() == 0 and integ.q_in.qsize() == 0, 30.0)
    finally:
        integ.quit()
