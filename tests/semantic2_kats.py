"""Hand-derived known answers for the reference's two "*2" semantic payloads (cpp/volumetric/voxel_data_semantic2.h), written against
any grid factory `make(kind, voxel_size)` whose objects offer integrate / get_voxels / merge_segments (the compiled reference on the
CPU: tests/test_semantic2_payloads_reference.py; the GPU grids: tests/test_gpu_semantic2_payloads.py).

kind 2 = VoxelBlockSemanticGrid2 (VoxelSemanticData2, :46-196): the object id and the class id each have their own confidence counter
         (+1 on a match, -1 on a mismatch, replaced and reset to 1 when the counter reaches 0); confidence = min(object counter / count,
         class counter / count), each clamped to 1.
kind 3 = VoxelBlockSemanticProbabilisticGrid2 (VoxelSemanticDataProbabilistic2, :256-787): one map object id -> log-probability and one
         class id -> log-probability; an observation adds HALF its log-probability to its object's entry and half to its class's; the
         log-probability of an observation is 0 without a depth or at depth <= kDepthThreshold and -(depth - threshold) * kDepthDecayRate
         beyond (no base log-probability per observation: near observations leave every entry at 0); most likely id = the largest
         log-probability, the smallest id among equals; confidence = exp(lp_obj + lp_cls - logsumexp(objects) - logsumexp(classes)).
"""
import numpy as np

VOTE2, PROB2 = 2, 3


def _one_voxel(g, obj, cls, depths=None):
    n = len(obj)
    pts = np.full((n, 3), 0.01, np.float32)
    g.integrate(pts, np.zeros((n, 3), np.uint8), np.asarray(cls, np.int32), np.asarray(obj, np.int32),
                None if depths is None else np.asarray(depths, np.float32))
    pts, cols, c, o, conf = g.get_voxels(1, -1.0)
    assert len(o) == 1
    return int(o[0]), int(c[0]), float(conf[0])


def run_semantic2_kats(make):
    # ---- kind 2: the two counters move independently -------------------------------------------------------------------------------
    # (7,3) (8,3) (8,4) (8,4) (9,4): object 7:1 -> 0 -> 8:1 -> 2 -> 1; class 3:1 -> 2 -> 1 -> 0 -> 4:1 ... see below
    g = make(VOTE2, 0.1)
    # object: init 7 (1); 8 != 7 -> 0 -> object 8 (1); 8 -> 2; 8 -> 3; 9 -> 2           => object 8, counter 2
    # class:  init 3 (1); 3 -> 2;                      4 -> 1; 4 -> 0 -> class 4 (1); 4 -> 2 => class 4, counter 2
    o, c, conf = _one_voxel(g, [7, 8, 8, 8, 9], [3, 3, 4, 4, 4])
    assert (o, c) == (8, 4)
    assert conf == np.float32(2) / np.float32(5)
    # the joint counter of the plain voting payload would have ended elsewhere: (7,3) 1; (8,3) 0 -> (8,3) 1; (8,4) 0 -> (8,4) 1; (8,4) 2; (9,4) 1
    # depth gate: observations at depth >= kDepthThreshold (10 m) leave the labels alone but are counted
    g = make(VOTE2, 0.1)
    o, c, conf = _one_voxel(g, [5, 6, 6], [1, 2, 2], depths=[20.0, 1.0, 30.0])
    # first observation gated out (no label), second: update on an unlabelled voxel: counters 0 - 1 <= 0 -> (6, 2) with counters 1
    assert (o, c) == (6, 2) and conf == np.float32(1) / np.float32(3)

    # ---- kind 3: near observations carry log-probability 0 ---------------------------------------------------------------------------
    g = make(PROB2, 0.1)
    # objects {7: 0, 8: 0}, classes {3: 0, 4: 0}: ties -> the smallest ids; confidence = 1 / (2 * 2)
    o, c, conf = _one_voxel(g, [8, 7, 8], [4, 4, 3])
    assert (o, c) == (7, 3)
    assert abs(conf - 0.25) < 1e-7
    # far observations (threshold 5 m, decay 0.07 / m): (1, 10) at 15 m adds -0.35 to object 1 and to class 10; (2, 10) near adds 0 to
    # object 2 and class 10 -> objects {1: -0.35, 2: 0}, classes {10: -0.35}: most likely (2, 10),
    # confidence = exp(0 - 0.35 - log(exp(-0.35) + 1) + 0.35) = 1 / (1 + exp(-0.35))
    g = make(PROB2, 0.1)
    o, c, conf = _one_voxel(g, [1, 2], [10, 10], depths=[15.0, 1.0])
    assert (o, c) == (2, 10)
    assert abs(conf - 1.0 / (1.0 + np.exp(-0.35))) < 2e-6
    # set_object_id through merge_segments(9, 7): object 9 enters the map with the log-probability of the most likely object and IS
    # the most likely one although 7 < 9 holds the same value (:424-452) - until the next update drops the cache and the smallest id
    # among equals wins again (:366-368, :601-622)
    g = make(PROB2, 0.1)
    _one_voxel(g, [8, 7, 8], [4, 4, 3])
    g.merge_segments(9, 7)
    pts, cols, c, o, conf = g.get_voxels(1, -1.0)
    assert (int(o[0]), int(c[0])) == (9, 3) and abs(float(conf[0]) - 1.0 / 6.0) < 1e-7  # objects {7, 8, 9} x classes {3, 4}
    o, c, conf = _one_voxel(g, [8], [4])
    assert (o, c) == (7, 3) and abs(conf - 1.0 / 6.0) < 1e-7
