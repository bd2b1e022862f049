"""CPU: the hand-derived known answers of tests/semantic2_kats.py on the COMPILED REFERENCE's two "*2" semantic block grids
(oracle/_ref: VoxelBlockSemanticGrid2 / VoxelBlockSemanticProbabilisticGrid2, unmodified sources of cpp/volumetric) - what the GPU
payloads HvSem2Voxel / HvProb2Voxel are held to in tests/test_gpu_semantic2_payloads.py.  The reference's own tests hold no value for
these two payloads (cpp/test_volumetric_voxel_semantic.py exercises VoxelSemanticData only): the answers are derived from
voxel_data_semantic2.h by hand."""
import numpy as np
import pytest

import oracle
from oracle.semantic import RefSemGrid2
from tests.semantic2_kats import PROB2, VOTE2, run_semantic2_kats

pytestmark = pytest.mark.skipif(not oracle.ref_available(), reason="compiled reference not available")


@pytest.fixture(autouse=True)
def reference_statics():
    def put():
        RefSemGrid2(VOTE2, 0.05).set_depth_threshold(10.0)  # voxel_data_semantic2.h:47-48
        g = RefSemGrid2(PROB2, 0.05)
        g.set_depth_threshold(5.0)  # :258-261
        g.set_depth_decay_rate(0.07)

    put()
    yield
    put()


def test_hand_derived_answers_on_the_compiled_reference():
    run_semantic2_kats(lambda kind, voxel: RefSemGrid2(kind, voxel))


def test_marginal_confidences_of_the_compiled_reference():
    """get_object_confidence / get_class_confidence: the marginal probability of the most likely id (kind 3), counter / count (kind 2)."""
    g = RefSemGrid2(PROB2, 0.1)
    n = 3
    g.integrate(np.full((n, 3), 0.01, np.float32), np.zeros((n, 3), np.uint8), np.array([4, 4, 3], np.int32), np.array([8, 7, 8], np.int32))
    ints = g.dump()[1]
    at = np.nonzero(ints[..., 0])
    oc, cc = g.dump_marginals()
    assert abs(float(oc[at][0]) - 0.5) < 1e-7 and abs(float(cc[at][0]) - 0.5) < 1e-7
    g = RefSemGrid2(VOTE2, 0.1)
    g.integrate(np.full((5, 3), 0.01, np.float32), np.zeros((5, 3), np.uint8), np.array([3, 3, 3, 4, 3], np.int32), np.array([7, 8, 8, 8, 8], np.int32))
    ints = g.dump()[1]
    at = np.nonzero(ints[..., 0])
    oc, cc = g.dump_marginals()
    # object: 7 (1) -> 8 (1) -> 2 -> 3 -> 4 => 4/5; class: 3 (1) -> 2 -> 3 -> 2 -> 3 => 3/5
    assert float(oc[at][0]) == np.float32(4) / np.float32(5) and float(cc[at][0]) == np.float32(3) / np.float32(5)
    assert tuple(ints[at][0]) == (5, 8, 3, 3)  # count, object, class, min of the two counters
