"""CPU: pyslam_amd.volumetric_module offers every name the reference's `volumetric` extension module binds (fixture
tests/golden/volumetric_module_names.json, read from the binding sources by tools/make_golden_module_names.py); plus the small host helpers of that namespace against their C++ definitions."""
import json
import os
import pickle

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "volumetric_module_names.json")


def test_every_bound_name_exists():
    import pyslam_amd.volumetric_module as volumetric

    names = json.load(open(GOLD))
    assert len(names) >= 34
    missing = [n for n in names if not hasattr(volumetric, n)]
    assert missing == [], missing


def test_quaterniond_like_eigen():
    from pyslam_amd.volumetric_module import CameraFrustrum, Quaterniond

    assert list(Quaterniond().coeffs()) == [1.0, 0.0, 0.0, 0.0]
    q = Quaterniond(0.5, -0.5, 0.5, 0.5)
    assert (q.w(), q.x(), q.y(), q.z()) == (0.5, -0.5, 0.5, 0.5)
    R = q.toRotationMatrix()
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-15)
    np.testing.assert_allclose(q.inverse().toRotationMatrix(), R.T, atol=1e-15)
    np.testing.assert_allclose(q.conjugate().coeffs(), [0.5, 0.5, -0.5, -0.5])
    big = Quaterniond(np.array([2.0, 0.0, 0.0, 0.0]))
    assert list(big.normalized().coeffs()) == [1.0, 0.0, 0.0, 0.0]
    big.normalize()
    assert big.w() == 1.0
    back = pickle.loads(pickle.dumps(q))
    assert list(back.coeffs()) == list(q.coeffs())
    # the frustum's (orientation, translation) constructor and setter take it; get_orientation_cw hands one back
    fr = CameraFrustrum(500.0, 500.0, 320.0, 240.0, 640, 480, q, np.zeros(3), 10.0, 0.1)
    np.testing.assert_allclose(fr.get_R_cw(), R, atol=1e-15)
    got = fr.get_orientation_cw()
    assert isinstance(got, Quaterniond)
    np.testing.assert_allclose(np.abs(got.coeffs()), np.abs(q.coeffs()), atol=1e-15)


def test_image_helpers(capsys):
    from pyslam_amd.volumetric_module import check_image_size, convert_image_type_if_needed

    img = np.zeros((4, 6), np.int32)
    assert check_image_size(img, 4, 6, "ids") and not check_image_size(img, 6, 4, "ids") and not check_image_size(np.zeros((0, 0)), 0, 0, "e")
    assert "Image size does not match expected size" in capsys.readouterr().out
    assert convert_image_type_if_needed(img, np.int32, "ids") is img and convert_image_type_if_needed(img, 4, "ids") is img  # CV_32S
    f = np.array([[0.5, 1.5, 2.5, -3.7, 300.0, -1e9]], np.float32)
    np.testing.assert_array_equal(convert_image_type_if_needed(f, 0, "f"), np.array([[0, 2, 2, 0, 255, 0]], np.uint8))  # saturate_cast<uchar>
    np.testing.assert_array_equal(convert_image_type_if_needed(f, np.int32, "f"), np.array([[0, 2, 2, -4, 300, -1000000000]], np.int32))
    assert convert_image_type_if_needed(np.zeros((0, 0), np.uint8), 4, "e").size == 0
