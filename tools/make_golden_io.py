#!/usr/bin/env python3
"""Generate tests/golden/numpy_b64.json by importing the *reference's* NumpyB64Json
(/root/reference/pyslam/utilities/serialization.py:421-484) in the dev container.  The module's unrelated imports
(ujson, the package-relative logger) are stubbed; only NumpyB64Json.numpy_to_json runs.  The fixture pins the
on-disk encoding of the images inside map.json that pyslam_amd/io/system_state.py reads and writes."""
import importlib.util
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_reference_serialization():
    sys.modules.setdefault("ujson", json)
    pkg = types.ModuleType("refpkg")
    pkg.__path__ = []
    sys.modules["refpkg"] = pkg
    log = types.ModuleType("refpkg.logging")
    log.Printer = type("Printer", (), {k: staticmethod(lambda *a, **kw: None) for k in ("red", "green", "yellow", "orange", "blue", "error")})
    sys.modules["refpkg.logging"] = log
    spec = importlib.util.spec_from_file_location("refpkg.serialization", "/root/reference/pyslam/utilities/serialization.py")
    m = importlib.util.module_from_spec(spec)
    sys.modules["refpkg.serialization"] = m
    spec.loader.exec_module(m)
    return m


def arrays():
    rng = np.random.default_rng(21)
    return {
        "img_u8": rng.integers(0, 255, (4, 5, 3)).astype(np.uint8),
        "depth_f32": rng.random((4, 5)).astype(np.float32),
        "depth_u16": rng.integers(0, 65535, (3, 4)).astype(np.uint16),
        "labels_i32": rng.integers(-1, 40, (3, 4)).astype(np.int32),
        "fortran_f64": np.asfortranarray(rng.random((3, 2))),
    }


def main():
    m = load_reference_serialization()
    out = {k: m.NumpyB64Json.numpy_to_json(v) for k, v in arrays().items()}
    for k, v in out.items():
        back = m.NumpyB64Json.json_to_numpy(v)
        assert np.array_equal(back, arrays()[k])
    with open(os.path.join(ROOT, "tests", "golden", "numpy_b64.json"), "w") as f:
        json.dump(out, f, indent=1, default=lambda o: list(o))
    print("wrote tests/golden/numpy_b64.json", {k: (v["dtype"], v["shape"], v.get("order")) for k, v in out.items()})


if __name__ == "__main__":
    main()
