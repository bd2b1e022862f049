"""CPU: the C-ABI shared library builds for gfx950, loads, exports every symbol that include/hipvol.h
declares, and fails loudly (no CPU fallback) when no GPU is present.  No compute calls here."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hipvol.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hv_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from pyslam_amd import _lib, build

    path = build.build(verbose=False)
    assert os.path.exists(path)
    names = declared_symbols()
    assert len(names) >= 30
    lib = ctypes.CDLL(path)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/hipvol.h but not exported"
    # the Python binding covers the whole header, one to one
    assert sorted(_lib.SIGNATURES) == names
    _lib.load()


def test_code_object_targets_gfx950_only():
    from pyslam_amd import build

    path = build.build(verbose=False)
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    out = subprocess.run([objdump, "--offloading", path], capture_output=True, text=True).stdout
    archs = set(re.findall(r"gfx[0-9a-f]+", out))
    assert archs == {"gfx950"}, archs


def test_create_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from pyslam_amd._lib import HipVolError
    from pyslam_amd.volumetric import ScalableTSDFVolume, VoxelBlockGrid

    with pytest.raises(HipVolError):
        ScalableTSDFVolume(0.005, 0.04)
    with pytest.raises(HipVolError):
        VoxelBlockGrid(0.005)


def test_product_never_touches_the_oracle():
    """The product path must not import, link or execute anything under oracle/."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "pyslam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M) or re.search(r"dlopen\(.*oracle|CDLL\(.*oracle", text):
                    bad.append(os.path.join(base, f))
                if re.search(r"#include\s+[<\"].*oracle", text):
                    bad.append(os.path.join(base, f))
    assert not bad, bad
    out = subprocess.run(["ldd", os.path.join(ROOT, "pyslam_amd", "lib", "libpyslam_hipvol.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_build_digest_covers_every_source_and_header(tmp_path, monkeypatch):
    """a header-only edit (hv_semantic.h, hv_bucket.h, hv_query.h were not hashed) must change the digest the
    .so and every profile under profiles/ are keyed by."""
    import shutil

    from pyslam_amd import build

    hashed = {os.path.basename(p) for p in build._digest_files()}
    for name in os.listdir(build.CSRC):
        if name.endswith((".hip", ".h")):
            assert name in hashed, name
    assert "hipvol.h" in hashed
    # the same tree copied elsewhere hashes the same (the digest names a source state, not a path) ...
    csrc, inc = tmp_path / "pkg" / "csrc", tmp_path / "include"
    shutil.copytree(build.CSRC, csrc)
    shutil.copytree(build.INCLUDE, inc)
    before = build._digest()
    monkeypatch.setattr(build, "CSRC", str(csrc))
    monkeypatch.setattr(build, "INCLUDE", str(inc))
    monkeypatch.setattr(build, "ROOT", str(tmp_path))
    monkeypatch.setattr(build, "HERE", str(tmp_path / "pkg"))
    same = build._digest()
    # ... and touching one header that only other headers' users include changes it
    for header in ("hv_semantic.h", "hv_bucket.h", "hv_query.h"):
        with open(csrc / header, "a") as f:
            f.write("\n// touched\n")
        after = build._digest()
        assert after != same, header
        same = after
    assert before != same
