"""Build libpyslam_hipvol.so for gfx950 in-tree (pyslam_amd/lib/), with plain hipcc.

hipcc cross-compiles without a GPU.  The shared object is git-ignored but travels to the GPU box
with the working-tree snapshot.  Flags that matter for parity:
  -ffp-contract=off   every float op is one IEEE op (no FMA contraction), so key arithmetic and
                      the TSDF update are bit-identical to the CPU reference / restatement;
  (default) -fhip-fp32-correctly-rounded-divide-sqrt   IEEE f32 division and sqrt.
"""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libpyslam_hipvol.so")
SOURCES = ["hv_core.hip", "hv_tsdf.hip", "hv_voxel_grid.hip", "hv_semantic.hip", "hv_semantic_ops.hip", "hv_extract.hip", "hv_prep.hip", "hv_halo.hip"]
INCLUDE = os.path.join(ROOT, "include")


def _digest_files():
    """Every file the library is compiled from: all of csrc/ and include/ (a list of names went stale - three
    headers included by four translation units were not hashed, so a header-only edit reused the old .so)."""
    files = []
    for d in (CSRC, INCLUDE):
        for name in sorted(os.listdir(d)):
            path = os.path.join(d, name)
            if os.path.isfile(path) and name.endswith((".hip", ".h", ".hpp", ".inc")):
                files.append(path)
    return files


ARCH = "gfx950"


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 to build libpyslam_hipvol.so)")


def _flags():
    return [
        f"--offload-arch={ARCH}",
        "-O3",
        "-std=c++17",
        "-fPIC",
        "-ffp-contract=off",
        "-fno-fast-math",
        "-Wall",
        "-Wno-unused-function",
        "-I" + INCLUDE,
        "-I" + CSRC,
    ]


def _digest():
    h = hashlib.sha256()
    for path in _digest_files():
        h.update(os.path.relpath(path, ROOT).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    # the flags enter with the checkout's root written as "." - the digest names a SOURCE state and must be the same on every
    # box the tree travels to (the evidence under profiles/ is keyed by it)
    h.update(" ".join(_flags()).replace(ROOT, ".").encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    """Compile every HIP translation unit for gfx950 and link the shared library."""
    os.makedirs(LIB_DIR, exist_ok=True)
    stamp = os.path.join(LIB_DIR, ".build_digest")
    digest = _digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(stamp):
        if open(stamp).read().strip() == digest:
            return LIB_PATH
    hipcc = _hipcc()
    objs = []

    def compile_one(src):
        obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        cmd = [hipcc] + _flags() + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(digest)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
