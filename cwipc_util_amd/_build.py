"""Build libcwipc_util.so (the MI355X drop-in library) in-tree with hipcc for gfx950.

Usage:  python -m cwipc_util_amd._build [--force]

The shared object lands in cwipc_util_amd/lib/ -- a directory named "lib" in an
ancestor of the package is exactly where the reference's loader looks
(reference python/cwipc/util.py:203-223), and the built file travels to the GPU
box with the repository snapshot.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
SRC_DIR = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(SRC_DIR, "build")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libcwipc_util.so")

SOURCES = [
    "logging.cpp",
    "device.cpp",
    "pointcloud.cpp",
    "synthetic.cpp",
    "stubs.cpp",
    "ply.cpp",
    "filters.cpp",
    "exchange.cpp",
    "kernels_basic.hip",
    "kernels_voxel.hip",
    "kernels_sor.hip",
]

# -ffp-contract=off: the parity contract is stated in separately rounded fp32/f64
# operations (PCL / FLANN / Python float semantics); hipcc would otherwise fuse a*b+c.
COMMON_FLAGS = [
    "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
    "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-unused-result",
    "-DCWIPC_VERSION=amd-gfx950-r4",
    "-I" + os.path.join(REPO_DIR, "include"), "-I" + SRC_DIR,
]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X library cannot be built")
    return exe


def _newest_header_mtime() -> float:
    newest = 0.0
    for root in (SRC_DIR, os.path.join(REPO_DIR, "include")):
        for dirpath, _dirs, files in os.walk(root):
            for f in files:
                if f.endswith((".h", ".hpp", ".inc")):
                    newest = max(newest, os.path.getmtime(os.path.join(dirpath, f)))
    return newest


def _compile(src: str, force: bool, header_mtime: float) -> str:
    src_path = os.path.join(SRC_DIR, src)
    obj_path = os.path.join(OBJ_DIR, src + ".o")
    if not force and os.path.exists(obj_path):
        if os.path.getmtime(obj_path) >= max(os.path.getmtime(src_path), header_mtime):
            return obj_path
    cmd = [hipcc()] + COMMON_FLAGS + ["-x", "hip", "-c", src_path, "-o", obj_path]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{proc.stdout}\n{proc.stderr}")
    if proc.stderr.strip():
        sys.stderr.write(proc.stderr)
    return obj_path


def build(force: bool = False, jobs: int = 6, debug_knobs: bool = False) -> str:
    """debug_knobs: a second copy of the library with -DCWIPC_DEBUG_KNOBS (stage switches for timing experiments; results are
    wrong when they are set) in scratch/lib_dbg/ -- select it with CWIPC_LIBRARY_DIR.  The shipped library has none of them."""
    global OBJ_DIR, LIB_DIR, LIB_PATH
    if debug_knobs:
        OBJ_DIR = os.path.join(SRC_DIR, "build_dbg")
        LIB_DIR = os.path.join(REPO_DIR, "scratch", "lib_dbg")
        LIB_PATH = os.path.join(LIB_DIR, "libcwipc_util.so")
        if "-DCWIPC_DEBUG_KNOBS" not in COMMON_FLAGS:
            COMMON_FLAGS.append("-DCWIPC_DEBUG_KNOBS")
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    header_mtime = _newest_header_mtime()
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        objs = list(pool.map(lambda s: _compile(s, force, header_mtime), SOURCES))
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(o) > os.path.getmtime(LIB_PATH) for o in objs):
        # RCCL for the multi-GPU join (exchange.cpp).  Where torch has been imported first its librccl.so.1 is already in the
        # process and serves this dependency too (same soname), as with the HIP runtime.
        cmd = [hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB_PATH] + objs + ["-L/opt/rocm/lib", "-lrccl"]
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f"link failed:\n{proc.stdout}\n{proc.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, debug_knobs="--debug-knobs" in sys.argv))
