"""Build tests/standin/lib/libcwipc_util.so: the library's own objects, csrc/exchange.cpp compiled once more with
-DCWIPC_EXCHANGE_TEST_HOOKS, and tests/standin/rccl_standin.cpp in the place of librccl.  TEST INFRASTRUCTURE (like oracle/):
selected by CWIPC_LIBRARY_DIR in the child processes of tests/test_gpu_exchange_ranks.py, never loaded by the product.

Usage: python tests/standin/build_standin.py   (also run by __graft_entry__.build())
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def build(force: bool = False) -> str:
    from cwipc_util_amd import _build as b
    b.build()                                            # the product's objects (and the product itself) are up to date
    obj_dir, lib_dir = os.path.join(HERE, "build"), os.path.join(HERE, "lib")
    os.makedirs(obj_dir, exist_ok=True)
    os.makedirs(lib_dir, exist_ok=True)
    lib = os.path.join(lib_dir, "libcwipc_util.so")
    product_objs = [os.path.join(b.SRC_DIR, "build", s + ".o") for s in b.SOURCES if s != "exchange.cpp"]
    mine = [(os.path.join(b.SRC_DIR, "exchange.cpp"), os.path.join(obj_dir, "exchange_hooks.o"), ["-DCWIPC_EXCHANGE_TEST_HOOKS"]),
            (os.path.join(HERE, "rccl_standin.cpp"), os.path.join(obj_dir, "rccl_standin.o"), [])]
    header_mtime = b._newest_header_mtime()
    objs = list(product_objs)
    for src, obj, extra in mine:
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), header_mtime):
            cmd = [b.hipcc()] + b.COMMON_FLAGS + extra + ["-x", "hip", "-c", src, "-o", obj]
            proc = subprocess.run(cmd, capture_output=True, text=True)
            if proc.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src}:\n{proc.stdout}\n{proc.stderr}")
        objs.append(obj)
    if force or not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs):
        cmd = [b.hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib] + objs     # no -lrccl: the stand-in is inside
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f"link failed:\n{proc.stdout}\n{proc.stderr}")
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
