"""Run test cases of the reference's own python/test_cwipc_util.py, where it lies under /root/reference, against
THIS repository's libcwipc_util.so (the drop-in claim of INTEGRATION.md section 1).  Child process of
tests/test_reference_wrapper.py; never used on the GPU box (the reference does not travel).

The reference wrapper imports open3d at module level (python/cwipc/util.py:24), which this image lacks and
which none of the selected test cases touches: an empty placeholder module takes its place, in memory only.
usage: run_reference_tests.py <reference_root> <library_dir> <test names...>
"""
import importlib.machinery
import os
import sys
import types
import unittest


def placeholder(name: str) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []

    class _Anything:
        def __init__(self, *a, **k): pass
        def __getattr__(self, n): return _Anything

    m.__getattr__ = lambda n: _Anything
    sys.modules[name] = m
    return m


def main() -> int:
    ref_root, libdir, names = sys.argv[1], sys.argv[2], sys.argv[3:]
    sys.dont_write_bytecode = True                      # nothing is written into the reference tree
    top = placeholder("open3d")
    for sub in ("geometry", "utility", "visualization", "io", "pipelines"):
        setattr(top, sub, placeholder("open3d." + sub))
    setattr(sys.modules["open3d.pipelines"], "registration", placeholder("open3d.pipelines.registration"))
    sys.path.insert(0, os.path.join(ref_root, "python"))
    import cwipc.util
    # an absolute path given before first use selects the native library (reference python/cwipc/util.py:368-385)
    cwipc.util.cwipc_util_dll_load(os.path.join(libdir, "libcwipc_util.so"))
    import test_cwipc_util                              # the reference's file, unmodified
    suite = unittest.TestSuite(test_cwipc_util.TestApi(n) for n in names)
    result = unittest.TextTestRunner(verbosity=2, stream=sys.stdout).run(suite)
    print("RAN", result.testsRun, "FAILED", len(result.failures), "ERRORS", len(result.errors), "SKIPPED", len(result.skipped))
    return 0 if result.wasSuccessful() else 1


if __name__ == "__main__":
    sys.exit(main())
