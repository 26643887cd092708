import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cwipc():
    """The product package, with the HIP library built if it is missing."""
    # torch first: it brings its own copy of the HIP runtime, and a process can hold only one.  Loaded after
    # libcwipc_util.so (which is linked against /opt/rocm's), torch finds "No HIP GPUs"; the other way round both share
    # torch's copy.  (INTEGRATION.md section 3 states the same rule for applications that use both.)
    import torch  # noqa: F401
    import cwipc_util_amd
    try:
        cwipc_util_amd.cwipc_util_dll_load()
    except RuntimeError:
        from cwipc_util_amd import _build
        _build.build()
        cwipc_util_amd.cwipc_util_dll_load()
    return cwipc_util_amd


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle as o
    o.load()
    return o


@pytest.fixture(scope="session")
def gpu(cwipc):
    if cwipc.cwipc_hip_device_count() < 1:
        pytest.fail("test marked gpu but no HIP device is visible (the product has no CPU fallback)")
    cwipc.cwipc_hip_set_device(0)
    return cwipc


_synth_cache = {}


@pytest.fixture(scope="session")
def synth(oracle):
    """synth(npoints, angle=0.0) -> (points, cellsize), cached."""
    def make(npoints, angle=0.0):
        key = (npoints, angle)
        if key not in _synth_cache:
            _synth_cache[key] = oracle.synthetic(npoints, angle)
        return _synth_cache[key]
    return make


def make_cloud(cwipc, pts, cellsize=0.0, timestamp=1234):
    pc = cwipc.cwipc_from_numpy_array(np.ascontiguousarray(pts), timestamp)
    pc._set_cellsize(cellsize)
    return pc
